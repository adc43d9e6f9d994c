#!/usr/bin/env python3
"""PCIe-inclusive throughput of the host-pointer entry points (the drop-in bodies of transform_bcN_with_settings):
H2D + kernel + D2H, pageable and pinned host buffers.  Not the headline metric -- see DESIGN.md "Measurement"."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dxt_lossless_transform_amd as pkg  # noqa: E402

pkg.load()
out = {}
for fmt, st in (("bc1", pkg.Bc1TransformSettings()), ("bc3", pkg.Bc3TransformSettings())):
    for mib in (8, 256, 2048):
        n = mib << 20
        for kind in ("pageable", "pinned"):
            if kind == "pinned":
                xt = torch.empty(n, dtype=torch.uint8).pin_memory()
                yt = torch.empty(n, dtype=torch.uint8).pin_memory()
                x, y = xt.numpy(), yt.numpy()
            else:
                x, y = np.empty(n, dtype=np.uint8), np.empty(n, dtype=np.uint8)
            x[:] = np.random.default_rng(1).integers(0, 256, n, dtype=np.uint8)
            y[:] = 0
            f = getattr(pkg, f"transform_{fmt}_with_settings")
            f(x, y, st)
            reps = 5 if mib <= 256 else 3
            t0 = time.perf_counter()
            for _ in range(reps):
                f(x, y, st)
            dt = (time.perf_counter() - t0) / reps
            out[f"{fmt}_{mib}MiB_{kind}"] = round(n / dt / 2**30, 2)
print(json.dumps(out))
