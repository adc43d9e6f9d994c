#!/usr/bin/env python3
"""Host-pointer entry point (dxtlt_transform_bc1, forward + inverse) with pageable and with page-locked caller buffers:
GiB/s of blocks through the PCIe round trip."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import dxt_lossless_transform_amd as pkg  # noqa: E402

rows = []
for mib in (64, 256, 1024, 4096):
    n = mib << 20
    row = {"MiB": mib}
    for kind in ("pageable", "pinned"):
        if kind == "pinned":
            x, y, z = (torch.empty(n, dtype=torch.uint8, pin_memory=True).numpy() for _ in range(3))
        else:
            x, y, z = (np.empty(n, dtype=np.uint8) for _ in range(3))
        x[:] = np.random.default_rng(mib).integers(0, 256, n, dtype=np.uint8)
        y[:] = 0
        z[:] = 0
        pkg.transform_bc1_with_settings(x, y)
        pkg.untransform_bc1_with_settings(y, z)
        assert np.array_equal(x, z)
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            pkg.transform_bc1_with_settings(x, y)
        t1 = time.perf_counter()
        for _ in range(reps):
            pkg.untransform_bc1_with_settings(y, z)
        t2 = time.perf_counter()
        row[kind] = {"fwd_GiBps": round(n * reps / (t1 - t0) / 2**30, 1), "inv_GiBps": round(n * reps / (t2 - t1) / 2**30, 1)}
        del x, y, z
    rows.append(row)
print(json.dumps(rows))
