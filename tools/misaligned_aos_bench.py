#!/usr/bin/env python3
"""AoS pointer off 16-byte alignment (a DDS file that lies whole in HBM: payload at byte 128 + 20 = 148): tile kernels
with unaligned vector accesses (default) against the element kernel (round 1's rule, switch 0x1000).  4 GiB."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dxt_lossless_transform_amd as pkg  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, steps=20):
    for _ in range(2):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(steps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / steps


out = {}
for fmt, B in (("bc1", 8), ("bc3", 16)):
    n = (4 << 30) // B
    for shift in (0, 4, 20, 1):
        for force, label in ((0, "tiles"), (0x1000, "element kernel")):
            if shift == 0 and force:
                continue
            raw = torch.empty(n * B + 64, dtype=torch.uint8, device=dev)
            x = raw[shift:shift + n * B]
            pkg.fill_splitmix64(raw, 7)
            y = torch.empty(n * B, dtype=torch.uint8, device=dev)
            zraw = torch.empty(n * B + 64, dtype=torch.uint8, device=dev)
            z = zraw[shift:shift + n * B]
            pkg.set_tuning(0, force)
            f = getattr(pkg, f"transform_{fmt}_with_settings")
            g = getattr(pkg, f"untransform_{fmt}_with_settings")
            tf, ti = timed(lambda: f(x, y)), timed(lambda: g(y, z))
            assert torch.equal(x, z)
            pkg.set_tuning(0, 0)
            out[f"{fmt} aos+{shift} {label}"] = [round(2 * n * B / (tf * 1e-3) / 8e12, 3), round(2 * n * B / (ti * 1e-3) / 8e12, 3)]
            del raw, x, y, zraw, z
for k, v in out.items():
    print(f"{k:34s} fwd {v[0]:.3f}  inv {v[1]:.3f}")
