#!/usr/bin/env python3
"""BC1 forward transform with block normalisation fused in (reference experimental module) next to the plain forward
transform and the stand-alone normalise kernel: ms and fraction of the 8 TB/s HBM peak on the algorithmic 2*len."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dxt_lossless_transform_amd as pkg  # noqa: E402
from dxt_lossless_transform_amd import normalize as norm  # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
share = sys.argv[2] if len(sys.argv) > 2 else "mixed"   # "mixed": 1/4 solid + 1/4 transparent blocks; "random": none
dev = torch.device("cuda:0")
n = int(gib * (1 << 30)) // 8
x = torch.empty(n * 8, dtype=torch.uint8, device=dev)
pkg.fill_splitmix64(x, 0x0BC14E01)
if share == "mixed":
    b = x.view(-1, 8)
    for lo in range(0, n, 1 << 26):   # in slices: boolean index temporaries are large
        hi = min(n, lo + (1 << 26))
        k = torch.arange(lo, hi, device=dev) % 4
        v = b[lo:hi]
        v[k == 1, 4:] = 0
        rows = (k == 2).nonzero().squeeze(1)
        v[rows, 0:2] = 0
        v[rows, 4:] = 0xFF
        del k, rows
y = torch.empty_like(x)
D = norm.Bc1TransformDetailsWithNormalization
M = norm.ColorNormalizationMode


def timed(fn, steps=10):
    import time as _t
    _t0 = _t.perf_counter()
    while (_t.perf_counter() - _t0) < 0.1:   # warm up by wall time: the chip ramps its clocks for ~40 ms after idling (profiles/r03_clock_ramp.txt)
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(steps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / steps
    return round(ms, 4), round(2 * x.numel() / (ms * 1e-3) / 8e12, 4)


res = {"workload": f"BC1 {gib:g} GiB, {share} blocks"}
res["plain_transform"] = timed(lambda: pkg.transform_bc1_with_settings(x, y))
for m in (M.COLOR0_ONLY, M.REPLICATE_COLOR):
    res[f"fused_{m.name.lower()}"] = timed(lambda: norm.transform_bc1_with_normalize_blocks(x, y, D(m, 1, True)))
res["normalize_only_color0"] = timed(lambda: norm.normalize_blocks(x, y, M.COLOR0_ONLY))
res["normalize_in_place"] = timed(lambda: norm.normalize_blocks(y, y, M.COLOR0_ONLY))
print(json.dumps(res))
