#!/usr/bin/env python3
"""Array-level RGB565 colour operations (include/dxtlt_color565.h): ms and fraction of the 8 TB/s HBM peak on the
algorithmic 2 * bytes (one read + one write) for a colour array of the given size, device resident."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dxt_lossless_transform_amd as pkg  # noqa: E402
from dxt_lossless_transform_amd import color565 as mod  # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
dev = torch.device("cuda:0")
nbytes = int(gib * (1 << 30)) // 16 * 16
x = torch.empty(nbytes, dtype=torch.uint8, device=dev)
pkg.fill_splitmix64(x, 0x0C565001)
y = torch.empty_like(x)
half = nbytes // 2


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
    return round(ms, 4), round(2 * nbytes / (ms * 1e-3) / 8e12, 4)


res = {"workload": f"{gib:g} GiB of RGB565 colours"}
for v in (1, 2, 3):
    res[f"decorrelate_v{v}"] = timed(lambda: mod.decorrelate_ycocg_r(x, y, v))
    res[f"recorrelate_v{v}"] = timed(lambda: mod.recorrelate_ycocg_r(x, y, v))
res["decorrelate_v1_in_place"] = timed(lambda: mod.decorrelate_ycocg_r(y, y, 1))
res["split_color_endpoints"] = timed(lambda: mod.split_color_endpoints(x, y))
res["recorrelate_split_v1"] = timed(lambda: mod.recorrelate_ycocg_r_split(x[:half], x[half:], y, 1))
res["recorrelate_split_none"] = timed(lambda: mod.recorrelate_ycocg_r_split(x[:half], x[half:], y, 0))
print(json.dumps(res))
