#!/usr/bin/env python3
"""Block decoders (include/dxtlt_decode.h): ms and fraction of the 8 TB/s HBM peak on the algorithmic bytes
(block bytes in + 64 bytes of pixels out per block; the difference count reads two block arrays)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dxt_lossless_transform_amd as pkg  # noqa: E402
from dxt_lossless_transform_amd import decode as dec  # noqa: E402

pixel_gib = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
dev = torch.device("cuda:0")
n = int(pixel_gib * (1 << 30)) // 64
out = torch.empty(n * 64, dtype=torch.uint8, device=dev)


def timed(fn, nbytes, steps=10):
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
    return round(ms, 4), round(nbytes / (ms * 1e-3) / 8e12, 4)


res = {"workload": f"{n} blocks -> {pixel_gib:g} GiB of RGBA8888"}
res["write_ceiling_torch_fill"] = timed(lambda: out.fill_(0x5A), n * 64)
res["copy_ceiling_torch_copy"] = timed(lambda: out[: n * 32].copy_(out[n * 32:]), n * 64)
for fmt, bs in (("bc1", 8), ("bc2", 16), ("bc3", 16)):
    x = torch.empty(n * bs, dtype=torch.uint8, device=dev)
    pkg.fill_splitmix64(x, 0xDEC0 + bs)
    res[f"decode_{fmt}"] = timed(lambda: dec.decode_blocks(fmt, x, out), n * (bs + 64))
    y = x.clone()
    y[::4097] ^= 1
    res[f"differences_{fmt}_sparse"] = timed(lambda: dec.count_pixel_differences(fmt, x, y), 2 * n * bs, steps=5)
    pkg.fill_splitmix64(y, 0xD1FF + bs)
    res[f"differences_{fmt}_all_differ"] = timed(lambda: dec.count_pixel_differences(fmt, x, y), 2 * n * bs, steps=5)
    del x, y
print(json.dumps(res))
