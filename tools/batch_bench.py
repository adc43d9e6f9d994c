#!/usr/bin/env python3
"""Many small device-resident BC1 textures: one call per texture on one stream, the same calls replayed from a HIP
graph, and dxtlt_transform_batch_device.
GiB/s of blocks transformed (forward only), Python call overhead included in both."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dxt_lossless_transform_amd as pkg  # noqa: E402
from dxt_lossless_transform_amd import batch  # noqa: E402

dev = torch.device("cuda:0")
st = pkg.Bc1TransformSettings()
rows = []
extra = int(sys.argv[1]) if len(sys.argv) > 1 else 0   # extra blocks per buffer (1 -> odd, mip-chain-like counts)
for kib, count in ((256, 1024), (1024, 1024), (4096, 512), (16384, 128)):
    n = (kib << 10) + 8 * extra
    xs = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(count)]
    ys = [torch.empty_like(x) for x in xs]
    for k, x in enumerate(xs):
        pkg.fill_splitmix64(x, k)
    items = [("bc1", False, x, y, st) for x, y in zip(xs, ys)]

    def sequential():
        for x, y in zip(xs, ys):
            pkg.transform_bc1_with_settings(x, y, st)

    def batched():
        batch.transform_batch(items)

    # the C call alone, item array prepared once (what a C/C++/Rust caller pays; the Python wrapper above spends ~2 us
    # per item building the array)
    import ctypes as C
    arr = (batch.DxtltBatchItem * count)()
    for k, (x, y) in enumerate(zip(xs, ys)):
        arr[k].d_input, arr[k].d_output, arr[k].len = x.data_ptr(), y.data_ptr(), n
        arr[k].format, arr[k].decorrelation_mode, arr[k].split_colour_endpoints = 1, 1, 1
    lib = pkg.load()
    lib.dxtlt_transform_batch_device.argtypes = [C.POINTER(batch.DxtltBatchItem), C.c_size_t, C.c_void_p]
    lib.dxtlt_transform_batch_device.restype = C.c_int32
    stream = torch.cuda.current_stream().cuda_stream

    def batched_c():
        assert lib.dxtlt_transform_batch_device(arr, count, stream) == 0

    # the same per-texture calls captured once into a HIP graph and replayed (fixed buffers: the per-frame case)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        sequential()

    res = {}
    for name, fn in (("sequential", sequential), ("graph_replay", graph.replay), ("batched", batched), ("batched_c_call", batched_c)):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        res[name] = {"ms": round(dt * 1e3, 3), "GiBps": round(n * count / dt / 2**30, 1), "us_per_item": round(dt / count * 1e6, 2)}
    rows.append({"KiB": kib, "extra_blocks": extra, "count": count, **res})
print(json.dumps(rows))
