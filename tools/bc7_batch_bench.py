#!/usr/bin/env python3
"""BC7 buffers through the batch calls (format 7) against one call per buffer: 1024 x 1 MiB device-resident (launch
bound without the batch), and the same from host memory.  usage: python tools/bc7_batch_bench.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import dxt_lossless_transform_amd as pkg  # noqa: E402
from dxt_lossless_transform_amd import batch, bc7  # noqa: E402

dev = torch.device("cuda:0")
out = {}
for count, mib in ((1024, 1), (256, 4), (4096, 0.25)):
    n = int(mib * (1 << 20))
    big = torch.empty(count * n, dtype=torch.uint8, device=dev)
    pkg.fill_splitmix64(big, 7)
    xs = [big[i * n:(i + 1) * n] for i in range(count)]
    ys = [torch.empty_like(x) for x in xs]
    items = [("bc7", False, x, y, None) for x, y in zip(xs, ys)]

    def timed(fn, reps=5):
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps

    # the C call alone (the Python wrapper spends 2 us per item building the ctypes array)
    import ctypes as C
    arr = (batch.DxtltBatchItem * count)()
    for k, (x, y) in enumerate(zip(xs, ys)):
        arr[k].d_input, arr[k].d_output, arr[k].len, arr[k].format = x.data_ptr(), y.data_ptr(), n, 7
    lib = pkg.load()
    lib.dxtlt_transform_batch_device.argtypes = [C.POINTER(batch.DxtltBatchItem), C.c_size_t, C.c_void_p]
    lib.dxtlt_transform_batch_device.restype = C.c_int32
    stream = torch.cuda.current_stream().cuda_stream

    def run_batch():
        assert lib.dxtlt_transform_batch_device(arr, count, stream) == 0

    t_batch = timed(run_batch)
    ref = [y.clone() for y in ys[:3]]
    t_each = timed(lambda: [bc7.transform_bc7(x, y) for x, y in zip(xs, ys)], reps=2)
    assert all(torch.equal(a, b) for a, b in zip(ref, ys[:3]))
    hx = [x.cpu().numpy() for x in xs]
    hy = [np.empty_like(h) for h in hx]
    prepared = batch.prepare_batch_host([("bc7", False, a, b, None) for a, b in zip(hx, hy)])
    batch.run_prepared_batch_host(prepared)
    t = time.perf_counter(); batch.run_prepared_batch_host(prepared); t_host = time.perf_counter() - t
    assert np.array_equal(hy[1], ys[1].cpu().numpy())
    total = count * n
    out[f"{count} x {mib} MiB"] = {"batch_device_GiBps": round(total / t_batch / 2**30, 1), "one_call_per_buffer_GiBps": round(total / t_each / 2**30, 1),
                                  "batch_host_GiBps": round(total / t_host / 2**30, 1)}
    del big, xs, ys, items
print(json.dumps(out))
