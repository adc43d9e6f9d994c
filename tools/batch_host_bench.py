#!/usr/bin/env python3
"""dxtlt_transform_batch_host against one host-pointer call per buffer: N host buffers of S MiB each (BC1, default
settings), end to end (host memory in, host memory out), GiB/s of input."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import dxt_lossless_transform_amd as pkg  # noqa: E402
from dxt_lossless_transform_amd import batch  # noqa: E402
from oracle import oracle_c  # noqa: E402

out = {}
st = pkg.Bc1TransformSettings()
for count, mib in ((1024, 1), (256, 4), (4096, 0.25), (64, 16)):
    nbytes = int(mib * (1 << 20))
    src = oracle_c.fill_splitmix64(count * nbytes, 0xB47C)
    xs = [src[i * nbytes:(i + 1) * nbytes] for i in range(count)]
    ys = [np.empty(nbytes, dtype=np.uint8) for _ in range(count)]
    items = [("bc1", False, x, y, st) for x, y in zip(xs, ys)]
    prepared = batch.prepare_batch_host(items)   # the C item array; the timed part is the C call alone
    batch.run_prepared_batch_host(prepared)      # warm-up: staging buffers, pinned arenas, streams
    best = None
    for _ in range(4):
        t = time.perf_counter()
        batch.run_prepared_batch_host(prepared)
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    want = oracle_c.transform("bc1", xs[count // 2], 1, True)
    assert np.array_equal(ys[count // 2], want)
    per_call = None
    if count <= 1024:
        t = time.perf_counter()
        for x, y in zip(xs, ys):
            pkg.transform_bc1_with_settings(x, y, st)
        per_call = time.perf_counter() - t
    out[f"{count} x {mib} MiB"] = {"batch_host_GiBps": round(count * nbytes / best / 2**30, 2),
                                  "one_call_per_buffer_GiBps": round(count * nbytes / per_call / 2**30, 2) if per_call else None}
print(json.dumps(out, indent=1))
