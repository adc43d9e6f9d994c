"""dxtlt_transform_batch_host on a MIXED batch -- BC1 forward, BC7 forward and BC7 inverse items alternating, 1 MiB each -- end to end
(host memory in, host memory out), and the same items as device batches issued back to back without a synchronisation.  Round 5 staged
the tables of such a call in three ring slots, so the next chunk's call waited on the host for this one's kernels; round 6
stages them in one.  Run once per library (DXTLT_LIB_PATH) on one box.   usage: python tools/batch_mixed_host_probe.py [count]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import dxt_lossless_transform_amd as pkg
from dxt_lossless_transform_amd import batch
from oracle import oracle_c

count = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
n = 1 << 20
pkg.load()
print("library:", pkg._lib.lib_path(), flush=True)
rng = np.random.default_rng(7)
st = pkg.Bc1TransformSettings()
src7 = rng.integers(0, 256, n, dtype=np.uint8)
tr7 = oracle_c.transform_bc7(src7)
src1 = oracle_c.fill_splitmix64(n, 11)
want1 = oracle_c.transform("bc1", src1, 1, True, True)
items, checks = [], []
for i in range(count):
    kind = i % 3
    x = (src1, src7, tr7)[kind].copy()
    y = np.empty_like(x)
    items.append((("bc1", "bc7", "bc7")[kind], kind == 2, x, y, st if kind == 0 else None))
    checks.append((y, (want1, tr7, src7)[kind]))
prep = batch.prepare_batch_host(items)
for _ in range(2):
    batch.run_prepared_batch_host(prep)
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    batch.run_prepared_batch_host(prep)
    best = min(best, time.perf_counter() - t0)
assert all(np.array_equal(y, w) for y, w in checks[:: max(1, count // 48)])
print(f"  host batch, {count} x 1 MiB mixed (BC1 fwd / BC7 fwd / BC7 inv): {count * n / best / 2**30:.1f} GiB/s end to end", flush=True)

# the same mix as DEVICE batches of 96 items, 16 calls back to back: host time to ENQUEUE them (an asynchronous call returns at once)
dev = torch.device("cuda:0")
calls = []
for c in range(16):
    its = []
    for i in range(96):
        kind = i % 3
        xd = torch.from_numpy((src1, src7, tr7)[kind]).to(dev)
        its.append((("bc1", "bc7", "bc7")[kind], kind == 2, xd, torch.empty_like(xd), st if kind == 0 else None))
    calls.append(batch.prepare_batch(its))
for p in calls:
    batch.run_prepared_batch(p)
torch.cuda.synchronize()
best_enq, best_all = 1e9, 1e9
for _ in range(5):
    t0 = time.perf_counter()
    for p in calls:
        batch.run_prepared_batch(p)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    best_enq, best_all = min(best_enq, t1 - t0), min(best_all, t2 - t0)
print(f"  16 device batch calls of 96 mixed items: enqueued in {best_enq * 1e6:.0f} us, finished in {best_all * 1e6:.0f} us", flush=True)
