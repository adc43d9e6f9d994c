"""dxtlt_transform_batch_host: does a longer batch help?  1024 / 4096 / 8192 x 1 MiB BC1 host buffers, C call only."""
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import dxt_lossless_transform_amd as pkg
from dxt_lossless_transform_amd import batch
from oracle import oracle_c
st = pkg.Bc1TransformSettings()
for count in (1024, 4096, 8192):
    nbytes = 1 << 20
    src = oracle_c.fill_splitmix64(count * nbytes, 0xB47C)
    xs = [src[i * nbytes:(i + 1) * nbytes] for i in range(count)]
    ys = [np.empty(nbytes, dtype=np.uint8) for _ in range(count)]
    prepared = batch.prepare_batch_host([("bc1", False, x, y, st) for x, y in zip(xs, ys)])
    batch.run_prepared_batch_host(prepared)
    best = None
    for _ in range(3):
        t = time.perf_counter(); batch.run_prepared_batch_host(prepared); dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    assert np.array_equal(ys[count // 2], oracle_c.transform("bc1", xs[count // 2], 1, True))
    print(count, "x 1 MiB:", round(count * nbytes / best / 2**30, 2), "GiB/s", flush=True)
    del src, xs, ys, prepared
