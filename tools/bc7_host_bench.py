"""BC7 host-pointer path (dxtlt_transform_bc7): GiB/s end to end at several sizes; DXTLT_PIPELINE_CHUNK_BYTES /
DXTLT_PIPELINE_MIN_BYTES steer the chunked pipeline (experiments)."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dxt_lossless_transform_amd as pkg
from dxt_lossless_transform_amd import bc7
from oracle import oracle_c
res = {}
for mib in (64, 128, 256, 512, 1024, 2048):
    x = oracle_c.fill_splitmix64(mib << 20, 7)
    y = np.empty_like(x); z = np.empty_like(x)
    bc7.transform_bc7(x, y)
    best = None
    for _ in range(3):
        t = time.perf_counter(); bc7.transform_bc7(x, y); dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    bc7.untransform_bc7(y, z)
    assert np.array_equal(z, x)
    res[mib] = round((mib / 1024) / best, 2)
print(os.environ.get("DXTLT_PIPELINE_CHUNK_BYTES", "default chunk"), os.environ.get("DXTLT_PIPELINE_MIN_BYTES", "default min"), res, flush=True)
