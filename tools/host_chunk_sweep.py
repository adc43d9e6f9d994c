"""Host-pointer path of BC1 / BC2 / BC3 (default settings) by buffer size (SWEEP_FMTS, SWEEP_MIB); DXTLT_PIPELINE_CHUNK_BYTES steers the
chunk size, DXTLT_PIPELINE_MIN_BYTES the size from which the chunked pipeline is used."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dxt_lossless_transform_amd as pkg
from oracle import oracle_c
out = {}
for fmt in os.environ.get("SWEEP_FMTS", "bc1,bc2,bc3").split(","):
    f = getattr(pkg, f"transform_{fmt}_with_settings"); g = getattr(pkg, f"untransform_{fmt}_with_settings")
    out[fmt] = {}
    for mib in [int(m) for m in os.environ.get("SWEEP_MIB", "128,256,512,1024").split(",")]:
        x = oracle_c.fill_splitmix64(mib << 20, 7); y = np.empty_like(x); z = np.empty_like(x)
        f(x, y)
        best = None
        for _ in range(3):
            t = time.perf_counter(); f(x, y); dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
        g(y, z)
        bi = None
        for _ in range(3):
            t = time.perf_counter(); g(y, z); dt = time.perf_counter() - t
            bi = dt if bi is None else min(bi, dt)
        assert np.array_equal(z, x)
        out[fmt][mib] = (round(mib / 1024 / best, 1), round(mib / 1024 / bi, 1))
print("chunk", os.environ.get("DXTLT_PIPELINE_CHUNK_BYTES", "default"), "min", os.environ.get("DXTLT_PIPELINE_MIN_BYTES", "default"), out, flush=True)
