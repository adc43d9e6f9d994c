import os, sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import dxt_lossless_transform_amd as pkg
from oracle import oracle_c
gib = int(os.environ.get("PROBE_GIB", "4"))
n = gib << 30
src = oracle_c.fill_splitmix64(n, 7)
res = {}
for kind in ("pageable", "pinned"):
    if kind == "pinned":
        X = torch.empty(n, dtype=torch.uint8, pin_memory=True); Y = torch.empty(n, dtype=torch.uint8, pin_memory=True); Z = torch.empty(n, dtype=torch.uint8, pin_memory=True)
        x, y, z = X.numpy(), Y.numpy(), Z.numpy(); x[:] = src
    else:
        x, y, z = src, np.empty_like(src), np.empty_like(src)
    st = pkg.Bc3TransformSettings()
    for name, fn in (("host_fwd", lambda: pkg.transform_bc3_with_settings(x, y, st)), ("host_inv", lambda: pkg.untransform_bc3_with_settings(y, z, st)),
                     ("sharded_fwd", lambda: pkg.transform_sharded("bc3", False, x, y, st, 1)), ("sharded_inv", lambda: pkg.transform_sharded("bc3", True, y, z, st, 1))):
        fn(); best = None
        for _ in range(2):
            t = time.perf_counter(); fn(); dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
        res[f"{kind}_{name}"] = round(gib / best, 1)
    assert np.array_equal(z, x)
print(os.environ.get("DXTLT_PIPELINE_CHUNK_BYTES", "default"), res, flush=True)
