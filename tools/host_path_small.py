import json, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import dxt_lossless_transform_amd as pkg
pkg.load()
out = {}
st = pkg.Bc1TransformSettings()
for mib in [int(v) for v in os.environ.get('SIZES_MIB', '1,4,8,16,32,64').split(',')]:
    n = mib << 20
    x = np.random.default_rng(1).integers(0, 256, n, dtype=np.uint8); y = np.empty_like(x)
    pkg.transform_bc1_with_settings(x, y, st)
    reps = 20 if mib <= 64 else 6
    t0 = time.perf_counter()
    for _ in range(reps): pkg.transform_bc1_with_settings(x, y, st)
    out[f"{mib}MiB"] = round(n / ((time.perf_counter() - t0) / reps) / 2**30, 2)
print(os.environ.get("DXTLT_PIPELINE_MIN_BYTES"), os.environ.get("DXTLT_PIPELINE_CHUNK_BYTES"), json.dumps(out))
