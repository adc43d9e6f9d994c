import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dxt_lossless_transform_amd as pkg
from dxt_lossless_transform_amd import bc7
from oracle import oracle_c
rows = {}
for kib in (16, 64, 256, 512, 1024):
    n = kib << 10
    x = oracle_c.fill_splitmix64(n, 3); oracle_c.bc7_force_modes(x)
    y = np.empty_like(x)
    for _ in range(5): bc7.transform_bc7(x, y)
    t = time.perf_counter()
    for _ in range(200): bc7.transform_bc7(x, y)
    rows[kib] = round((time.perf_counter() - t) / 200 * 1e6, 1)
print(os.environ.get("DXTLT_MAPPED_MAX_BYTES", "default"), rows)
