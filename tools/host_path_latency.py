#!/usr/bin/env python3
"""Host-pointer entry point (the reference's drop-in boundary), small buffers: time per call and GiB/s, next to the
CPU port (AVX-512BW/AVX2, one core) on the same buffer.  Shows where the PCIe round trip stops paying."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402,F401

import dxt_lossless_transform_amd as pkg  # noqa: E402
from oracle import oracle_c  # noqa: E402

pkg.load()
st = pkg.Bc1TransformSettings()
rows = []
for kib in [int(k) for k in os.environ.get("LAT_KIB", "64,256,1024,4096,16384,65536").split(",")]:
    n = kib << 10
    x = np.random.default_rng(1).integers(0, 256, n, dtype=np.uint8)
    y = np.empty_like(x)
    for _ in range(3):
        pkg.transform_bc1_with_settings(x, y, st)
    reps = 200 if kib <= 4096 else 20
    t0 = time.perf_counter()
    for _ in range(reps):
        pkg.transform_bc1_with_settings(x, y, st)
    gpu = (time.perf_counter() - t0) / reps
    oracle_c.run_bc1_default_simd(x, y, False, 1)
    t0 = time.perf_counter()
    for _ in range(reps):
        oracle_c.run_bc1_default_simd(x, y, False, 1)
    cpu = (time.perf_counter() - t0) / reps
    rows.append({"KiB": kib, "gpu_host_path_us": round(gpu * 1e6, 1), "gpu_GiBps": round(n / gpu / 2**30, 2),
                 "cpu_1core_us": round(cpu * 1e6, 1), "cpu_GiBps": round(n / cpu / 2**30, 2)})
print(json.dumps({"isa": oracle_c.SIMD_NAMES[oracle_c.simd_level()], "rows": rows}))
