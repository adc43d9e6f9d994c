#!/usr/bin/env python3
"""transform_bcN_auto, device side: the fused candidate kernel (one read of the input -> every endpoint section) against
one full transform per candidate (DXTLT_AUTO_FUSED=0).  256 MiB per format, a constant-time estimator in C (the callback
returns the length), so what is left is upload + kernels + the per-candidate section downloads.  Run under
`rocprofv3 --kernel-trace --stats` to see the kernel list; prints wall time per call.

    python tools/auto_bench.py            # fused (default)
    DXTLT_AUTO_FUSED=0 python tools/auto_bench.py
"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402

import cabi  # noqa: E402
import dxt_lossless_transform_amd as pkg  # noqa: E402
from oracle import oracle_c  # noqa: E402

lib = cabi.bind(C.CDLL(pkg._lib.lib_path()))
out = {"fused": os.environ.get("DXTLT_AUTO_FUSED", "1") != "0"}
nbytes = 256 << 20
x = oracle_c.fill_splitmix64(nbytes, 0xA070)
y = np.zeros_like(x)
for n in (1, 2, 3):
    for use_all in (False, True):
        made = cabi.zstd_c_estimator(None)            # size = len, in C: a Python callback would copy every section it is shown
        est = made[0] if made else cabi.make_estimator("dummy")[0]
        settings = {1: cabi.CoreSettings2, 2: cabi.CoreSettings2, 3: cabi.CoreSettings3}[n]()
        f = getattr(lib, f"dltbc{n}core_transform_auto")
        f(x.ctypes.data, x.size, y.ctypes.data, y.size, C.byref(est), cabi.AutoSettings(use_all), C.byref(settings))
        best = None
        for _ in range(3):
            t = time.perf_counter()
            r = f(x.ctypes.data, x.size, y.ctypes.data, y.size, C.byref(est), cabi.AutoSettings(use_all), C.byref(settings))
            dt = time.perf_counter() - t
            assert r.ErrorCode == 0
            best = dt if best is None else min(best, dt)
        out[f"bc{n}_{'all' if use_all else 'fast'}_ms"] = round(best * 1e3, 2)
print(json.dumps(out))
