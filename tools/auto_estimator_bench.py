#!/usr/bin/env python3
"""transform_bcN_auto end to end with a REAL estimator -- zstd level 1 through the system libzstd (tests/cpp/zstd_estimator.c:
what the reference's estimator crate does; the reference quotes ~265 MiB/s for BC1 auto with it on one 9950X3D thread,
core/dxt-lossless-transform-bc1/src/transform/mod.rs:33-34) -- with the reference's sequence of estimator calls (1 thread)
and with dxtlt_set_auto_estimator_threads(N).  Texture-like data of ordinary compressibility: 64 MiB of blocks drawn at
random from the reference's 256x256 textures with the low two bits of both colour endpoints jittered (zstd -1 ratio
1.8 / 2.6 for BC1 / BC3; tiling the texture instead gives zstd whole-tile matches and GB/s).  Host memory in, host
memory out.  Prints MiB/s of input.   usage: python tools/auto_estimator_bench.py [threads ...]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import cabi  # noqa: E402
import dxt_lossless_transform_amd as pkg  # noqa: E402
from helpers import payload  # noqa: E402

threads_list = [int(a) for a in sys.argv[1:]] or [1, 4, 8, 16]
MIB = float(os.environ.get("AUTO_BENCH_MIB", "64"))            # input size per call
lib = cabi.bind(C.CDLL(pkg._lib.lib_path()))
made = cabi.zstd_c_estimator(1)
assert made is not None, "needs gcc and libzstd.so.1"
est, zest = made
out = {"estimator": "zstd level 1 (libzstd via tests/cpp/zstd_estimator.c)", "input_MiB": MIB, "MiB_per_s": {}}
for n in (1, 2, 3):
    fmt = f"bc{n}"
    block = 8 if n == 1 else 16
    tex = payload(fmt).reshape(-1, block)
    rng = np.random.default_rng(0xA070 + n)
    count = int(MIB * (1 << 20)) // block
    x = tex[rng.integers(0, tex.shape[0], count)].copy()
    colour = 0 if n == 1 else 8
    x[:, colour] ^= rng.integers(0, 4, count).astype(np.uint8)
    x[:, colour + 2] ^= rng.integers(0, 4, count).astype(np.uint8)
    x = x.reshape(-1)
    y = np.zeros_like(x)
    settings = {1: cabi.CoreSettings2, 2: cabi.CoreSettings2, 3: cabi.CoreSettings3}[n]()
    f = getattr(lib, f"dltbc{n}core_transform_auto")
    for use_all in (False, True):
        key = f"{fmt}_{'all' if use_all else 'fast'}"
        out["MiB_per_s"][key] = {}
        chosen = None
        for threads in threads_list:
            pkg.set_auto_estimator_threads(threads)
            f(x.ctypes.data, x.size, y.ctypes.data, y.size, C.byref(est), cabi.AutoSettings(use_all), C.byref(settings))
            reps = max(1, int(64 / MIB))
            zest.zest_reset()
            t = time.perf_counter()
            for _ in range(reps):
                r = f(x.ctypes.data, x.size, y.ctypes.data, y.size, C.byref(est), cabi.AutoSettings(use_all), C.byref(settings))
                assert r.ErrorCode == 0
            dt = (time.perf_counter() - t) / reps
            pick = bytes(settings)
            assert chosen is None or pick == chosen, "the choice must not depend on the thread count"
            chosen = pick
            out["MiB_per_s"][key][str(threads)] = {"MiB_per_s": round(x.size / dt / 2**20, 1), "estimator_calls": zest.zest_calls() // reps,
                                                  "max_concurrent_calls": zest.zest_max_concurrency()}
pkg.set_auto_estimator_threads(1)
print(json.dumps(out))
