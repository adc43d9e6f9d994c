"""Seeded random cases for the HOST-pointer entry points around the chunked pipeline's threshold (96 MiB): BC1-3 through
dxtlt_{un,}transform_bcN_with_settings and dxtlt_transform_sharded (1-3 shards), BC7 through dxtlt_{un,}transform_bc7
and the sharded form; sizes 60-220 MiB with ragged block counts.  BC1-3 against the multi-threaded oracle, BC7 against
the device-pointer path (itself checked against the oracle by the tests) plus exact round trips.
usage: python tools/fuzz_host.py [--cases 24] [--seed 1]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dxt_lossless_transform_amd as pkg  # noqa: E402
from dxt_lossless_transform_amd import bc7  # noqa: E402
from oracle import oracle_c  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=24)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
pkg.load()
B = {"bc1": 8, "bc2": 16, "bc3": 16, "bc7": 16}
for case in range(args.cases):
    rng = np.random.default_rng([args.seed, case])
    fmt = ("bc1", "bc2", "bc3", "bc7")[int(rng.integers(0, 4))]
    nbytes = int(rng.integers(60 << 20, 220 << 20))
    blocks = nbytes // B[fmt] + int(rng.integers(0, 3000))
    shards = int(rng.integers(0, 4))            # 0 = the single-buffer entry point
    x = oracle_c.fill_splitmix64(blocks * B[fmt], 0xF057 + case)
    y, z = np.zeros_like(x), np.zeros_like(x)
    tag = dict(case=case, fmt=fmt, blocks=blocks, shards=shards)
    if fmt == "bc7":
        if shards:
            bc7.transform_bc7_sharded(x, y, shards)
        else:
            bc7.transform_bc7(x, y)
        xd = torch.from_numpy(x).to("cuda:0"); yd = torch.empty_like(xd)
        bc7.transform_bc7(xd, yd)
        assert np.array_equal(y, yd.cpu().numpy()), tag
        del xd, yd
        if shards:
            bc7.transform_bc7_sharded(y, z, shards, inverse=True)
        else:
            bc7.untransform_bc7(y, z)
    else:
        v, sa, sc = int(rng.integers(0, 4)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        tag["settings"] = (v, sa, sc)
        st = (pkg.Bc3TransformSettings(pkg.YCoCgVariant(v), sa, sc) if fmt == "bc3" else
              (pkg.Bc1TransformSettings if fmt == "bc1" else pkg.Bc2TransformSettings)(pkg.YCoCgVariant(v), sc))
        if shards:
            pkg.transform_sharded(fmt, False, x, y, st, shards)
        else:
            getattr(pkg, f"transform_{fmt}_with_settings")(x, y, st)
        want = np.empty_like(x)
        oracle_c.run_mt(fmt, x, want, v, sc, sa, False, 8)
        assert np.array_equal(y, want), tag
        if shards:
            pkg.transform_sharded(fmt, True, y, z, st, shards)
        else:
            getattr(pkg, f"untransform_{fmt}_with_settings")(y, z, st)
    assert np.array_equal(z, x), tag
    print("ok", tag, flush=True)
print(f"done: {args.cases} host-path cases, all exact, seed {args.seed}")
