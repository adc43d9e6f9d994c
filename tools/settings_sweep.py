#!/usr/bin/env python3
"""Every settings combination of BC1 (8), BC2 (8) and BC3 (16) on one box: forward / inverse fraction of the 8 TB/s HBM peak
on 2 * len, 4 GiB buffers, at a block count whose stream bases all sit on 128-byte lines (aligned tiles) and at that
count + 1 (every stream base off its line: forward halo tiles, inverse shifted tiles).  Exact round trip asserted.
Mirrors Bc3TransformSettings::all_combinations (bc3/src/transform/settings.rs:74) and its BC1 / BC2 twins."""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dxt_lossless_transform_amd as pkg  # noqa: E402

dev = torch.device("cuda:0")
GIB = float(os.environ.get("SWEEP_GIB", "4"))


def timed(fn, steps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps * 1e-3


names = {0: "None", 1: "Var1", 2: "Var2", 3: "Var3"}
print(f"{'format settings':34s} {'aligned fwd':>11s} {'inv':>6s}   {'2^k+1 fwd':>9s} {'inv':>6s}")
for fmt, B in (("bc1", 8), ("bc2", 16), ("bc3", 16)):
    base = int(GIB * (1 << 30)) // B
    buf = torch.empty((base + 1) * B, dtype=torch.uint8, device=dev)
    pkg.fill_splitmix64(buf[: (base + 1) * B // 8 * 8], 0x5EED + B)
    y = torch.empty_like(buf)
    z = torch.empty_like(buf)
    combos = [(v, sa, sc) for v in (1, 2, 3, 0) for sa in ((True, False) if fmt == "bc3" else (False,)) for sc in (True, False)]
    f = getattr(pkg, f"transform_{fmt}_with_settings")
    g = getattr(pkg, f"untransform_{fmt}_with_settings")
    for v, sa, sc in combos:
        st = {"bc1": lambda: pkg.Bc1TransformSettings(pkg.YCoCgVariant(v), sc),
              "bc2": lambda: pkg.Bc2TransformSettings(pkg.YCoCgVariant(v), sc),
              "bc3": lambda: pkg.Bc3TransformSettings(pkg.YCoCgVariant(v), sa, sc)}[fmt]()
        row = []
        for n in (base, base + 1):
            x, yy, zz = buf[: n * B], y[: n * B], z[: n * B]
            zz.zero_()
            tf = timed(lambda: f(x, yy, st))
            ti = timed(lambda: g(yy, zz, st))
            assert torch.equal(x, zz), (fmt, v, sa, sc, n)
            row += [2 * n * B / tf / 8e12, 2 * n * B / ti / 8e12]
        label = f"{fmt} {names[v]}" + (f" split_alpha={int(sa)}" if fmt == "bc3" else "") + f" split_colour={int(sc)}"
        print(f"{label:34s} {row[0]:11.3f} {row[1]:6.3f}   {row[2]:9.3f} {row[3]:6.3f}", flush=True)
    del buf, y, z
