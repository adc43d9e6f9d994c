"""BC1 WITHOUT the colour split through the BATCH kernel, forward (and inverse beside it): the corpus shape (2130 mip-chained
textures, every block count odd: halo + edge tiles) and regular arrays of equal buffers (aligned tiles through batch_kernel when the
array route is off, odd counts otherwise).  Run once per library (DXTLT_LIB_PATH) on ONE box: the shipped one plans 128-lane tiles
for this launch (batch_tile_threads), a side build with -DDXTLT_BATCH_BC1_NOSPLIT_FWD_THREADS=256 the round-5 shape.  Every case is
checked: exact round trip, and forward bytes against the oracle on three textures.
    python tools/batch_nosplit_probe.py [--settings 1,0 | 0,0 | 1,1] [--reps 10]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
import dxt_lossless_transform_amd as pkg
from dxt_lossless_transform_amd import batch
from oracle import oracle_c

ap = argparse.ArgumentParser()
ap.add_argument("--settings", default="1,0;0,0;1,1", help="variant,split_colour;...")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--more", action="store_true", help="more shapes: smaller buffers, the corpus by size class")
args = ap.parse_args()
dev = torch.device("cuda:0")
pkg.load()
print("library:", pkg._lib.lib_path(), flush=True)


def timed(fn, reps):
    bench.clock_warm(torch, fn, lambda: None, 60.0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def run(name, sizes_blocks, st, align=256):
    B = 8
    offs, at = [], 0
    for n in sizes_blocks:
        offs.append(at)
        at = (at + n * B + align - 1) // align * align
    x = torch.empty(at, dtype=torch.uint8, device=dev)
    pkg.fill_splitmix64(x, 0xB0A7, 0)
    y, z = torch.zeros_like(x), torch.zeros_like(x)
    views = [(x[o:o + n * B], y[o:o + n * B], z[o:o + n * B]) for n, o in zip(sizes_blocks, offs)]
    fwd = batch.prepare_batch([("bc1", False, a, b, st) for a, b, _ in views])
    inv = batch.prepare_batch([("bc1", True, b, c, st) for _, b, c in views])
    nbytes = sum(sizes_blocks) * B
    tf = timed(lambda: batch.run_prepared_batch(fwd), args.reps)
    ti = timed(lambda: batch.run_prepared_batch(inv), args.reps)
    ok = all(bool(torch.equal(c, a)) for a, _, c in views[:: max(1, len(views) // 64)])
    for i in (0, len(views) // 2, len(views) - 1):
        want = oracle_c.transform("bc1", views[i][0].cpu().numpy(), int(st.decorrelation_mode), st.split_colour_endpoints, True)
        ok = ok and bool(np.array_equal(views[i][1].cpu().numpy(), want))
    print(f"  {name:44s} fwd {2 * nbytes / tf / 8e12:.3f}  inv {2 * nbytes / ti / 8e12:.3f}  exact {ok}", flush=True)
    assert ok


for text in args.settings.split(";"):
    v, sc = (int(t) for t in text.split(","))
    st = pkg.Bc1TransformSettings(pkg.YCoCgVariant(v), bool(sc))
    print(f"settings variant {v} split_colour {sc}", flush=True)
    texs = bench.corpus_textures(1.0)
    run("corpus: 2130 mip-chained textures", [n for _, _, n in texs], st)
    run("64 x 16 MiB - 1 block", [(16 << 20) // 8 - 1] * 64, st)
    run("1024 x 1 MiB - 1 block", [(1 << 20) // 8 - 1] * 1024, st)
    run("1024 x 1 MiB (aligned: the array route)", [(1 << 20) // 8] * 1024, st)
    run("540 x 4096^2 with mips (1398101 blocks)", [1398101] * 540, st)
    run("mixed sizes, aligned bases (batch_kernel, form 1)", [(1 << 20) // 8, (2 << 20) // 8] * 512, st)
    if args.more:
        run("4096 x 256 KiB - 1 block", [(256 << 10) // 8 - 1] * 4096, st)
        run("2000 x 1024^2 with mips (87383 blocks)", [87383] * 2000, st)
        run("4000 x 512^2 with mips (21847 blocks)", [21847] * 4000, st)
        run("8000 x 256^2 with mips (5463 blocks)", [5463] * 8000, st)
        for lo, hi in ((0, 6000), (6000, 100000), (100000, 1 << 30)):
            part = [n for _, _, n in texs if lo <= n < hi]
            run(f"corpus textures of {lo}..{hi} blocks ({len(part)})", part * max(1, (1 << 27) // max(1, sum(part))), st)
