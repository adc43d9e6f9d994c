#!/usr/bin/env python3
"""bench.py -- BCn block-transform hot path on MI355X: GiB/s of BC blocks transformed (forward + inverse).

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload at every N: BASELINE.json configs[1] taken with the north-star's "forward+inverse" reading -- BC1, default
settings {YCoCg Variant1, split colour endpoints}, an 8 GiB block buffer per GPU (2^30 blocks) of splitmix64 random
blocks generated on the device.  One step = one forward transform of the buffer + one inverse transform of the
result, both through the C ABI of libdxtlt_gfx950.so, inputs resident in HBM.  `value` counts the block bytes fed to
each direction: (len + len) * steps * N / time.

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), used only for the barrier and the max-over-ranks
reduction of the timing.  The block array shards by contiguous range; every rank transforms its own 8 GiB shard
(weak scaling); there is no data-path collective (DESIGN.md "Multi-GPU").

The JSON line also carries
  roofline      for the dominant kernel (BC1 forward, fwd_tiled): algorithmic bytes = 16 B/block = 2*len per launch,
                divided by the launch's average duration measured with HIP events on the launch stream
  cpu_baseline  the C oracle (a port of the reference's scalar loops) timed on this box's host cores on a bounded
                sample of the same workload, rank 0 at N=1 only
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--size-gib", type=float, default=None,
                   help="block buffer per GPU (default: 8 GiB, BASELINE.json configs[1]/[2]; 4 GiB for --format bc7, configs[3])")
    p.add_argument("--format", default="bc1", choices=["bc1", "bc2", "bc3", "bc7"])
    p.add_argument("--workload", default="buffer", choices=["buffer", "archive"],
                   help="buffer: one block buffer per GPU (configs[1..3]); archive: BASELINE.json configs[4], alternating "
                        "256 MiB BC1 / BC3 textures, --size-gib per GPU (default 8: 64 GiB over 8 GPUs)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-sample-mib", type=int, default=1024)
    p.add_argument("--settings", default="", help="variant,split_alpha,split_colour (e.g. 0,0,1) instead of the "
                   "format's default settings; a sweep knob, the headline run uses the defaults")
    p.add_argument("--drop-blocks", type=int, default=0, help="experiment: shorten the buffer by this many blocks "
                   "(an odd count exercises the shifted-tile kernels)")
    p.add_argument("--force-path", type=int, default=0, help="experiment: 1 = element-granular kernel, 2 = shifted tiles")
    p.add_argument("--tile-threads", type=int, default=0, help="tuning experiment: tile workgroup size 256 (default) or 512")
    return p.parse_args()


def cpu_baseline(fmt: str, settings, sample_mib: int) -> dict:
    """Oracle timed on the host: single thread (comparable to the reference's per-core figures) and all cores."""
    import numpy as np

    from oracle import oracle_c

    nbytes = sample_mib << 20
    x = oracle_c.fill_splitmix64(nbytes, 0x0BC10002)
    y = np.zeros_like(x)
    z = np.zeros_like(x)
    v, sa, sc = settings
    cores = os.cpu_count() or 1

    def run(threads, reps):
        best = None
        for _ in range(reps):
            t0 = time.perf_counter()
            oracle_c.run_mt(fmt, x, y, v, sc, sa, False, threads)
            oracle_c.run_mt(fmt, y, z, v, sc, sa, True, threads)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return 2 * nbytes / best / 2**30

    run(cores, 1)  # touch all pages
    one = run(1, 3)
    allc = run(cores, 3)
    assert np.array_equal(z, x)
    out = {
        "value": round(one, 3), "unit": "GiB/s", "cores": 1, "kind": "port",
        "sample": f"{sample_mib} MiB of the same {fmt.upper()} splitmix64 workload, forward+inverse, best of 3, "
                  f"scalar C oracle (gcc -O3)",
        "all_cores_value": round(allc, 3), "all_cores": cores,
    }
    # Vectorised ports of the reference's AVX2 / AVX-512BW strategy exist for the headline settings (BC1, Variant1 +
    # split): when the host has AVX2 the widest one becomes the quoted figure (closest analogue of "the reference's SIMD path on one core").
    if fmt == "bc1" and (v, bool(sc)) == (1, True) and oracle_c.simd_available():
        def run_simd(threads, reps):
            best = None
            for _ in range(reps):
                t0 = time.perf_counter()
                oracle_c.run_bc1_default_simd(x, y, False, threads)
                oracle_c.run_bc1_default_simd(y, z, True, threads)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            return 2 * nbytes / best / 2**30

        z[:] = 0
        simd_one = run_simd(1, 3)
        simd_all = run_simd(cores, 3)
        assert np.array_equal(z, x)
        out.update({
            "scalar_value": out["value"], "scalar_all_cores_value": out["all_cores_value"],
            "value": round(simd_one, 3), "all_cores_value": round(simd_all, 3),
            "isa": oracle_c.SIMD_NAMES[oracle_c.simd_level()],
            "sample": f"{sample_mib} MiB of the same BC1 splitmix64 workload, forward+inverse, best of 3, "
                      f"{oracle_c.SIMD_NAMES[oracle_c.simd_level()]} port of the reference's SIMD strategy "
                      f"(oracle/dxtlt_oracle_avx2.c, gcc -O3; the widest level this host has); scalar_* = scalar C oracle",
        })
        if oracle_c.simd_level() == 5:   # also quote the AVX2 port on the same host
            oracle_c.simd_set_cap(2)
            try:
                out["avx2_value"] = round(run_simd(1, 3), 3)
            finally:
                oracle_c.simd_set_cap(5)
    return out


def bc7_main(args) -> None:
    """BASELINE.json configs[3]: BC7 forward (+ inverse) on a synthetic mode-mixed buffer.  Same JSON contract; the
    transform is this build's own format (docs/BC7_FORMAT.md; the reference has none), so parity is a round trip plus the
    build's own CPU restatement.  The `roofline` entry is the whole forward pipeline (histogram + scans + scatter)
    priced on the algorithmic 2 * len; the pipeline itself moves 3 * len."""
    import torch

    import dxt_lossless_transform_amd as pkg
    from dxt_lossless_transform_amd import bc7

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the library has no CPU fallback)"
    backend = os.environ.get("DXTLT_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    pkg.load()
    nbytes = int((args.size_gib if args.size_gib else 4.0) * (1 << 30))
    nbytes -= nbytes % (16 * 2048)
    blocks = nbytes // 16
    seed = 0x0BC70004
    x = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    pkg.fill_splitmix64(x, seed, rank * (nbytes // 8))
    # mode-mixed: mode m uniform in 0..7, low m+1 bits of byte 0 = 1 << m
    b = x.view(-1, 16)
    for lo in range(0, blocks, 1 << 26):
        v = b[lo:lo + (1 << 26)]
        m = (v[:, 15] & 7).to(torch.int32)
        low = ((2 << m) - 1).to(torch.uint8)
        v[:, 0] = (v[:, 0] & ~low) | (1 << m).to(torch.uint8)
        del m, low
    y, z = torch.empty_like(x), torch.empty_like(x)
    ws = torch.empty(bc7.workspace_bytes(nbytes), dtype=torch.uint8, device=dev)

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        bc7.transform_bc7(x, y, ws)
        bc7.untransform_bc7(y, z, ws)
    torch.cuda.synchronize()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        bc7.transform_bc7(x, y, ws)
        ev[k][1].record()
        bc7.untransform_bc7(y, z, ws)
        ev[k][2].record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    fwd_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps
    inv_ms = sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps
    ok = bool(torch.equal(z, x))
    cpu = None
    if rank == 0:
        import numpy as np

        from oracle import oracle_c

        # the `first` stream is byte 0 of every block in order; and a 64 MiB prefix against the CPU restatement
        ok = ok and bool(torch.equal(y[:blocks], x.view(-1, 16)[:, 0]))
        sample = 64 << 20
        small_y = torch.empty(sample, dtype=torch.uint8, device=dev)
        bc7.transform_bc7(x[:sample], small_y)
        xin = x[:sample].cpu().numpy()
        t1 = time.perf_counter()
        want = oracle_c.transform_bc7(xin)
        t2 = time.perf_counter()
        back = oracle_c.transform_bc7(want, inverse=True)
        t3 = time.perf_counter()
        ok = ok and bool(np.array_equal(small_y.cpu().numpy(), want)) and bool(np.array_equal(back, xin))
        cpu = {"value": round(2 * sample / (t3 - t1) / 2**30, 3), "unit": "GiB/s", "cores": 1, "kind": "port",
               "sample": "64 MiB of the same mode-mixed workload, forward+inverse, scalar C restatement of this build's "
                         "own BC7 format (oracle/dxtlt_oracle_bc7.c; the reference has no BC7 transform to time)",
               "fwd_value": round(sample / (t2 - t1) / 2**30, 3)}
    assert ok, "GPU result differs from the oracle / round trip failed"
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    achieved = 2 * nbytes / (fwd_ms * 1e-3) / 1e9
    achieved_inv = 2 * nbytes / (inv_ms * 1e-3) / 1e9
    out = {
        "metric": "GiB/s BC blocks transformed (fwd+inv)",
        "value": round(2 * nbytes * args.steps * world / elapsed / 2**30, 2),
        "unit": "GiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {
            "workload": f"BC7 mode-split forward+inverse (this build's own format, parity unpinned), {nbytes / 2**30:g} GiB "
                        "synthetic mode-mixed buffer per GPU, modes 0-7 uniform (BASELINE.json configs[3])",
            "format": "bc7", "blocks_per_gpu": blocks, "bytes_per_gpu": nbytes, "seed": hex(seed),
            "sharding": "independent buffer per rank, no collective",
            "bit_exact_roundtrip_and_oracle_prefix": ok,
            "fwd_ms": round(fwd_ms, 4), "inv_ms": round(inv_ms, 4),
            "fwd_GiBps": round(nbytes / (fwd_ms * 1e-3) / 2**30, 1), "inv_GiBps": round(nbytes / (inv_ms * 1e-3) / 2**30, 1),
        },
        "roofline": {
            "bound": "hbm", "kernel": "bc7 forward pipeline (bc7_hist_fwd + scans + bc7_scatter_fwd)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": None, "algorithmic_bytes_per_launch": 2 * nbytes, "pipeline_bytes": 3 * nbytes,
            "inverse_kernel": {"kernel": "bc7 inverse pipeline (bc7_hist_inv + scans + bc7_gather_inv)",
                               "achieved": round(achieved_inv, 1), "frac": round(achieved_inv / HBM_PEAK_GBPS, 4)},
        },
    }
    if world == 1 and not args.no_cpu_baseline and cpu is not None:
        out["cpu_baseline"] = cpu
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def archive_main(args) -> None:
    """BASELINE.json configs[4]: an archive of alternating 256 MiB BC1 and BC3 textures, each transformed with its
    format's default settings; the archive is split over the ranks by texture (contiguous ranges of whole textures: rank
    r holds textures [r*K, (r+1)*K)), no collective, the host places each rank's result at its offset.  A step transforms
    and restores every texture of the rank.  Verified per texture (exact round trip, an oracle window) and, since random
    blocks compress to ratio 1, the compression-ratio half of the config is checked on the reference's real 256x256 test
    textures: GPU output == CPU output byte for byte, so the ratios are equal, and both are reported."""
    import zlib

    import numpy as np
    import torch

    import dxt_lossless_transform_amd as pkg
    from oracle import oracle_c

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the library has no CPU fallback)"
    backend = os.environ.get("DXTLT_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    pkg.load()
    tex_bytes = 256 << 20
    per_gpu = int((args.size_gib if args.size_gib else 8.0) * (1 << 30))
    k = max(2, per_gpu // tex_bytes // 2 * 2)           # textures per rank, BC1 and BC3 alternating
    fmts = ["bc1" if i % 2 == 0 else "bc3" for i in range(k)]
    st = {"bc1": pkg.Bc1TransformSettings(), "bc3": pkg.Bc3TransformSettings()}
    fwd = {f: getattr(pkg, f"transform_{f}_with_settings") for f in st}
    inv = {f: getattr(pkg, f"untransform_{f}_with_settings") for f in st}
    xs = [torch.empty(tex_bytes, dtype=torch.uint8, device=dev) for _ in range(k)]
    ys = [torch.empty_like(x) for x in xs]
    zs = [torch.empty_like(x) for x in xs]
    for i, x in enumerate(xs):
        pkg.fill_splitmix64(x, 0x0A5C0005, (rank * k + i) * (tex_bytes // 8))   # one logical 64 GiB stream of blocks

    def barrier():
        if dist is not None:
            dist.barrier()

    def step():
        for i in range(k):
            fwd[fmts[i]](xs[i], ys[i], st[fmts[i]])
        for i in range(k):
            inv[fmts[i]](ys[i], zs[i], st[fmts[i]])

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ok = all(bool(torch.equal(z, x)) for x, z in zip(xs, zs))
    win = 1 << 15
    for i in (0, 1, k - 1):                                # one window per format and the rank's last texture
        f = fmts[i]
        B = pkg.BLOCK_BYTES[f]
        blocks = tex_bytes // B
        first = blocks // 3 + 17
        xin = xs[i][first * B:(first + win) * B].cpu().numpy()
        want = oracle_c.transform(f, xin, 1, True, True)
        got = np.empty_like(want)
        for off, w in pkg.stream_table(f, st[f]):
            got[off * win: off * win + w * win] = ys[i][off * blocks + w * first: off * blocks + w * (first + win)].cpu().numpy()
        ok = ok and bool(np.array_equal(got, want))
    ratios = {}
    if rank == 0:
        golden = os.path.join(ROOT, "tests", "golden")
        for f in ("bc1", "bc3"):
            # one real 256x256 texture as it is: tiling it would hand zlib repeats that the transform happens to line up
            tiled = np.fromfile(os.path.join(golden, f"r2-256-{f}.payload.bin"), dtype=np.uint8)
            d = torch.from_numpy(tiled).to(dev)
            o = torch.empty_like(d)
            fwd[f](d, o, st[f])
            gpu_out = o.cpu().numpy()
            cpu_out = oracle_c.transform(f, tiled, 1, True, True)
            ok = ok and bool(np.array_equal(gpu_out, cpu_out))
            ratios[f] = {"plain_zlib6": round(tiled.size / len(zlib.compress(tiled.tobytes(), 6)), 4),
                         "transformed_gpu_zlib6": round(tiled.size / len(zlib.compress(gpu_out.tobytes(), 6)), 4),
                         "transformed_cpu_zlib6": round(tiled.size / len(zlib.compress(cpu_out.tobytes(), 6)), 4)}
    assert ok, "GPU result differs from the oracle / round trip failed"
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    total = 2 * k * tex_bytes * args.steps * world
    out = {
        "metric": "GiB/s BC blocks transformed (fwd+inv)", "value": round(total / elapsed / 2**30, 2), "unit": "GiB/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {
            "workload": f"BC1+BC3 mixed archive, {k} alternating 256 MiB textures per GPU ({k * world * tex_bytes / 2**30:g} GiB "
                        "in all), default settings per format, split over the ranks by texture, no collective "
                        "(BASELINE.json configs[4])",
            "textures_per_gpu": k, "texture_bytes": tex_bytes, "seed": "0xa5c0005",
            "bit_exact_roundtrip_and_oracle_windows": ok,
            "zlib6_ratio_on_real_textures": ratios,
        },
        "roofline": {"bound": "hbm", "kernel": "fwd_tiled + inv_tiled over the archive", "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "achieved": round(2 * (2 * k * tex_bytes * args.steps) / elapsed / 1e9, 1),
                     "frac": round(2 * (2 * k * tex_bytes * args.steps) / elapsed / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                     "note": "wall clock of the whole step per GPU (launch gaps included), algorithmic 2 * bytes per direction"},
    }
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def main() -> None:
    args = parse_args()
    if args.workload == "archive":
        return archive_main(args)
    if args.format == "bc7":
        return bc7_main(args)
    if args.size_gib is None:
        args.size_gib = 8.0
    import torch

    import dxt_lossless_transform_amd as pkg

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py needs a GPU (the library has no CPU fallback)"
    # Rehearsal knob (never set by the driver): DXTLT_BENCH_BACKEND=gloo lets several ranks share one GPU so that the
    # N > 1 code path (rank-dependent data, barrier, MAX-reduce, rank-0 reporting) can be exercised on a 1-GPU box.
    backend = os.environ.get("DXTLT_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        dist = dist_mod
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    pkg.load()
    if args.tile_threads or args.force_path:
        pkg.set_tuning(args.tile_threads, args.force_path)
    fmt = args.format
    block = pkg.BLOCK_BYTES[fmt]
    settings = {"bc1": pkg.Bc1TransformSettings(), "bc2": pkg.Bc2TransformSettings(),
                "bc3": pkg.Bc3TransformSettings()}[fmt]
    if args.settings:
        v, sa, sc = (int(t) for t in args.settings.split(","))
        settings = {"bc1": lambda: pkg.Bc1TransformSettings(pkg.YCoCgVariant(v), bool(sc)),
                    "bc2": lambda: pkg.Bc2TransformSettings(pkg.YCoCgVariant(v), bool(sc)),
                    "bc3": lambda: pkg.Bc3TransformSettings(pkg.YCoCgVariant(v), bool(sa), bool(sc))}[fmt]()
    fwd = getattr(pkg, f"transform_{fmt}_with_settings")
    inv = getattr(pkg, f"untransform_{fmt}_with_settings")

    nbytes = int(args.size_gib * (1 << 30))
    nbytes -= nbytes % (block * 2048)
    nbytes -= args.drop_blocks * block
    blocks = nbytes // block
    seed = {"bc1": 0x0BC10002, "bc2": 0x0BC20002, "bc3": 0x0BC30003}[fmt]

    x = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    y = torch.empty_like(x)
    z = torch.empty_like(x)
    # rank r holds blocks [r*blocks, (r+1)*blocks) of one logical array
    pkg.fill_splitmix64(x, seed, rank * (nbytes // 8))
    torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()

    def step():
        fwd(x, y, settings)
        inv(y, z, settings)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()

    # per-kernel timing: HIP events on the stream the kernels are launched on (torch's current stream)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]

    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        fwd(x, y, settings)
        ev[k][1].record()
        inv(y, z, settings)
        ev[k][2].record()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    fwd_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps
    inv_ms = sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps

    # correctness inside the run: exact round trip, and one window against the oracle
    bit_exact = bool(torch.equal(z, x))
    if rank == 0:
        import numpy as np

        from oracle import oracle_c

        win = 1 << 16
        first = blocks // 2 + 4097
        xin = x[first * block:(first + win) * block].cpu().numpy()
        want = oracle_c.transform(fmt, xin, int(settings.decorrelation_mode), settings.split_colour_endpoints,
                                  getattr(settings, "split_alpha_endpoints", True))
        got = np.empty_like(want)
        for off, w in pkg.stream_table(fmt, settings):
            got[off * win: off * win + w * win] = y[off * blocks + w * first: off * blocks + w * (first + win)].cpu().numpy()
        bit_exact = bit_exact and bool(np.array_equal(got, want))
    assert bit_exact, "GPU result differs from the oracle / round trip failed"

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    total_in = 2 * nbytes * args.steps * world
    value = total_in / elapsed / 2**30
    achieved = 2 * nbytes / (fwd_ms * 1e-3) / 1e9  # algorithmic bytes: read len + write len
    achieved_inv = 2 * nbytes / (inv_ms * 1e-3) / 1e9

    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                rec = json.load(f)
            if rec.get("workload_bytes") == nbytes and rec.get("format") == fmt:
                traffic = rec.get("fwd_hbm_bytes_per_launch")
        except Exception:
            traffic = None

    out = {
        "metric": "GiB/s BC blocks transformed (fwd+inv)",
        "value": round(value, 2),
        "unit": "GiB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {
            "workload": f"{fmt.upper()} forward+inverse, "
                        + (f"settings {args.settings} (variant,split_alpha,split_colour), " if args.settings else
                           f"default settings ({'YCoCg Variant1, split colour endpoints' if fmt != 'bc3' else 'YCoCg Variant1, split alpha + colour endpoints'}), ")
                        + f"{nbytes / 2**30:g} GiB random block buffer per GPU (BASELINE.json configs[1])",
            "format": fmt, "blocks_per_gpu": blocks, "bytes_per_gpu": nbytes, "seed": hex(seed),
            "sharding": "contiguous block range per rank, no collective",
            "bit_exact_roundtrip_and_oracle_window": bit_exact,
            "fwd_ms": round(fwd_ms, 4), "inv_ms": round(inv_ms, 4),
            "fwd_GiBps": round(nbytes / (fwd_ms * 1e-3) / 2**30, 1), "inv_GiBps": round(nbytes / (inv_ms * 1e-3) / 2**30, 1),
        },
        "roofline": {
            "bound": "hbm", "kernel": f"fwd_tiled<{fmt}>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic,
            "algorithmic_bytes_per_launch": 2 * nbytes,
            "inverse_kernel": {"kernel": f"inv_tiled<{fmt}>", "achieved": round(achieved_inv, 1),
                               "frac": round(achieved_inv / HBM_PEAK_GBPS, 4)},
        },
    }
    if world == 1 and not args.no_cpu_baseline:
        s = (int(settings.decorrelation_mode), bool(getattr(settings, "split_alpha_endpoints", True)),
             bool(settings.split_colour_endpoints))
        out["cpu_baseline"] = cpu_baseline(fmt, s, args.cpu_sample_mib)
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
