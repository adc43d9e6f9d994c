#!/usr/bin/env python3
"""bench.py -- BCn block-transform hot path on MI355X: GiB/s of BC blocks transformed (forward + inverse).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling weak|strong]

`python bench.py --gpus N` with N > 1 and no launcher environment starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same args>` as a
CHILD process before anything touches the GPU, relays rank 0's JSON line and exits with the child's code.  Under a
launcher (WORLD_SIZE set, as the driver does it) it is one rank of N.

Workload: BASELINE.json configs[1] taken with the north-star's "forward+inverse" reading -- BC1, default settings
{YCoCg Variant1, split colour endpoints}, splitmix64 random blocks generated on the device.  One step = one forward
transform + one inverse transform of the result, both through the C ABI of libdxtlt_gfx950.so, inputs resident in
HBM.  `value` counts the block bytes fed to each direction: (len + len) * steps / time.

  --scaling weak   (default) one logical array of N x 8 GiB; rank r holds blocks [r * 2^30, (r+1) * 2^30) -- 8 GiB per
                   GPU -- and transforms them as a stand-alone buffer (a shard's stand-alone result IS its slice of
                   every stream, packed: DESIGN.md "Multi-GPU").  JSON "scaling": "weak".
  --scaling strong one logical 8 GiB array whatever N; rank r owns the contiguous block range plan_shards() gives it and
                   calls dxtlt_transform_range_device on it: AoS slice in, its slice of every stream of the WHOLE
                   transformed buffer out (and back).  JSON "scaling": "strong".

Either way there is no data-path collective: torch.distributed (backend nccl = RCCL) carries the barrier and the
MAX-over-ranks of the elapsed time only.

The JSON line also carries
  roofline            for the dominant kernel (BC1 forward, fwd_tiled): algorithmic bytes = 16 B/block = 2*len per
                      launch, divided by the launch's average duration measured with HIP events on the launch stream
  cpu_baseline        the CPU port of the reference's algorithm timed on this box's host cores on a bounded sample of
                      the same workload, rank 0 at N=1 only
  sharded_host_array  north_star's "shard by contiguous range across the GPUs, concatenate on the host": ONE host
                      resident array, one call (dxtlt_transform_sharded) that splits it by block range over all N
                      devices and places every shard's stream slices at their final host offsets.  PCIe included,
                      reported beside `value`, never as `value`.
"""
from __future__ import annotations

import argparse
import contextlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


def self_launch_if_needed(args) -> None:
    """`python bench.py --gpus N` from a bare shell: become the parent of a torch.distributed.run job.  Runs before
    torch is imported; the parent never initialises the GPU and never exec()s (it waits for the child)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    for line in child.stdout:          # rank 0's JSON line goes to stdout as it comes; anything else a library printed, to stderr
        out = sys.stdout if line.lstrip().startswith("{") else sys.stderr
        out.write(line)
        out.flush()
    sys.exit(child.wait())


@contextlib.contextmanager
def stdout_to_stderr():
    """File descriptor 1 -> 2 for the duration: the gloo transport announces its connections on stdout from C++
    ("[Gloo] Rank 0 is connected to ..."), and this program's stdout carries ONE JSON line."""
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        sys.stdout.flush()
        os.dup2(saved, 1)
        os.close(saved)


class Ranks:
    """This process's place in the job: rank / world from the launcher's environment, its device, and the only two
    collectives the bench uses (barrier, MAX of a double)."""

    def __init__(self, args, need_gpu: bool = True):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if self.world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}")
        # Rehearsal knob (never set by the driver): DXTLT_BENCH_BACKEND=gloo lets several ranks share one GPU so that
        # the N > 1 code path (rank-dependent data, barrier, MAX-reduce, rank-0 reporting) runs on a 1-GPU box.
        self.backend = os.environ.get("DXTLT_BENCH_BACKEND", "nccl")
        self.dev = None
        self.dist = None
        self.cpu_group = None
        import torch

        self.torch = torch
        if need_gpu:
            assert torch.cuda.is_available(), "bench.py needs a GPU (the library has no CPU fallback)"
            n = torch.cuda.device_count()
            index = self.local_rank if self.backend == "nccl" else self.local_rank % n
            torch.cuda.set_device(index)
            self.dev = torch.device("cuda", index)
        # Rehearsal knob (never set by the driver): DXTLT_BENCH_FORCE_DIST=1 makes a one-rank job under a launcher go
        # through the process-group path too, so that the RCCL calls of the N > 1 path (init with device_id, the gloo
        # side group, barrier, MAX all-reduce of a double on the device) run on a 1-GPU box.
        force_dist = os.environ.get("DXTLT_BENCH_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ
        if self.world > 1 or force_dist:
            import torch.distributed as dist

            self.dist = dist
            with stdout_to_stderr():
                if self.backend == "nccl":
                    dist.init_process_group("nccl", device_id=self.dev)
                    # waiting for rank 0's host-side legs must not park a spinning RCCL kernel on every GPU
                    self.cpu_group = dist.new_group(backend="gloo")
                else:
                    dist.init_process_group(self.backend)
                    self.cpu_group = None
                dist.barrier(group=self.cpu_group)   # connections are made (and announced) here at the latest

    def barrier(self) -> None:
        if self.dist is not None:
            self.dist.barrier()

    def cpu_barrier(self) -> None:
        """A barrier that idles on the host (gloo), for waits that last seconds."""
        if self.dist is not None:
            self.dist.barrier(group=self.cpu_group)

    def max_over_ranks(self, seconds: float) -> float:
        if self.dist is None:
            return seconds
        on = self.dev if self.backend == "nccl" else "cpu"
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=on)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_over_ranks(self, values) -> list:
        """Every rank's list of floats, in rank order (all_gather; one rank: its own list).  The N > 1 line carries them as
        `per_rank`, so that a straggler shows in the record and not only in the MAX."""
        if self.dist is None:
            return [[float(v) for v in values]]
        on = self.dev if self.backend == "nccl" else "cpu"
        mine = self.torch.tensor([float(v) for v in values], dtype=self.torch.float64, device=on)
        out = [self.torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [[float(x) for x in t.tolist()] for t in out]

    def world_size_seen(self) -> int:
        """The world size the process group itself reports (RCCL's view under the nccl backend), not the environment's."""
        return int(self.dist.get_world_size()) if self.dist is not None else 1

    def finish(self) -> None:
        if self.dist is not None:
            self.dist.destroy_process_group()


def per_rank_record(R, fwd_ms, inv_ms, elapsed_s, flags=None) -> dict:
    """`world_size_seen` + `per_rank` of the JSON line: every rank's own HIP-event times, wall clock of the timed steps and the
    exactness flags of ITS OWN data (`flags`: name -> bool; the ranks' blocks differ, so each checks its own round trip and its own
    oracle window) -- a collective: every rank calls it.  With them the first record taken on more than one device shows a
    straggler, proves that the process group -- RCCL under the nccl backend -- saw N ranks, and verifies every rank's result."""
    names = sorted(flags or {})
    rows = R.gather_over_ranks([R.rank, -1.0 if fwd_ms is None else fwd_ms, -1.0 if inv_ms is None else inv_ms, elapsed_s]
                               + [1.0 if flags[n] else 0.0 for n in names])
    per_rank = []
    for r, f, i, e, *fl in rows:
        row = {"rank": int(r), "elapsed_s": round(e, 5)}
        if f >= 0:
            row.update({"fwd_ms": round(f, 4), "inv_ms": round(i, 4)})
        row.update({n: bool(v == 1.0) for n, v in zip(names, fl)})
        per_rank.append(row)
    return {"world_size_seen": R.world_size_seen(), "backend": R.backend if R.dist is not None else "none", "per_rank": per_rank}


def all_ranks_exact(per_rank: dict, names) -> bool:
    """The AND over every rank's row of every flag in `names` (a rank that did not report a flag counts as false)."""
    return all(row.get(n) is True for row in per_rank["per_rank"] for n in names)


BLOCK_BYTES = {"bc1": 8, "bc2": 16, "bc3": 16, "bc7": 16}


def job_shape(args, block: int, world: int, rank: int, plan_shards=None):
    """(total_blocks, first_block, blocks) of rank `rank`: --scaling weak = one stand-alone shard of --size-gib per GPU, the
    logical array is world x that (rank r holds blocks [r * blocks, (r + 1) * blocks)); strong = ONE array of --size-gib split by
    contiguous block range.  Shared by the real run and --rendezvous-only, so that the CPU tests of the launch shape check the
    very numbers a SCALE record will carry."""
    size_gib = 8.0 if args.size_gib is None else args.size_gib
    size_bytes = int(size_gib * (1 << 30))
    size_bytes -= size_bytes % (block * 2048)
    size_bytes -= args.drop_blocks * block
    if args.scaling == "strong":
        total_blocks = size_bytes // block
        if plan_shards is None:
            from dxt_lossless_transform_amd import plan_shards
        first, blocks = plan_shards(total_blocks, world)[rank]
        return total_blocks, first, blocks
    blocks = size_bytes // block
    return blocks * world, rank * blocks, blocks


def rendezvous_only(args) -> None:
    """--rendezvous-only: the launcher / rank plumbing without a GPU (CPU tests): rendezvous over gloo, barrier,
    MAX-reduce, rank 0 prints one JSON line -- with the keys that make an N > 1 record self-verifying (`world_size_seen`, one
    `per_rank` row per rank, all-gathered) and the block ranges the real run would give every rank (`config`, `ranges`)."""
    os.environ.setdefault("DXTLT_BENCH_BACKEND", "gloo")
    R = Ranks(args, need_gpu=False)
    R.barrier()
    worst = R.max_over_ranks(float(R.rank + 1))
    per_rank = R.gather_over_ranks([float(R.rank + 1), float(10 * R.rank)])
    fmt = args.format if args.workload == "buffer" else "bc1"
    total, first, blocks = job_shape(args, BLOCK_BYTES[fmt], R.world, R.rank)
    ranges = R.gather_over_ranks([float(R.rank), float(first), float(blocks)])
    seen = R.world_size_seen()
    # the per-rank verification of a real N > 1 line, rehearsed without a device: every rank takes a window of ITS OWN block range
    # (the same place and the same rank-dependent splitmix64 offset as main()), has two CPU statements of the transform agree on it
    # (here they stand in for the HIP path and the oracle), and the flags travel through the very per_rank_record / all_ranks_exact
    # the real run uses
    flags_rows, all_exact = None, None
    if fmt != "bc7":
        import numpy as np

        if R.rank == 0:
            from oracle import oracle_c

            oracle_c.lib()
        R.cpu_barrier()
        from oracle import oracle_c, oracle_np

        block = BLOCK_BYTES[fmt]
        seed = {"bc1": 0x0BC10002, "bc2": 0x0BC20002, "bc3": 0x0BC30003}[fmt]
        win = min(1 << 12, blocks)
        lf = min(blocks // 2 + 4097, blocks - win)
        xin = oracle_c.fill_splitmix64(win * block, seed, (first + lf) * block // 8)
        a = oracle_c.transform(fmt, xin, 1, True, True)
        b = oracle_np.transform(fmt, xin, 1, True, True)
        back = oracle_c.transform(fmt, a, 1, True, True, inverse=True)
        rec = per_rank_record(R, None, None, 0.0, {"round_trip_exact": bool(np.array_equal(back, xin)),
                                                   "oracle_window_exact": bool(np.array_equal(a, b))})
        flags_rows = rec["per_rank"]
        all_exact = all_ranks_exact(rec, ("round_trip_exact", "oracle_window_exact"))
    if R.rank == 0:
        print(json.dumps({"rendezvous": "ok", "n_gpus": R.world, "max_over_ranks": worst, "backend": R.backend,
                          "world_size_seen": seen, "per_rank": per_rank, "scaling": args.scaling,
                          "config": {"format": fmt, "total_blocks": total, "blocks_per_gpu": blocks},
                          "ranges": [[int(r), int(f), int(b)] for r, f, b in ranges],
                          "per_rank_flags": flags_rows, "bit_exact_roundtrip_and_oracle_window": all_exact}), flush=True)
    R.finish()


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
    p.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                   help="weak: --size-gib per GPU (default); strong: --size-gib in all, split by contiguous block range")
    p.add_argument("--host-array-gib", type=float, default=None,
                   help="size of the host-resident array of the sharded_host_array leg (default: --size-gib for "
                        "bc1/bc2/bc3 buffers; 0 = skip)")
    p.add_argument("--archive-split", default="texture", choices=["texture", "range"],
                   help="--workload archive: 'texture' = every rank owns a contiguous byte range of the archive (whole "
                        "textures); 'range' = every texture's block range is cut over the ranks (range calls)")
    p.add_argument("--rendezvous-only", action="store_true", help=argparse.SUPPRESS)
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--small-legs", action="store_true", help=argparse.SUPPRESS)   # CLI test: legs on a sub-4-GiB buffer
    p.add_argument("--leg-steps", type=int, default=10,
                   help="timed steps of each extra leg (BC3 8 GiB, BC7 4 GiB uniform + skewed, archive slice) that the "
                        "default single-GPU BC1 run reports under `legs`; 0 = no legs")
    p.add_argument("--cpu-sample-mib", type=int, default=1024)
    p.add_argument("--settings", default="", help="variant,split_alpha,split_colour (e.g. 0,0,1) instead of the "
                   "format's default settings; a sweep knob, the headline run uses the defaults")
    p.add_argument("--drop-blocks", type=int, default=0, help="experiment: shorten the buffer by this many blocks "
                   "(an odd count exercises the shifted-tile kernels)")
    p.add_argument("--force-path", type=int, default=0, help="experiment: 1 = element-granular kernel, 2 = shifted tiles")
    p.add_argument("--tile-threads", type=int, default=0, help="tuning experiment: tile workgroup size 256 (default) or 512")
    return p.parse_args()


def physical_cores() -> int:
    """Physical cores of this host (distinct (package, core) pairs in /proc/cpuinfo); SMT threads are not cores.
    Falls back to os.cpu_count() where the file does not say."""
    pairs, phys, core = set(), None, None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        pairs.add((phys, core))
                    phys = core = None
        if phys is not None and core is not None:
            pairs.add((phys, core))
    except OSError:
        pass
    return len(pairs) or (os.cpu_count() or 1)


def cpu_baseline(fmt: str, settings, sample_mib: int) -> dict:
    """Oracle timed on the host: single thread (comparable to the reference's per-core figures) and one thread per
    physical core."""
    import numpy as np

    from oracle import oracle_c

    nbytes = sample_mib << 20
    x = oracle_c.fill_splitmix64(nbytes, 0x0BC10002)
    y = np.zeros_like(x)
    z = np.zeros_like(x)
    v, sa, sc = settings
    cores = physical_cores()

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
        "all_cores_value": round(allc, 3), "all_cores": cores, "host_threads": os.cpu_count() or 1,
        "note": "`value` = scalar port of the reference's loops" + (
            "; for BC3 with split alphas + split colours + decorrelation (the default settings) the reference itself "
            "dispatches to its scalar loop (bc3 with_split_alphas_colour_and_recorr/transform/mod.rs:31), so this leg matches "
            "upstream's own path" if fmt == "bc3" and (v, bool(sa), bool(sc)) == (1, True, True) else
            "; the reference has vectorised paths for these settings that this leg does not reproduce"),
    }
    # AVX2 ports exist for BC2 default {Variant1, split colours} and BC3 "standard" {None, no splits} too
    simd23 = {("bc2", 1, True): 2, ("bc3", 0, False): 3}.get((fmt, v, bool(sc))) if not (fmt == "bc3" and sa) else None
    if simd23 and oracle_c.simd_level() >= 2:
        def run23(threads, reps):
            best = None
            for _ in range(reps):
                t0 = time.perf_counter()
                oracle_c.run_bc23_simd(simd23, x, y, False, threads)
                oracle_c.run_bc23_simd(simd23, y, z, True, threads)
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            return 2 * nbytes / best / 2**30

        z[:] = 0
        one23, all23 = run23(1, 3), run23(cores, 3)
        assert np.array_equal(z, x)
        out.update({
            "scalar_value": out["value"], "scalar_all_cores_value": out["all_cores_value"],
            "value": round(one23, 3), "all_cores_value": round(all23, 3), "isa": "AVX2",
            "sample": f"{sample_mib} MiB of the same {fmt.upper()} splitmix64 workload, forward+inverse, best of 3, AVX2 port of "
                      "the reference's SIMD strategy (oracle/dxtlt_oracle_avx2.c, gcc -O3); scalar_* = scalar C oracle",
            "note": "`value` = AVX2 port of the reference's vectorised path for these settings, pinned to the scalar oracle",
        })
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
            "note": f"`value` = {oracle_c.SIMD_NAMES[oracle_c.simd_level()]} port of the reference's vectorised path for these "
                    "settings (with_split_colour_and_recorr), pinned to the scalar oracle byte for byte; all_cores = one "
                    "thread per physical core",
        })
        if oracle_c.simd_level() == 5:   # also quote the AVX2 port on the same host
            oracle_c.simd_set_cap(2)
            try:
                out["avx2_value"] = round(run_simd(1, 3), 3)
            finally:
                oracle_c.simd_set_cap(5)
    return out


def bc7_force_modes_device(torch, x, mix: str) -> list:
    """Mode-mixed BC7 data in place: for mode m the low m + 1 bits of byte 0 become 1 << m.
    `uniform`: m uniform in 0..7; `skewed`: a texture-like histogram, mode 6 > 1 > 3 > the rest.  Returns the mode counts."""
    b = x.view(-1, 16)
    counts = torch.zeros(9, dtype=torch.int64, device=x.device)
    for lo in range(0, b.shape[0], 1 << 26):
        v = b[lo:lo + (1 << 26)]
        r = v[:, 15].to(torch.int32)
        if mix == "uniform":
            m = r & 7
        else:
            m = torch.where(r < 140, 6, torch.where(r < 200, 1, torch.where(r < 230, 3, r & 7))).to(torch.int32)
        low = ((2 << m) - 1).to(torch.uint8)
        v[:, 0] = (v[:, 0] & ~low) | (1 << m).to(torch.uint8)
        counts += torch.bincount(m, minlength=9)
        del r, m, low
    return counts.tolist()


LEG_WARM_MS = 100.0


def clock_warm(torch, fwd, inv, ms: float = LEG_WARM_MS) -> int:
    """Untimed (fwd, inv) pairs, back to back, for `ms` of wall time; returns how many.  A leg starts after an idle phase (the
    checks of the leg before it run on the CPU), and after idling the chip takes ~40 ms of load to reach its steady clocks:
    the BC7 kernels, close to the vector-issue bound, run at 0.5-0.7 of peak for their first dozen launches and at 0.75-0.80
    afterwards, the BC1 kernels need four (tools/clock_ramp_probe.py, profiles/r03_clock_ramp.txt).  Warming up by launch
    count (two pairs) measured that ramp, not the kernels."""
    n = 0
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(4):
            fwd()
            inv()
        n += 4
        torch.cuda.synchronize()
    return n


def timed_pair(torch, fwd, inv, steps: int, warmup: int):
    """`steps` x (fwd, inv) on torch's current stream -- the stream the C ABI is handed -- with HIP events around each
    half, after LEG_WARM_MS of untimed pairs (clock_warm) and `warmup` more.  Returns (fwd_ms, inv_ms, wall_s)."""
    clock_warm(torch, fwd, inv)
    for _ in range(warmup):
        fwd()
        inv()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        ev[k][0].record()
        fwd()
        ev[k][1].record()
        inv()
        ev[k][2].record()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return (sum(e[0].elapsed_time(e[1]) for e in ev) / steps, sum(e[1].elapsed_time(e[2]) for e in ev) / steps, wall)


def leg_record(workload: str, nbytes: int, fwd_ms: float, inv_ms: float, wall_s: float, steps: int, fwd_kernel: str,
               inv_kernel: str, checks: dict, launches_per_direction: int = 1) -> dict:
    """One entry of `legs`: the same quantities as the headline, for one more BASELINE.json configuration."""
    fa = 2 * nbytes / (fwd_ms * 1e-3) / 1e9
    ia = 2 * nbytes / (inv_ms * 1e-3) / 1e9
    return {
        "workload": workload, "bytes": nbytes, "steps": steps, "warmup_ms": LEG_WARM_MS,
        "value": round(2 * nbytes * steps / wall_s / 2**30, 2), "unit": "GiB/s (fwd+inv, wall clock of the timed steps)",
        "fwd_ms": round(fwd_ms, 4), "inv_ms": round(inv_ms, 4),
        "fwd_GiBps": round(nbytes / (fwd_ms * 1e-3) / 2**30, 1), "inv_GiBps": round(nbytes / (inv_ms * 1e-3) / 2**30, 1),
        "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s", "kernel": fwd_kernel,
                     "achieved": round(fa, 1), "frac": round(fa / HBM_PEAK_GBPS, 4),
                     "algorithmic_bytes_per_direction": 2 * nbytes, "launches_per_direction": launches_per_direction,
                     "inverse_kernel": {"kernel": inv_kernel, "achieved": round(ia, 1), "frac": round(ia / HBM_PEAK_GBPS, 4)}},
        **checks,
    }


def leg_cpu_baseline(fmt: str, sample, mode_mix=None) -> dict:
    """The CPU side of a leg on a bounded sample (about a second of work): the oracle's scalar C loops on one core and,
    for BC3, on every physical core.  For BC3 with the default settings the reference itself dispatches to its scalar loop
    (bc3 with_split_alphas_colour_and_recorr/transform/mod.rs:31); BC7 has no reference implementation at all -- what is
    timed is the CPU statement of this build's own format."""
    import numpy as np

    from oracle import oracle_c

    if fmt == "bc7":
        t0 = time.perf_counter()
        fwd = oracle_c.transform_bc7(sample)
        t1 = time.perf_counter()
        back = oracle_c.transform_bc7(fwd, inverse=True)
        t2 = time.perf_counter()
        assert np.array_equal(back, sample)
        return {"value": round(2 * sample.size / (t2 - t0) / 2**30, 3), "unit": "GiB/s", "cores": 1, "kind": "port",
                "sample": f"{sample.size >> 20} MiB of the leg's {mode_mix} mode mix, forward+inverse, scalar C statement of this "
                          "build's own BC7 format (oracle/dxtlt_oracle_bc7.c; the reference has no BC7 transform to time)",
                "fwd_value": round(sample.size / (t1 - t0) / 2**30, 3)}
    y, z = np.zeros_like(sample), np.zeros_like(sample)
    cores = physical_cores()

    def run(threads):
        best = None
        for _ in range(3):
            t0 = time.perf_counter()
            oracle_c.run_mt(fmt, sample, y, 1, True, True, False, threads)
            oracle_c.run_mt(fmt, y, z, 1, True, True, True, threads)
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        return round(2 * sample.size / best / 2**30, 3)

    run(cores)
    one, allc = run(1), run(cores)
    assert np.array_equal(z, sample)
    note = ("the reference dispatches BC3 with split alphas + split colours + decorrelation to its scalar loop "
            "(bc3 with_split_alphas_colour_and_recorr/transform/mod.rs:31): this leg is upstream's own path") if fmt == "bc3" else (
            "scalar port of the reference's loops; the reference has vectorised paths for these settings that this figure does "
            "not reproduce (the headline's cpu_baseline carries the AVX-512 / AVX2 ports)")
    return {"value": one, "unit": "GiB/s", "cores": 1, "kind": "port", "all_cores_value": allc, "all_cores": cores,
            "sample": f"{sample.size >> 20} MiB of the leg's blocks, forward+inverse, best of 3, scalar C oracle (gcc -O3)",
            "note": note}


def run_legs(pkg, torch, dev, x, y, z, steps: int, warmup: int, cpu: bool = True) -> dict:
    """The other single-GPU configurations of BASELINE.json in the same run, on the headline's three device buffers:
    configs[2] BC3 default fwd+inv over 8 GiB, configs[3] BC7 fwd+inv over 4 GiB (uniform mode mix and a texture-like
    skewed one), and one GPU's share of configs[4] (8 GiB of alternating 256 MiB BC1 / BC3 textures).  Every leg: HIP
    events per direction, exact round trip, a window of the forward output against the CPU oracle.  Legs never enter
    `value`."""
    import numpy as np

    from dxt_lossless_transform_amd import bc7
    from oracle import oracle_c

    legs = {}
    cap = int(x.numel())

    # configs[2]: BC3, default settings
    n3 = min(cap, 8 << 30)
    n3 -= n3 % (16 * 2048)
    st3 = pkg.Bc3TransformSettings()
    x3, y3, z3 = x[:n3], y[:n3], z[:n3]
    pkg.fill_splitmix64(x3, 0x0BC30003, 0)
    z3.zero_()
    f_ms, i_ms, wall = timed_pair(torch, lambda: pkg.transform_bc3_with_settings(x3, y3, st3),
                                  lambda: pkg.untransform_bc3_with_settings(y3, z3, st3), steps, warmup)
    blocks = n3 // 16
    win = 1 << 16
    lf = min(blocks // 2 + 4097, blocks - win)
    want = oracle_c.transform("bc3", x3[lf * 16:(lf + win) * 16].cpu().numpy(), 1, True, True)
    got = np.empty_like(want)
    for off, w in pkg.stream_table("bc3", st3):
        lo = off * blocks + w * lf
        got[off * win: off * win + w * win] = y3[lo: lo + w * win].cpu().numpy()
    legs["bc3"] = leg_record(
        f"BC3 forward+inverse, default settings (YCoCg Variant1, split alpha + colour endpoints), {n3 / 2**30:g} GiB random "
        "block buffer (BASELINE.json configs[2])", n3, f_ms, i_ms, wall, steps, "fwd_tiled<bc3>", "inv_tiled<bc3>",
        {"bit_exact_roundtrip": bool(torch.equal(z3, x3)), "oracle_window_exact": bool(np.array_equal(got, want))})
    if cpu:
        legs["bc3"]["cpu_baseline"] = leg_cpu_baseline("bc3", x3[: min(n3, 256 << 20)].cpu().numpy())

    # BC2 (north_star names it beside BC1 / BC3; bc2 transform_with_settings.rs:30,93): default settings, same shape as the BC3 leg
    st2 = pkg.Bc2TransformSettings()
    pkg.fill_splitmix64(x3, 0x0BC20002, 0)
    z3.zero_()
    f_ms, i_ms, wall = timed_pair(torch, lambda: pkg.transform_bc2_with_settings(x3, y3, st2),
                                  lambda: pkg.untransform_bc2_with_settings(y3, z3, st2), steps, warmup)
    want = oracle_c.transform("bc2", x3[lf * 16:(lf + win) * 16].cpu().numpy(), 1, True, False)
    got = np.empty_like(want)
    for off, w in pkg.stream_table("bc2", st2):
        lo = off * blocks + w * lf
        got[off * win: off * win + w * win] = y3[lo: lo + w * win].cpu().numpy()
    legs["bc2"] = leg_record(
        f"BC2 forward+inverse, default settings (YCoCg Variant1, split colour endpoints), {n3 / 2**30:g} GiB random block buffer",
        n3, f_ms, i_ms, wall, steps, "fwd_tiled<bc2>", "inv_tiled<bc2>",
        {"bit_exact_roundtrip": bool(torch.equal(z3, x3)), "oracle_window_exact": bool(np.array_equal(got, want))})
    if cpu:
        legs["bc2"]["cpu_baseline"] = leg_cpu_baseline("bc2", x3[: min(n3, 256 << 20)].cpu().numpy())

    # configs[3]: BC7, this build's own format; two mode mixes
    n7 = min(cap, 4 << 30)
    n7 -= n7 % (16 * 2048)
    x7, y7, z7 = x[:n7], y[:n7], z[:n7]
    for mix in ("uniform", "skewed"):
        pkg.fill_splitmix64(x7, 0x0BC70004, 0)
        counts = bc7_force_modes_device(torch, x7, mix)
        z7.zero_()
        f_ms, i_ms, wall = timed_pair(torch, lambda: bc7.transform_bc7(x7, y7), lambda: bc7.untransform_bc7(y7, z7),
                                      steps, warmup)
        sample = 16 << 20                       # whole granules: the prefix's streams are a transform of their own
        small = torch.empty(sample, dtype=torch.uint8, device=dev)
        bc7.transform_bc7(x7[:sample], small)
        xin = x7[:sample].cpu().numpy()
        want = oracle_c.transform_bc7(xin)
        legs[f"bc7_{mix}"] = leg_record(
            f"BC7 granule-sorted field split v2 (this build's own format, parity unpinned), forward+inverse, {n7 / 2**30:g} GiB "
            f"synthetic mode-mixed buffer, modes {'0-7 uniform' if mix == 'uniform' else 'skewed like a texture: 6 > 1 > 3 > rest'} "
            "(BASELINE.json configs[3])", n7, f_ms, i_ms, wall, steps, "bc7_forward", "bc7_inverse",
            {"bit_exact_roundtrip": bool(torch.equal(z7, x7)),
             "oracle_prefix_exact": bool(np.array_equal(small.cpu().numpy(), want)), "mode_counts": counts})
        if cpu:
            legs[f"bc7_{mix}"]["cpu_baseline"] = leg_cpu_baseline("bc7", xin, mix)
        del small

    # configs[4], one GPU's share: alternating 256 MiB BC1 / BC3 textures, each with its format's default settings
    tex = min(256 << 20, cap // 2 // (16 * 2048) * (16 * 2048))   # 256 MiB; smaller only in the small-buffer CLI test
    k = max(2, min(cap, 8 << 30) // tex // 2 * 2)
    fmts = ["bc1" if i % 2 == 0 else "bc3" for i in range(k)]
    st = {"bc1": pkg.Bc1TransformSettings(), "bc3": st3}
    fwd = {f: getattr(pkg, f"transform_{f}_with_settings") for f in st}
    inv = {f: getattr(pkg, f"untransform_{f}_with_settings") for f in st}
    xa, ya, za = x[:k * tex], y[:k * tex], z[:k * tex]
    pkg.fill_splitmix64(xa, 0x0A5C0005, 0)
    za.zero_()
    xs, ys, zs = (list(t.view(k, tex).unbind(0)) for t in (xa, ya, za))

    def a_fwd():
        for i in range(k):
            fwd[fmts[i]](xs[i], ys[i], st[fmts[i]])

    def a_inv():
        for i in range(k):
            inv[fmts[i]](ys[i], zs[i], st[fmts[i]])

    f_ms, i_ms, wall = timed_pair(torch, a_fwd, a_inv, steps, warmup)
    ok = True
    win = 1 << 15
    for i in (0, 1, k - 1):
        f = fmts[i]
        B = pkg.BLOCK_BYTES[f]
        blocks = tex // B
        lf = blocks // 3 + 17
        want = oracle_c.transform(f, xs[i][lf * B:(lf + win) * B].cpu().numpy(), 1, True, True)
        got = np.empty_like(want)
        for off, w in pkg.stream_table(f, st[f]):
            got[off * win: off * win + w * win] = ys[i][off * blocks + w * lf: off * blocks + w * (lf + win)].cpu().numpy()
        ok = ok and bool(np.array_equal(got, want))
    legs["archive"] = leg_record(
        f"BC1+BC3 mixed archive, one GPU's share of BASELINE.json configs[4]: {k * tex / 2**30:g} GiB = {k} alternating {tex >> 20} MiB "
        f"textures, default settings per format, one call per texture and direction", k * tex, f_ms, i_ms, wall, steps,
        "fwd_tiled<bc1> + fwd_tiled<bc3>", "inv_tiled<bc1> + inv_tiled<bc3>",
        {"bit_exact_roundtrip": bool(torch.equal(za, xa)), "oracle_windows_exact": ok}, launches_per_direction=k)
    # HBM traffic per launch from the committed PMC passes (profiles/pmc_traffic.json), where the leg ran at the profiled size
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            rec = json.load(f)
        lt, src = rec.get("legs", {}), rec.get("source", "profiles/pmc_traffic.json")
        note = f"{src}: committed rocprofv3 --pmc passes (FETCH_SIZE x2 per the gfx950 rule, WRITE_SIZE); NOT measured by this run"
        if lt.get("bc3", {}).get("bytes") == n3 and lt["bc3"].get("fwd"):
            legs["bc3"]["roofline"].update({"traffic": lt["bc3"]["fwd"], "inverse_traffic": lt["bc3"]["inv"], "traffic_source": note})
        if lt.get("bc2", {}).get("bytes") == n3 and lt["bc2"].get("fwd"):
            legs["bc2"]["roofline"].update({"traffic": lt["bc2"]["fwd"], "inverse_traffic": lt["bc2"]["inv"], "traffic_source": note})
        for mix in ("uniform", "skewed"):
            # a traffic figure rides only in the leg whose mode mix was profiled ("bc7" = the uniform mix of earlier rounds)
            rec7 = lt.get(f"bc7_{mix}") or (lt.get("bc7") if mix == "uniform" else None) or {}
            if rec7.get("bytes") == n7 and rec7.get("fwd"):
                legs[f"bc7_{mix}"]["roofline"].update({"traffic": rec7["fwd"], "inverse_traffic": rec7["inv"],
                                                       "traffic_source": note + f" ({mix} mix)"})
        at = lt.get("archive_texture", {})
        if at.get("bytes") == tex and all(at.get(x) for x in ("bc1_fwd", "bc3_fwd", "bc1_inv", "bc3_inv")):
            legs["archive"]["roofline"].update({"traffic": (at["bc1_fwd"] + at["bc3_fwd"]) * (k // 2),
                                                "inverse_traffic": (at["bc1_inv"] + at["bc3_inv"]) * (k // 2),
                                                "traffic_source": note + "; per-texture figures x the textures of one direction"})
    except (OSError, ValueError, KeyError):
        pass
    for name, leg in legs.items():
        bad = [c for c, v in leg.items() if c.endswith("exact") or c.startswith("bit_exact") if v is not True]
        assert not bad, f"leg {name}: {bad} failed"
    return legs


# ---- the corpus-shaped leg ---------------------------------------------------------------------------------------------
# The reference's own published benchmark is not one big buffer: it is "2130 real files (8692.9 MiB)" of BC1 DDS textures
# (api/dxt-lossless-transform-bc1-api/README.MD:286-311), transformed file after file.  A DDS texture with a full mip chain
# has an ODD block count (... + 4 + 1 + 1 + 1 blocks for the 8x8, 4x4, 2x2 and 1x1 levels), so every stream base of its
# transformed buffer is off its 16-byte / 128-byte boundary: the halo / shifted tile forms and ragged tails, not the aligned
# tiles the 2^k-byte legs run.  This leg rebuilds that shape synthetically: 2130 textures with full mip chains, dimensions
# 256..4096, 8693 MiB as BC1, resident in HBM side by side (each at the next 256-byte boundary, what an allocator hands out),
# ONE dxtlt_transform_batch_device call per direction.
CORPUS_CLASSES = [  # (width, height, count): 2130 textures, 8693 MiB as BC1 with full mip chains
    (4096, 4096, 540), (4096, 2048, 200), (2048, 2048, 500), (2048, 1024, 200), (1024, 1024, 300), (1024, 512, 100),
    (512, 512, 150), (512, 256, 50), (256, 256, 90),
]


def mip_chain_blocks(width: int, height: int) -> int:
    """4x4 blocks of a texture with its full mip chain (every level down to 1x1, each at least one block)."""
    n = 0
    while True:
        n += ((width + 3) // 4) * ((height + 3) // 4)
        if width == 1 and height == 1:
            return n
        width, height = max(1, width // 2), max(1, height // 2)


def corpus_textures(count_scale: float = 1.0, seed: int = 0xC0A9005) -> list:
    """[(width, height, blocks)] of the corpus in a fixed shuffled order (a directory walk does not sort by size)."""
    import random

    texs = []
    for w, h, c in CORPUS_CLASSES:
        texs += [(w, h, mip_chain_blocks(w, h))] * max(1, int(round(c * count_scale)))
    random.Random(seed).shuffle(texs)
    # experiment knobs (tools/corpus_probe.py): the textures in size order; only textures of at least so many blocks
    if os.environ.get("DXTLT_CORPUS_SORT") == "1":
        texs.sort(key=lambda t: t[2])
    least = int(os.environ.get("DXTLT_CORPUS_MIN_BLOCKS", "0"))
    return [t for t in texs if t[2] >= least]


def corpus_layout(texs, block_bytes: int, align: int = 256):
    """Byte offset of every texture in an arena where each starts at the next `align`-byte boundary; (offsets, arena bytes)."""
    offs, at = [], 0
    for _, _, blocks in texs:
        offs.append(at)
        at = (at + blocks * block_bytes + align - 1) // align * align
    return offs, at


def run_corpus_leg(pkg, torch, dev, fmt: str, steps: int, warmup: int, count_scale: float = 1.0, cpu: bool = True,
                   align: int = 256) -> dict:
    """One corpus-shaped leg (see CORPUS_CLASSES): device-resident textures, one batch call per direction, HIP events per
    direction, exact round trip over the whole arena (gaps between textures included, in both outputs), oracle equality on
    every 16th texture in size order plus the smallest, a middle one and the largest."""
    import numpy as np

    from dxt_lossless_transform_amd import batch
    from oracle import oracle_c

    B = pkg.BLOCK_BYTES[fmt]
    st = {"bc1": pkg.Bc1TransformSettings, "bc2": pkg.Bc2TransformSettings, "bc3": pkg.Bc3TransformSettings}[fmt]()
    texs = corpus_textures(count_scale)
    if fmt != "bc1":
        texs = texs[::2]                     # 16-byte blocks: every second texture, the same ~8.5 GiB
    offs, arena = corpus_layout(texs, B, align)
    x = torch.empty(arena, dtype=torch.uint8, device=dev)
    y = torch.zeros(arena, dtype=torch.uint8, device=dev)
    z = torch.zeros(arena, dtype=torch.uint8, device=dev)
    pkg.fill_splitmix64(x, 0xC0A90000 + B, 0)
    nbytes = 0
    for (w, h, blocks), o in zip(texs, offs):   # the gaps between textures: zero in all three arenas
        end = o + blocks * B
        nbytes += blocks * B
        x[end:(end + align - 1) // align * align].zero_()
    views = [(x[o:o + n * B], y[o:o + n * B], z[o:o + n * B]) for (_, _, n), o in zip(texs, offs)]
    fwd_items = batch.prepare_batch([(fmt, False, xi, yi, st) for xi, yi, _ in views])
    inv_items = batch.prepare_batch([(fmt, True, yi, zi, st) for _, yi, zi in views])
    run_fwd, run_inv = (lambda: batch.run_prepared_batch(fwd_items)), (lambda: batch.run_prepared_batch(inv_items))
    if os.environ.get("DXTLT_BENCH_CORPUS_BY_SIZE") == "1":
        # experiment: one batch call per distinct texture size (what a size-class split inside the library would launch)
        sizes = sorted({n for _, _, n in texs})
        fw = [batch.prepare_batch([(fmt, False, v[0], v[1], st) for v, t in zip(views, texs) if t[2] == n]) for n in sizes]
        iv = [batch.prepare_batch([(fmt, True, v[1], v[2], st) for v, t in zip(views, texs) if t[2] == n]) for n in sizes]
        run_fwd = lambda: [batch.run_prepared_batch(p) for p in fw]
        run_inv = lambda: [batch.run_prepared_batch(p) for p in iv]
    f_ms, i_ms, wall = timed_pair(torch, run_fwd, run_inv, steps, warmup)
    order = sorted(range(len(texs)), key=lambda i: texs[i][2])
    ok = True
    mode, sa, sc = int(st.decorrelation_mode), getattr(st, "split_alpha_endpoints", True), st.split_colour_endpoints
    # against the oracle: every 16th texture in size order, and the smallest, a middle one, the largest (whole textures)
    checked = sorted(set(order[::16]) | {order[0], order[len(order) // 2], order[-1]})
    for i in checked:
        want = oracle_c.transform(fmt, views[i][0].cpu().numpy(), mode, sc, sa)
        ok = ok and bool(np.array_equal(views[i][1].cpu().numpy(), want))
    # The 1-255 bytes between one texture's end and the next one's start must still be zero in BOTH outputs: the forward edge
    # tiles write narrow pieces at every stream end and up to four extra segments past each window, and a store that strayed into
    # a gap (or into a neighbour the inverse does not read back wrongly) would leave the round trip exact.  (z: torch.equal(z, x)
    # below covers its gaps; y is checked here.)
    gaps = [y[o + n * B:(o + n * B + align - 1) // align * align] for (_, _, n), o in zip(texs, offs)]
    gaps_clean = not bool(torch.cat([g for g in gaps if g.numel()]).any()) if any(g.numel() for g in gaps) else True
    odd = sum(1 for _, _, n in texs if n % 2)
    leg = leg_record(
        f"{fmt.upper()} corpus shape of the reference's published benchmark (bc1-api README.MD:286-311: 2130 files, 8692.9 MiB): "
        f"{len(texs)} device-resident textures with full mip chains, 256..4096 pixels a side, {nbytes / 2**20:.1f} MiB, "
        f"{odd} of them with an odd block count, each at the next 256-byte boundary of one arena; default settings; ONE "
        "dxtlt_transform_batch_device call per direction", nbytes, f_ms, i_ms, wall, steps,
        f"batch tiles<{fmt}> forward (halo + edge tiles)", f"batch tiles<{fmt}> inverse (shifted + edge tiles)",
        {"bit_exact_roundtrip": bool(torch.equal(z, x)), "oracle_textures_exact": ok, "oracle_textures_checked": len(checked),
         "forward_gaps_exact": gaps_clean, "textures": len(texs),
         "smallest_blocks": texs[order[0]][2], "largest_blocks": texs[order[-1]][2]})
    if cpu:
        sample = views[order[-1]][0].cpu().numpy()
        leg["cpu_baseline"] = leg_cpu_baseline(fmt, sample) if fmt != "bc1" else cpu_baseline_corpus_bc1(sample)
    return leg


def cpu_baseline_corpus_bc1(sample) -> dict:
    """BC1 default settings on one of the corpus' largest textures, one core: the widest SIMD port this host has (the
    reference's published figures are one thread per file), scalar beside it."""
    import numpy as np

    from oracle import oracle_c

    y, z = np.zeros_like(sample), np.zeros_like(sample)

    def best(fn):
        b = None
        for _ in range(5):
            t0 = time.perf_counter()
            fn()
            dt = time.perf_counter() - t0
            b = dt if b is None else min(b, dt)
        return round(2 * sample.size / b / 2**30, 3)

    def scalar():
        oracle_c.run_mt("bc1", sample, y, 1, True, False, False, 1)
        oracle_c.run_mt("bc1", y, z, 1, True, False, True, 1)

    out = {"value": best(scalar), "unit": "GiB/s", "cores": 1, "kind": "port",
           "sample": f"one texture of the corpus ({sample.size / 2**20:.2f} MiB, odd block count), forward+inverse, best of 5, "
                     "scalar C oracle (gcc -O3)"}
    assert np.array_equal(z, sample)
    if oracle_c.simd_available():
        def simd():
            oracle_c.run_bc1_default_simd(sample, y, False, 1)
            oracle_c.run_bc1_default_simd(y, z, True, 1)

        z[:] = 0
        v = best(simd)
        assert np.array_equal(z, sample)
        out.update({"scalar_value": out["value"], "value": v, "isa": oracle_c.SIMD_NAMES[oracle_c.simd_level()]})
    return out


def leg_exact(leg: dict) -> bool:
    """Every exactness flag a leg carries (round trip; oracle window / prefix / textures) is true."""
    flags = [v for k, v in leg.items() if isinstance(v, bool) and ("exact" in k)]
    return bool(flags) and all(flags)


# The driver's record of a run (BENCH_rNN.json `parsed`) keeps the contract keys and the FIRST 24 members of `config`, and drops
# `legs`.  So `config` is ordered for it: the workload, then every BASELINE config's roofline fractions, then the headline's own
# figures; descriptive members (mode, sharding, block counts, GiB/s -- all derivable) come behind.
DRIVER_KEEPS = 24
LEG_ORDER = ("bc3", "bc7_uniform", "bc7_skewed", "archive", "corpus", "corpus_bc3", "bc2")   # configs[2], [3] x2, [4]'s share, the corpus, BC2
HEADLINE_ORDER = ("bit_exact_roundtrip_and_oracle_window", "fwd_frac", "inv_frac", "fwd_ms", "inv_ms", "format", "total_blocks")


def summarize_legs_into_config(out: dict) -> None:
    """Every leg rides in `config` in short, in the order the driver's 24 kept members want: `workload`, `legs_all_exact`,
    `legs_inexact` (names of legs with any exactness flag false: round trip, oracle window / prefix / textures), then
    `leg_<name>_{fwd_frac,inv_frac}` for LEG_ORDER, then the headline's flag, fractions and times.  Behind them, for readers of the
    whole line: the descriptive members, `leg_<name>_exact` and the nested `legs_summary[name] = [fwd frac, inv frac, exact]`."""
    legs = out.get("legs") or {}
    cfg = out["config"]
    names = [n for n in LEG_ORDER if n in legs] + [n for n in legs if n not in LEG_ORDER]
    summary, fracs, exact = {}, {}, {}
    for name in names:
        r = legs[name].get("roofline") or {}
        f, i, ok = r.get("frac"), (r.get("inverse_kernel") or {}).get("frac"), leg_exact(legs[name])
        summary[name] = [f, i, ok]
        fracs[f"leg_{name}_fwd_frac"], fracs[f"leg_{name}_inv_frac"], exact[f"leg_{name}_exact"] = f, i, ok
    head = {"workload": cfg["workload"], "legs_all_exact": bool(summary) and all(v[2] for v in summary.values()),
            "legs_inexact": [n for n in names if not summary[n][2]], **fracs}
    head.update({k: cfg[k] for k in HEADLINE_ORDER if k in cfg})
    out["config"] = {**head, **{k: v for k, v in cfg.items() if k not in head}, **exact, "legs_summary": summary}


def attach_corpus_traffic(legs: dict) -> None:
    """HBM bytes per launch of the corpus legs' batch kernels from the committed PMC passes, when the leg ran the profiled corpus."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return
    note = (f"{rec.get('source', 'profiles/pmc_traffic.json')}: committed rocprofv3 --pmc passes (FETCH_SIZE x2 per the gfx950 rule, "
            "WRITE_SIZE); NOT measured by this run")
    for name in ("corpus", "corpus_bc3"):
        r = (rec.get("legs") or {}).get(name) or {}
        if name in legs and r.get("bytes") == legs[name]["bytes"] and r.get("fwd"):
            legs[name]["roofline"].update({"traffic": r["fwd"], "inverse_traffic": r["inv"], "traffic_source": note})


def bc7_main(args) -> None:
    """BASELINE.json configs[3]: BC7 forward (+ inverse) on a synthetic mode-mixed buffer.  Same JSON contract; the
    transform is this build's own format (docs/BC7_FORMAT.md, version 2; the reference has none), so parity is a round
    trip plus the build's own CPU statement.  One kernel per direction, 2 * len of traffic."""
    import torch

    import dxt_lossless_transform_amd as pkg
    from dxt_lossless_transform_amd import bc7

    R = Ranks(args)
    rank, world, dev = R.rank, R.world, R.dev
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

    barrier = R.barrier

    barrier()     # the group's first collective before the warm-up, not between it and the timed region (see main())
    # before the W warm-up steps: bring the chip to its steady clocks (clock_warm; stated in the line as clock_warmup_ms)
    clock_warm(torch, lambda: bc7.transform_bc7(x, y), lambda: bc7.untransform_bc7(y, z))
    for _ in range(args.warmup):
        bc7.transform_bc7(x, y)
        bc7.untransform_bc7(y, z)
    torch.cuda.synchronize()
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        bc7.transform_bc7(x, y)
        ev[k][1].record()
        bc7.untransform_bc7(y, z)
        ev[k][2].record()
    torch.cuda.synchronize()
    barrier()
    elapsed_here = time.perf_counter() - t0
    elapsed = R.max_over_ranks(elapsed_here)
    fwd_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps
    inv_ms = sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps
    round_trip_exact = bool(torch.equal(z, x))
    cpu = None
    import numpy as np

    if rank == 0:
        from oracle import oracle_c

        oracle_c.lib()          # one process brings the checker's .so up to date before the others load it
    R.cpu_barrier()
    from oracle import oracle_c

    # a prefix of THIS rank's buffer (whole granules: its streams are a transform of their own) against the CPU statement: 64 MiB on
    # rank 0 (also the cpu_baseline sample), 8 MiB on the others (their data differ: the fill starts at rank * nbytes / 8)
    sample = min(nbytes, (64 << 20) if rank == 0 else (8 << 20))
    small_y = torch.empty(sample, dtype=torch.uint8, device=dev)
    bc7.transform_bc7(x[:sample], small_y)
    xin = x[:sample].cpu().numpy()
    t1 = time.perf_counter()
    want = oracle_c.transform_bc7(xin)
    t2 = time.perf_counter()
    back = oracle_c.transform_bc7(want, inverse=True)
    t3 = time.perf_counter()
    oracle_prefix_exact = bool(np.array_equal(small_y.cpu().numpy(), want)) and bool(np.array_equal(back, xin))
    if rank == 0:
        cpu = {"value": round(2 * sample / (t3 - t1) / 2**30, 3), "unit": "GiB/s", "cores": 1, "kind": "port",
               "sample": f"{sample >> 20} MiB of the same mode-mixed workload, forward+inverse, scalar C restatement of this build's "
                         "own BC7 format (oracle/dxtlt_oracle_bc7.c; the reference has no BC7 transform to time)",
               "fwd_value": round(sample / (t2 - t1) / 2**30, 3)}
    per_rank = per_rank_record(R, fwd_ms, inv_ms, elapsed_here,
                               {"round_trip_exact": round_trip_exact, "oracle_prefix_exact": oracle_prefix_exact})
    ok = all_ranks_exact(per_rank, ("round_trip_exact", "oracle_prefix_exact"))
    assert ok, f"GPU result differs from the oracle / round trip failed: {per_rank['per_rank']}"
    if rank != 0:
        R.finish()
        return
    achieved = 2 * nbytes / (fwd_ms * 1e-3) / 1e9
    achieved_inv = 2 * nbytes / (inv_ms * 1e-3) / 1e9
    out = {
        "metric": "GiB/s BC blocks transformed (fwd+inv)",
        "value": round(2 * nbytes * args.steps * world / elapsed / 2**30, 2),
        "unit": "GiB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "u8", "data": "synthetic", "clock_warmup_ms": LEG_WARM_MS, **per_rank,
        "config": {
            "workload": f"BC7 granule-sorted field split v2, forward+inverse (this build's own format, parity unpinned), {nbytes / 2**30:g} GiB "
                        "synthetic mode-mixed buffer per GPU, modes 0-7 uniform (BASELINE.json configs[3])",
            "format": "bc7", "blocks_per_gpu": blocks, "bytes_per_gpu": nbytes, "seed": hex(seed),
            "sharding": "independent buffer per rank, no collective",
            "bit_exact_roundtrip_and_oracle_prefix": ok,
            "fwd_ms": round(fwd_ms, 4), "inv_ms": round(inv_ms, 4),
            "fwd_GiBps": round(nbytes / (fwd_ms * 1e-3) / 2**30, 1), "inv_GiBps": round(nbytes / (inv_ms * 1e-3) / 2**30, 1),
            "fwd_frac": round(achieved / HBM_PEAK_GBPS, 4), "inv_frac": round(achieved_inv / HBM_PEAK_GBPS, 4),
        },
        "roofline": {
            "bound": "hbm", "kernel": "bc7_forward (one kernel, one pass)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
            "traffic": None, "algorithmic_bytes_per_launch": 2 * nbytes, "pipeline_bytes": 2 * nbytes,
            "inverse_kernel": {"kernel": "bc7_inverse (one kernel, one pass)",
                               "achieved": round(achieved_inv, 1), "frac": round(achieved_inv / HBM_PEAK_GBPS, 4)},
        },
    }
    if world == 1 and not args.no_cpu_baseline and cpu is not None:
        out["cpu_baseline"] = cpu
    print(json.dumps(out), flush=True)
    R.finish()


def archive_main(args) -> None:
    """BASELINE.json configs[4]: an archive of alternating 256 MiB BC1 and BC3 textures (--size-gib per GPU: 64 GiB over
    8 GPUs), each transformed with its format's default settings, no collective.  Two ways to cut it over the ranks:
      --archive-split texture  rank r owns the contiguous byte range [r, r + 1) * size of the archive, which is K whole
                               textures (the boundaries fall on texture boundaries); each is transformed whole
      --archive-split range    every texture's block range is cut into `world` contiguous ranges and rank r transforms
                               range r of EVERY texture with dxtlt_transform_range_device (AoS slice <-> its slices of
                               the texture's whole transformed buffer)
    A step transforms and restores every texture (slice) of the rank.  Verified per texture (exact round trip, an oracle
    window) and, since random blocks compress to ratio 1, the compression-ratio half of the config is checked on the
    reference's real 256x256 test textures with zstd levels 3 and 19 (the system libzstd through ctypes,
    tools/zstd_ratio.py; zlib level 6 beside it, and alone when libzstd is absent): GPU output == CPU output byte for
    byte, so the ratios are equal, and both are reported."""
    import zlib
    from tools import zstd_ratio

    import numpy as np
    import torch

    import dxt_lossless_transform_amd as pkg
    from oracle import oracle_c

    R = Ranks(args)
    rank, world, dev = R.rank, R.world, R.dev
    pkg.load()
    tex_bytes = 256 << 20
    per_gpu = int((args.size_gib if args.size_gib else 8.0) * (1 << 30))
    by_range = args.archive_split == "range"
    st = {"bc1": pkg.Bc1TransformSettings(), "bc3": pkg.Bc3TransformSettings()}
    fwd = {f: getattr(pkg, f"transform_{f}_with_settings") for f in st}
    inv = {f: getattr(pkg, f"untransform_{f}_with_settings") for f in st}
    if by_range:
        # every texture of the archive; this rank's block range of each (AoS slice, whole SoA buffer per texture)
        k = max(2, per_gpu * world // tex_bytes // 2 * 2)
        fmts = ["bc1" if i % 2 == 0 else "bc3" for i in range(k)]
        plans = [pkg.plan_shards(tex_bytes // pkg.BLOCK_BYTES[f], world)[rank] for f in fmts]
        xs = [torch.empty(n * pkg.BLOCK_BYTES[f], dtype=torch.uint8, device=dev) for f, (_, n) in zip(fmts, plans)]
        ys = [torch.empty(tex_bytes, dtype=torch.uint8, device=dev) for _ in range(k)]
        zs = [torch.empty_like(x) for x in xs]
        for i, x in enumerate(xs):
            first = plans[i][0]
            pkg.fill_splitmix64(x, 0x0A5C0005, i * (tex_bytes // 8) + first * pkg.BLOCK_BYTES[fmts[i]] // 8)

        def step():
            for i in range(k):
                total = tex_bytes // pkg.BLOCK_BYTES[fmts[i]]
                pkg.transform_range(fmts[i], False, xs[i], ys[i], total, plans[i][0], plans[i][1], st[fmts[i]])
            for i in range(k):
                total = tex_bytes // pkg.BLOCK_BYTES[fmts[i]]
                pkg.transform_range(fmts[i], True, ys[i], zs[i], total, plans[i][0], plans[i][1], st[fmts[i]])
    else:
        k = max(2, per_gpu // tex_bytes // 2 * 2)           # textures per rank, BC1 and BC3 alternating
        fmts = ["bc1" if i % 2 == 0 else "bc3" for i in range(k)]
        plans = [(0, tex_bytes // pkg.BLOCK_BYTES[f]) for f in fmts]
        xs = [torch.empty(tex_bytes, dtype=torch.uint8, device=dev) for _ in range(k)]
        ys = [torch.empty_like(x) for x in xs]
        zs = [torch.empty_like(x) for x in xs]
        for i, x in enumerate(xs):
            pkg.fill_splitmix64(x, 0x0A5C0005, (rank * k + i) * (tex_bytes // 8))   # one logical 64 GiB stream of blocks

        def step():
            for i in range(k):
                fwd[fmts[i]](xs[i], ys[i], st[fmts[i]])
            for i in range(k):
                inv[fmts[i]](ys[i], zs[i], st[fmts[i]])

    barrier = R.barrier

    barrier()     # the group's first collective before the warm-up, not between it and the timed region (see main())
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
    elapsed_here = time.perf_counter() - t0
    elapsed = R.max_over_ranks(elapsed_here)
    round_trip_exact = all(bool(torch.equal(z, x)) for x, z in zip(xs, zs))
    ok = True
    win = 1 << 15
    for i in (0, 1, k - 1):                                # one window per format and the rank's last texture
        f = fmts[i]
        B = pkg.BLOCK_BYTES[f]
        blocks = tex_bytes // B
        lf = min(plans[i][1] // 3 + 17, plans[i][1] - win)      # inside this rank's slice of the texture
        first = plans[i][0] + lf
        xin = xs[i][lf * B:(lf + win) * B].cpu().numpy()
        want = oracle_c.transform(f, xin, 1, True, True)
        got = np.empty_like(want)
        for off, w in pkg.stream_table(f, st[f]):
            got[off * win: off * win + w * win] = ys[i][off * blocks + w * first: off * blocks + w * (first + win)].cpu().numpy()
        ok = ok and bool(np.array_equal(got, want))
    # every rank's own textures (rank-dependent data): its round trip and its oracle windows ride in per_rank, the line's flag is the AND
    per_rank = per_rank_record(R, None, None, elapsed_here, {"round_trip_exact": round_trip_exact, "oracle_windows_exact": ok})
    ok = all_ranks_exact(per_rank, ("round_trip_exact", "oracle_windows_exact"))
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
            if zstd_ratio.available():
                for level in (3, 19):
                    ratios[f][f"plain_zstd{level}"] = round(tiled.size / zstd_ratio.compressed_size(tiled, level), 4)
                    ratios[f][f"transformed_gpu_zstd{level}"] = round(tiled.size / zstd_ratio.compressed_size(gpu_out, level), 4)
                    ratios[f][f"transformed_cpu_zstd{level}"] = round(tiled.size / zstd_ratio.compressed_size(cpu_out, level), 4)
    assert ok, f"GPU result differs from the oracle / round trip failed: {per_rank['per_rank']}"
    if rank != 0:
        R.finish()
        return
    rank_bytes = sum(int(x.numel()) for x in xs)          # block bytes this rank feeds to one direction per step
    total = 2 * rank_bytes * args.steps * world
    archive_gib = rank_bytes * world / 2**30
    out = {
        "metric": "GiB/s BC blocks transformed (fwd+inv)", "value": round(total / elapsed / 2**30, 2), "unit": "GiB/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic", **per_rank,
        "config": {
            "workload": f"BC1+BC3 mixed archive, {archive_gib:g} GiB of alternating 256 MiB textures, default settings per format, "
                        + ("every texture's block range cut over the ranks (dxtlt_transform_range_device), " if by_range else
                           "contiguous byte range of the archive per rank = whole textures, ")
                        + "no collective (BASELINE.json configs[4]; compression ratios: zstd -3 / -19 of the system libzstd when present, zlib-6 beside them)",
            "archive_split": args.archive_split, "textures_touched_per_gpu": k, "texture_bytes": tex_bytes, "seed": "0xa5c0005",
            "bit_exact_roundtrip_and_oracle_windows": ok,
            "ratio_on_real_textures": ratios, "zstd_version": zstd_ratio.version(),
        },
        "roofline": {"bound": "hbm", "kernel": "fwd_tiled + inv_tiled over the archive", "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "achieved": round(2 * (2 * rank_bytes * args.steps) / elapsed / 1e9, 1),
                     "frac": round(2 * (2 * rank_bytes * args.steps) / elapsed / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None,
                     "note": "wall clock of the whole step per GPU (launch gaps included), algorithmic 2 * bytes per direction"},
    }
    print(json.dumps(out), flush=True)
    R.finish()


def pcie_pair_ceiling(torch, dev, nbytes: int = 1 << 30, reps: int = 3) -> dict:
    """The ceiling of the host path, measured in this run: ONE pinned host -> device copy and ONE pinned device -> host copy
    of `nbytes` each, concurrently on two streams of `dev` -- what the link gives a caller whose buffers are already
    pinned, with no kernel and no placement in the way.  GiB/s per direction (both directions are busy at once)."""
    # the pinned buffers are first touched by a thread next to the device (their pages then sit on its NUMA node, as the shard
    # workers' staging does): measured from a far node the same pair moves at 27 instead of 45 GiB/s per direction
    import threading

    import dxt_lossless_transform_amd as pkg

    before = os.sched_getaffinity(0)
    cpus = set()
    for part in (pkg.device_local_cpulist(dev.index or 0) or "").split(","):
        a, _, b = part.partition("-")
        if a:
            cpus |= set(range(int(a), int(b or a) + 1))
    cpus &= before
    if cpus:
        os.sched_setaffinity(threading.get_native_id(), cpus)
    try:
        h_up = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        h_down = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
        h_up.zero_()
        h_down.zero_()
    finally:
        os.sched_setaffinity(threading.get_native_id(), before)
    d_up = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    d_down = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    best = {"pair": None, "up": None, "down": None}

    def run(up: bool, down: bool) -> float:
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        if up:
            with torch.cuda.stream(s1):
                d_up.copy_(h_up, non_blocking=True)
        if down:
            with torch.cuda.stream(s2):
                h_down.copy_(d_down, non_blocking=True)
        s1.synchronize()
        s2.synchronize()
        return time.perf_counter() - t0

    run(True, True)
    for _ in range(reps):
        for key, (u, d) in (("pair", (True, True)), ("up", (True, False)), ("down", (False, True))):
            dt = run(u, d)
            best[key] = dt if best[key] is None else min(best[key], dt)
    del h_up, h_down, d_up, d_down
    return {"bytes_each_way": nbytes,
            "concurrent_GiBps_per_direction": round(nbytes / best["pair"] / 2**30, 2),
            "h2d_alone_GiBps": round(nbytes / best["up"] / 2**30, 2), "d2h_alone_GiBps": round(nbytes / best["down"] / 2**30, 2),
            "what": "one pinned H2D and one pinned D2H hipMemcpyAsync of 1 GiB each on two streams of the same device, best of 3"}


def sharded_host_array(pkg, torch, fmt, settings, nbytes: int, seed: int, dev, want_devices: int, ceiling=None) -> dict:
    """north_star's multi-GPU statement as ONE call: a host-resident block array is split by contiguous block range over
    every visible device and each shard's slice of every stream lands at its final host offset (dxtlt_transform_sharded;
    the reference side of the contract is one call over the whole array, transform_with_settings.rs:31-72).  Host
    buffers are ordinary pageable memory, as a caller's are (DXTLT_BENCH_PINNED_HOST=1 pins them: measured SLOWER at this
    size -- three 8 GiB pinned arrays move at 18-25 GiB/s where pageable ones move at 41-43, tools/pinned_host_probe.py).
    Checked: exact round trip of the whole array, oracle windows that straddle every shard boundary."""
    import numpy as np

    from oracle import oracle_c

    block = pkg.BLOCK_BYTES[fmt]
    nbytes -= nbytes % (block * 2048)
    blocks = nbytes // block
    n_dev = max(1, min(int(want_devices), pkg.load().dxtlt_device_count()))   # as many devices as the job has ranks
    t0 = time.perf_counter()
    pinned = os.environ.get("DXTLT_BENCH_PINNED_HOST") == "1"
    h_in = torch.empty(nbytes, dtype=torch.uint8, pin_memory=pinned)
    h_soa = torch.empty(nbytes, dtype=torch.uint8, pin_memory=pinned)
    h_back = torch.empty(nbytes, dtype=torch.uint8, pin_memory=pinned)
    first_touch = os.environ.get("DXTLT_BENCH_FIRST_TOUCH", "near")
    if not pinned:
        # First touch outside the timed calls -- and, as a NUMA-aware caller would, by a thread that runs next to the device
        # that will read / write those pages: every shard's block range of the AoS arrays and its slice of every stream of
        # the transformed array is zeroed by a thread bound to its device's local CPUs (DXTLT_BENCH_FIRST_TOUCH=main: by
        # this thread, wherever it runs).
        import threading

        def touch(dev_index, ranges):
            cpus = set()
            for part in (pkg.device_local_cpulist(dev_index) or "").split(","):
                a, _, b = part.partition("-")
                if a:
                    cpus |= set(range(int(a), int(b or a) + 1))
            cpus &= os.sched_getaffinity(0)
            if cpus and first_touch == "near":
                os.sched_setaffinity(threading.get_native_id(), cpus)
            for arr, lo, n in ranges:
                arr[lo:lo + n].zero_()

        table = pkg.stream_table(fmt, settings)
        workers = []
        for d, (first, count) in enumerate(pkg.plan_shards(blocks, n_dev)):
            ranges = [(h_in, first * block, count * block), (h_back, first * block, count * block)]
            ranges += [(h_soa, off * blocks + w * first, w * count) for off, w in table]
            workers.append(threading.Thread(target=touch, args=(d, ranges)))
        for w in workers:
            w.start()
        for w in workers:
            w.join()
    # the same logical array as the kernel legs, generated on the device in pieces and copied out
    piece = min(nbytes, 1 << 30)
    scratch = torch.empty(piece, dtype=torch.uint8, device=dev)
    for lo in range(0, nbytes, piece):
        n = min(piece, nbytes - lo)
        pkg.fill_splitmix64(scratch[:n], seed, lo // 8)
        h_in[lo:lo + n].copy_(scratch[:n])
    torch.cuda.synchronize()
    del scratch
    setup_s = time.perf_counter() - t0
    a_in, a_soa, a_back = h_in.numpy(), h_soa.numpy(), h_back.numpy()

    def timed(inverse, src, dst):
        best = None
        for _ in range(3):
            t = time.perf_counter()
            pkg.transform_sharded(fmt, inverse, src, dst, settings, n_dev)
            dt = time.perf_counter() - t
            best = dt if best is None else min(best, dt)
        return best

    fwd_s = timed(False, a_in, a_soa)
    fwd_shards = pkg.sharded_last_stats()
    inv_s = timed(True, a_soa, a_back)
    inv_shards = pkg.sharded_last_stats()
    # per device: block bytes of its shard / the shard's own wall time (upload + kernels + downloads), last of the timed calls
    per_device = [{"device": f["device"], "bytes": f["blocks"] * block, "cpus_bound": f["cpus_bound"],
                   "local_cpulist": pkg.device_local_cpulist(f["device"]),
                   "fwd_GiBps": round(f["blocks"] * block / f["seconds"] / 2**30, 2),
                   "inv_GiBps": round(i["blocks"] * block / i["seconds"] / 2**30, 2)}
                  for f, i in zip(fwd_shards, inv_shards)]
    ok = bool(torch.equal(h_back, h_in))
    win = 1 << 14
    table = pkg.stream_table(fmt, settings)
    mode, sa, sc = int(settings.decorrelation_mode), getattr(settings, "split_alpha_endpoints", True), settings.split_colour_endpoints
    firsts = [0, blocks - win] + [max(0, f - win // 2) for f, _ in pkg.plan_shards(blocks, max(1, n_dev))[1:]]
    for first in firsts:
        want = oracle_c.transform(fmt, a_in[first * block:(first + win) * block], mode, sc, sa)
        got = np.empty_like(want)
        for off, w in table:
            got[off * win: off * win + w * win] = a_soa[off * blocks + w * first: off * blocks + w * (first + win)]
        ok = ok and bool(np.array_equal(got, want))
    assert ok, "sharded host array: result differs from the oracle / round trip failed"
    # the link's own ceiling, same run, same device: a sharded call moves every byte up and every byte down, both directions busy
    try:
        if ceiling is None:
            ceiling = pcie_pair_ceiling(torch, dev, min(1 << 30, max(nbytes, 1 << 20)))
        peak = ceiling["concurrent_GiBps_per_direction"]
        for row in per_device:
            row["roofline"] = {"bound": "pcie", "unit": "GiB/s per direction", "peak": peak,
                               "achieved": row["fwd_GiBps"], "frac": round(row["fwd_GiBps"] / peak, 4),
                               "inverse": {"achieved": row["inv_GiBps"], "frac": round(row["inv_GiBps"] / peak, 4)}}
    except Exception as e:  # noqa: BLE001
        ceiling = {"error": f"{type(e).__name__}: {e}"}
    return {
        "pcie_ceiling": ceiling,
        "entry_point": "dxtlt_transform_sharded (one process, one host thread per device, chunked H2D | kernel | per-stream D2H)",
        "array_bytes": nbytes, "devices": n_dev, "host_memory": "pinned" if pinned else "pageable",
        "fwd_GiBps": round(nbytes / fwd_s / 2**30, 2), "inv_GiBps": round(nbytes / inv_s / 2**30, 2),
        "fwd_plus_inv_GiBps": round(2 * nbytes / (fwd_s + inv_s) / 2**30, 2),
        "per_device": per_device, "first_touch": first_touch if not pinned else "pinned",
        "numa": "each shard's worker thread (and the downloader thread it starts) is bound to the CPUs local to its device "
                "(cpus_bound; 0 = the kernel names no node or none of its CPUs is available to this process)",
        "bit_exact_roundtrip_and_oracle_windows_across_shard_boundaries": ok,
        "setup_s": round(setup_s, 2),
        "note": "host -> device -> host with placement, PCIe-bound; reported beside `value`, never as `value`",
    }


def main() -> None:
    args = parse_args()
    self_launch_if_needed(args)
    if args.rendezvous_only:
        return rendezvous_only(args)
    if args.workload == "archive":
        return archive_main(args)
    if args.format == "bc7":
        return bc7_main(args)
    if args.size_gib is None:
        args.size_gib = 8.0
    R = Ranks(args)
    torch, rank, world, dev = R.torch, R.rank, R.world, R.dev

    import dxt_lossless_transform_amd as pkg

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
    seed = {"bc1": 0x0BC10002, "bc2": 0x0BC20002, "bc3": 0x0BC30003}[fmt]
    strong = args.scaling == "strong"

    # strong: ONE logical array of --size-gib, this rank owns a contiguous block range of it; weak: one logical array of
    # world * --size-gib, rank r holds blocks [r * blocks, (r + 1) * blocks)
    total_blocks, first, blocks = job_shape(args, block, world, rank, pkg.plan_shards)
    nbytes = blocks * block

    # The link's ceiling for the host-array leg is taken FIRST, while the host is as the process found it: measured at the end
    # of a full run -- 24 GiB of host arrays first-touched next to the device, the legs' CPU baselines behind it -- the same
    # pinned pair lands on far memory and reads 27 instead of 45 GiB/s per direction (profiles/r04_channel_map.txt, item 4).
    host_gib = args.host_array_gib if args.host_array_gib is not None else min(args.size_gib, 8.0)
    ceiling = None
    if rank == 0 and host_gib > 0 and not args.drop_blocks:
        try:
            ceiling = pcie_pair_ceiling(torch, dev, min(1 << 30, max(int(host_gib * (1 << 30)), 1 << 20)))
            ceiling["when"] = "first thing in the run"
        except Exception as e:  # noqa: BLE001
            ceiling = {"error": f"{type(e).__name__}: {e}"}

    x = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    z = torch.empty_like(x)
    pkg.fill_splitmix64(x, seed, first * block // 8)
    if strong:
        # the WHOLE transformed buffer; the range calls touch only this rank's slice of every stream
        y = torch.empty(total_blocks * block, dtype=torch.uint8, device=dev)
        y_total, y_first = total_blocks, first

        def fwd():
            pkg.transform_range(fmt, False, x, y, total_blocks, first, blocks, settings)

        def inv():
            pkg.transform_range(fmt, True, y, z, total_blocks, first, blocks, settings)
    else:
        # stand-alone shard: its compact result is this rank's slice of every stream, packed
        y = torch.empty_like(x)
        y_total, y_first = blocks, 0
        f_fwd = getattr(pkg, f"transform_{fmt}_with_settings")
        f_inv = getattr(pkg, f"untransform_{fmt}_with_settings")

        def fwd():
            f_fwd(x, y, settings)

        def inv():
            f_inv(y, z, settings)
    torch.cuda.synchronize()

    # the process group's first collective (RCCL sets its connections up here at the latest) happens BEFORE the warm-up steps, so that
    # the barrier in front of the timed region is a quick one and the chip does not idle back down its clock ramp between the two
    R.barrier()
    for _ in range(args.warmup):
        fwd()
        inv()
    torch.cuda.synchronize()

    # per-kernel timing: HIP events on the stream the kernels are launched on (torch's current stream)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]

    R.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        fwd()
        ev[k][1].record()
        inv()
        ev[k][2].record()
    torch.cuda.synchronize()
    R.barrier()
    elapsed_here = time.perf_counter() - t0
    elapsed = R.max_over_ranks(elapsed_here)

    fwd_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / args.steps
    inv_ms = sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps
    # correctness inside the run, on EVERY rank (rank r's blocks are its own: fill_splitmix64(x, seed, first * block / 8)): exact round
    # trip, and one window of the rank's own block range against the oracle; the flags ride in per_rank and the line's flag is their AND
    import numpy as np

    if rank == 0:
        from oracle import oracle_c

        oracle_c.lib()          # one process brings the checker's .so up to date before the others load it
    R.cpu_barrier()
    from oracle import oracle_c

    round_trip_exact = bool(torch.equal(z, x))
    win = min(1 << 16, blocks)
    lf = min(blocks // 2 + 4097, blocks - win)
    xin = x[lf * block:(lf + win) * block].cpu().numpy()
    want = oracle_c.transform(fmt, xin, int(settings.decorrelation_mode), settings.split_colour_endpoints,
                              getattr(settings, "split_alpha_endpoints", True))
    got = np.empty_like(want)
    for off, w in pkg.stream_table(fmt, settings):
        lo = off * y_total + w * (y_first + lf)
        got[off * win: off * win + w * win] = y[lo: lo + w * win].cpu().numpy()
    oracle_window_exact = bool(np.array_equal(got, want))
    per_rank = per_rank_record(R, fwd_ms, inv_ms, elapsed_here,
                               {"round_trip_exact": round_trip_exact, "oracle_window_exact": oracle_window_exact})
    bit_exact = all_ranks_exact(per_rank, ("round_trip_exact", "oracle_window_exact"))
    assert bit_exact, f"GPU result differs from the oracle / round trip failed: {per_rank['per_rank']}"

    if rank != 0:
        # rank 0 still drives every device for the sharded_host_array leg; stay out of its way, then leave together
        del x, y, z
        torch.cuda.empty_cache()
        R.cpu_barrier()
        R.finish()
        return

    job_bytes = (total_blocks * block) if strong else nbytes * world     # block bytes fed to one direction per step
    value = 2 * job_bytes * args.steps / elapsed / 2**30
    achieved = 2 * nbytes / (fwd_ms * 1e-3) / 1e9  # algorithmic bytes: read len + write len
    achieved_inv = 2 * nbytes / (inv_ms * 1e-3) / 1e9

    # HBM traffic is a PMC figure and PMC passes are separate runs (tools/pmc_traffic.sh): what is quoted here is the
    # committed result of such a pass over this very workload, and says so
    traffic, traffic_source = None, None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        try:
            with open(tpath) as f:
                rec = json.load(f)
            if rec.get("workload_bytes") == nbytes and rec.get("format") == fmt and not args.settings:
                traffic = rec.get("fwd_hbm_bytes_per_launch")
                traffic_source = (f"{rec.get('source', 'profiles/pmc_traffic.json')}: committed rocprofv3 --pmc passes over this "
                                  "workload (FETCH_SIZE x2 per the gfx950 rule, WRITE_SIZE); NOT measured by this run")
        except Exception:
            traffic = None

    dflt = "YCoCg Variant1, split colour endpoints" if fmt != "bc3" else "YCoCg Variant1, split alpha + colour endpoints"
    if strong:
        shape = (f"ONE logical {job_bytes / 2**30:g} GiB random block array split by contiguous block range over {world} GPU(s) "
                 f"(dxtlt_transform_range_device: AoS slice <-> its slices of the whole transformed buffer)")
    else:
        shape = f"{nbytes / 2**30:g} GiB random block buffer per GPU (BASELINE.json configs[1])"
    out = {
        "metric": "GiB/s BC blocks transformed (fwd+inv)",
        "value": round(value, 2),
        "unit": "GiB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True,
        "scaling": args.scaling,
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        **per_rank,
        "config": {
            "workload": f"{fmt.upper()} forward+inverse, "
                        + (f"settings {args.settings} (variant,split_alpha,split_colour), " if args.settings else f"default settings ({dflt}), ")
                        + shape,
            "mode": ("strong scaling: one array, range-sharded" if strong else
                     "weak scaling: one stand-alone shard of the logical array per GPU") + "; `value` = kernels only, data resident in HBM",
            "format": fmt, "blocks_per_gpu": blocks, "bytes_per_gpu": nbytes, "total_blocks": total_blocks, "seed": hex(seed),
            "sharding": "contiguous block range per rank, no collective",
            "bit_exact_roundtrip_and_oracle_window": bit_exact,
            "fwd_ms": round(fwd_ms, 4), "inv_ms": round(inv_ms, 4),
            "fwd_GiBps": round(nbytes / (fwd_ms * 1e-3) / 2**30, 1), "inv_GiBps": round(nbytes / (inv_ms * 1e-3) / 2**30, 1),
            "fwd_frac": round(achieved / HBM_PEAK_GBPS, 4), "inv_frac": round(achieved_inv / HBM_PEAK_GBPS, 4),
        },
        "roofline": {
            "bound": "hbm", "kernel": f"fwd_tiled<{fmt}>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_source,
            "algorithmic_bytes_per_launch": 2 * nbytes, "inv_achieved": round(achieved_inv, 1),
            "inv_frac": round(achieved_inv / HBM_PEAK_GBPS, 4),
            "inverse_kernel": {"kernel": f"inv_tiled<{fmt}>", "achieved": round(achieved_inv, 1),
                               "frac": round(achieved_inv / HBM_PEAK_GBPS, 4)},
        },
    }
    default_run = (world == 1 and fmt == "bc1" and not args.settings and not args.drop_blocks and not args.force_path
                   and not args.tile_threads and not strong and (nbytes >= (4 << 30) or args.small_legs))
    if default_run and args.leg_steps > 0:
        # the other single-GPU configurations of BASELINE.json, on the same three device buffers (x is overwritten)
        out["legs"] = run_legs(pkg, torch, dev, x, y, z, args.leg_steps, 2, cpu=not args.no_cpu_baseline)
        # the reference's own benchmark shape: a corpus of mip-chained textures through one batch call per direction
        del x, y, z
        torch.cuda.empty_cache()
        scale = 1.0 if nbytes >= (4 << 30) else 0.01
        out["legs"]["corpus"] = run_corpus_leg(pkg, torch, dev, "bc1", args.leg_steps, 2, scale, cpu=not args.no_cpu_baseline)
        out["legs"]["corpus_bc3"] = run_corpus_leg(pkg, torch, dev, "bc3", args.leg_steps, 2, scale, cpu=not args.no_cpu_baseline)
        attach_corpus_traffic(out["legs"])
        summarize_legs_into_config(out)
        x = y = z = None
    if host_gib > 0 and not args.drop_blocks:
        del y, z
        torch.cuda.empty_cache()
        # an extra leg beside `value`: if it cannot run (a device this rank may not open, host memory), say so and keep the line
        try:
            out["sharded_host_array"] = sharded_host_array(pkg, torch, fmt, settings, int(host_gib * (1 << 30)), seed, dev, world,
                                                           ceiling if ceiling and "error" not in ceiling else None)
        except Exception as e:  # noqa: BLE001
            out["sharded_host_array"] = {"error": f"{type(e).__name__}: {e}"}
    R.cpu_barrier()
    if world == 1 and not args.no_cpu_baseline:
        s = (int(settings.decorrelation_mode), bool(getattr(settings, "split_alpha_endpoints", True)),
             bool(settings.split_colour_endpoints))
        out["cpu_baseline"] = cpu_baseline(fmt, s, args.cpu_sample_mib)
    print(json.dumps(out), flush=True)
    R.finish()


if __name__ == "__main__":
    main()
