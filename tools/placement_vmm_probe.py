"""DO NOT RUN THE VMM ARMS ON A SHARED POOL AGAIN: on ROCm 7.2.0 / gfx950 hipMemMap-backed ranges returned wrong data once a virtual
range was reused and, in the last pass, raised a GPU memory access fault inside a range just reported mapped (profiles/r05_placement_vmm.txt
items 1 and 5).  The hipmalloc and hm: arms are plain hipMalloc and safe.  Kept as the record of how round 5's figures were taken.

What is "physical placement"?    The BC7 kernels ran at 0.72-0.78 of peak by allocation inside one process and
the BC3 corpus inverse at two levels by process (profiles/r04_bc7_placement.txt, r04_batch_edge_tiles.txt).  Here the allocator is under
the probe's control: the three buffers of a measurement are backed through HIP's virtual-memory API (tools/vmm_helper.cpp) --

    hipmalloc   plain hipMalloc (what torch's allocator hands the bench)                 -- the arm round 4 measured
    whole       hipMemCreate: ONE physical handle per buffer
    2m          one handle per 2 MiB chunk (the page-table fragment the driver prefers), mapped in creation order
    2m-shuf     the same chunks mapped in a seeded random order (placement decoupled from the order the driver hands pages out in)
    small       one handle per SMALL chunk: the minimum granularity hipMemGetAllocationGranularity reports (4 KiB on this box, where
                minimum == recommended), raised to at least --min-chunk-kib and doubled until a trio allocates within --alloc-budget-s
    small-shuf  the same, shuffled

twelve fresh mappings per arm (a spacer allocation of varying size in front of each, so that the driver cannot hand the same pages
back), same kernels, same data: BC7 4 GiB uniform mix (fwd / inv) and the BC3 corpus, one batch call per direction (fwd / inv).
Steady state: 150 ms of untimed pairs, 20 timed pairs, HIP events on the launch stream.  Prints fraction of 8 TB/s.

    python tools/placement_vmm_probe.py [--trials 12] [--workloads bc7,corpus] [--arms hipmalloc,whole,rec,rec-shuf,min,min-shuf]
"""
import argparse
import ctypes as C
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HELPER = os.path.join(ROOT, "tools", "libvmm_helper.so")

ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=12)
ap.add_argument("--workloads", default="bc7,corpus")
ap.add_argument("--arms", default="hipmalloc", help="round 5 ran hipmalloc,whole,2m,2m-shuf,small,small-shuf and the hm: / vw: arms")
ap.add_argument("--vmm-is-unsafe-here-and-i-run-it-anyway", dest="allow_vmm", action="store_true",
                help="the VMM arms (everything but hipmalloc and hm:...) returned wrong data and faulted on ROCm 7.2.0 / gfx950")
ap.add_argument("--arm-sep", default=",", help="separator of --arms (the hm: / vw: arms hold commas: use ';')")
ap.add_argument("--copy", default="kernel", choices=("kernel", "hipmemcpy"), help="how data gets into / is checked in the probe's buffers")
ap.add_argument("--diagnose", action="store_true", help="first: do hipMemcpy / hipMemset agree with kernel copies on VMM-backed buffers?")
ap.add_argument("--min-chunk-kib", type=int, default=64)
ap.add_argument("--min-trials", type=int, default=4, help="trials of the small-chunk arms (hundreds of thousands of handles each)")
ap.add_argument("--alloc-budget-s", type=float, default=15.0, help="projected seconds per trio above which the 'min' chunk is doubled")
ap.add_argument("--placed-trials", type=int, default=6, help="trials of the hm: / vw: arms (chosen virtual addresses)")
ap.add_argument("--bc7-gib", type=float, default=4.0)
ap.add_argument("--corpus-scale", type=float, default=1.0)
args = ap.parse_args()

_vmm_arms = [a for a in args.arms.split(args.arm_sep) if a != "hipmalloc" and not a.startswith("hm:")]
if (_vmm_arms or args.diagnose) and not args.allow_vmm:
    sys.exit(f"arms {_vmm_arms or ['--diagnose']} map memory through HIP's VMM, which returned wrong data and raised a GPU memory fault on this "
             "stack (profiles/r05_placement_vmm.txt items 1 and 5); pass --vmm-is-unsafe-here-and-i-run-it-anyway on a box of your own")
HELPER_SRC = os.path.join(ROOT, "tools", "vmm_helper.cpp")
if not os.path.exists(HELPER) or os.path.getmtime(HELPER) < os.path.getmtime(HELPER_SRC):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", HELPER_SRC, "-o", HELPER])

import torch  # noqa: E402

import bench  # noqa: E402
import dxt_lossless_transform_amd as pkg  # noqa: E402
from dxt_lossless_transform_amd import _lib, batch, bc7  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.init()
lib = _lib.load()
bc7._l()
h = C.CDLL(HELPER)
vp, sz, u64 = C.c_void_p, C.c_size_t, C.c_uint64
h.vmm_last_error.restype = C.c_char_p
h.vmm_granularity.argtypes = [C.c_int, C.POINTER(sz), C.POINTER(sz)]
h.vmm_alloc.argtypes = [C.c_int, sz, sz, u64, sz, C.POINTER(vp)]
h.vmm_free.argtypes = [vp]
h.vmm_copy.argtypes = [vp, vp, sz]
h.vmm_zero.argtypes = [vp, sz]
h.vmm_copy_by_kernel.argtypes = [vp, vp, sz]
h.vmm_fill_by_kernel.argtypes = [vp, sz, C.c_uint32]
h.vmm_differ_by_kernel.argtypes = [vp, vp, sz, C.POINTER(C.c_ulonglong)]
h.vmm_differ_by_kernel.restype = C.c_longlong
h.plain_alloc.argtypes = [C.c_int, sz, C.POINTER(vp)]
h.plain_free.argtypes = [vp]


def ck(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what}: {h.vmm_last_error().decode()}")


gmin, grec = sz(0), sz(0)
ck(h.vmm_granularity(0, C.byref(gmin), C.byref(grec)), "granularity")
gmin, grec = gmin.value, grec.value
min_chunk = max(gmin, args.min_chunk_kib << 10)

def _seconds_per_handle(chunk, total=256 << 20):
    p = vp(0)
    t0 = time.perf_counter()
    ck(h.vmm_alloc(0, total, chunk, 0, 0, C.byref(p)), "vmm_alloc (timing)")
    dt = time.perf_counter() - t0
    ck(h.vmm_free(p.value), "vmm_free (timing)")
    return dt / (total // chunk)


_largest = int(max(args.bc7_gib * (1 << 30), 8.5 * (1 << 30) * args.corpus_scale))
while min_chunk < (2 << 20) and any(a.startswith("small") for a in args.arms.split(args.arm_sep)):
    per = _seconds_per_handle(min_chunk)
    proj = per * 3 * (_largest // min_chunk)
    print(f"chunk {min_chunk} B: {per * 1e6:.1f} us per handle (create + map), projected {proj:.1f} s per trio of the largest workload", flush=True)
    if proj <= args.alloc_budget_s:
        break
    min_chunk *= 2
print(f"hipMemGetAllocationGranularity: minimum {gmin} B, recommended {grec} B; 'small' arms use {min_chunk} B chunks", flush=True)

ARMS = {  # name -> (chunk bytes or None for hipMalloc, shuffled)
    "hipmalloc": (None, False), "whole": (0, False), "2m": (2 << 20, False), "2m-shuf": (2 << 20, True),
    "small": (min_chunk, False), "small-shuf": (min_chunk, True),
}


def copy_in(dst, src, n):
    ck((h.vmm_copy_by_kernel if args.copy == "kernel" else h.vmm_copy)(dst, src, n), "copy")


def clear(dst, n):
    ck(h.vmm_fill_by_kernel(dst, n, 0) if args.copy == "kernel" else h.vmm_zero(dst, n), "clear")


def differ(a, b, n):
    first = C.c_ulonglong(0)
    d = h.vmm_differ_by_kernel(a, b, n, C.byref(first))
    if d < 0:
        raise RuntimeError(f"compare: {h.vmm_last_error().decode()}")
    return d, first.value


def parse_placed_arm(arm):
    """`hm:A:ox,oy,oz` -- hipMalloc of the buffer + A MiB + slack, the pointer rounded up to an A-MiB boundary, then ox / oy / oz MiB
    further in (x, y, z): the VIRTUAL address bits of every buffer are the probe's choice, the physical pages the driver's.
    `vw:A:ox,oy,oz` -- the same with one hipMemCreate handle behind a range reserved at an A-MiB boundary."""
    kind, align, offs = arm.split(":")
    return kind, int(float(align) * (1 << 20)), [int(float(o) * (1 << 20)) for o in offs.split(",")]


class Buf:
    def __init__(self, nbytes, arm, seed, which=0):
        p = vp(0)
        t0 = time.perf_counter()
        self.offset = 0
        if ":" in arm:
            kind, align, offs = parse_placed_arm(arm)
            off = offs[which]
            self.vmm = kind == "vw"
            if self.vmm:
                ck(h.vmm_alloc(0, nbytes + off, 0, 0, align, C.byref(p)), f"vmm_alloc({arm})")
                self.offset = off
            else:
                ck(h.plain_alloc(0, nbytes + align + off, C.byref(p)), "hipMalloc")
                self.offset = (-p.value) % align + off
        else:
            chunk, shuf = ARMS[arm]
            self.vmm = chunk is not None
            if self.vmm:
                ck(h.vmm_alloc(0, nbytes, chunk, seed if shuf else 0, 0, C.byref(p)), f"vmm_alloc({arm})")
            else:
                ck(h.plain_alloc(0, nbytes, C.byref(p)), "hipMalloc")
        self.alloc_s = time.perf_counter() - t0
        self.base, self.ptr, self.nbytes = p.value, p.value + self.offset, nbytes

    def free(self):
        if self.base:
            ck((h.vmm_free if self.vmm else h.plain_free)(self.base), "free")
            self.base = self.ptr = 0


def steady(fwd, inv, nbytes, steps=20):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        for _ in range(8):
            fwd(); inv()
        torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * steps + 1)]
    for i in range(steps):
        ev[2 * i].record(); fwd(); ev[2 * i + 1].record(); inv()
    ev[2 * steps].record()
    torch.cuda.synchronize()
    fw = sum(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(steps)) / steps
    iv = sum(ev[2 * i + 1].elapsed_time(ev[2 * i + 2]) for i in range(steps)) / steps
    return 2 * nbytes / (fw * 1e-3) / 8e12, 2 * nbytes / (iv * 1e-3) / 8e12


SPACERS_MIB = (0, 2, 6, 34, 130, 514, 1026, 2050, 0, 4098, 8194, 2, 18, 66, 258, 3074)


LIVE = []


def run_arm(name, arm, nbytes, source, make_calls, algorithmic=None):
    """`trials` fresh trios (x, y, z) of `nbytes` each under `arm`; `source`: a torch tensor holding x's bytes."""
    try:
        return _run_arm(name, arm, nbytes, source, make_calls, algorithmic or nbytes)
    except (RuntimeError, AssertionError) as e:
        print(f"== {name} {arm}: FAILED -- {e}", flush=True)
        torch.cuda.synchronize()
        for b in LIVE:
            b.free()
        del LIVE[:]
        return []


def _run_arm(name, arm, nbytes, source, make_calls, algorithmic):
    rows = []
    for k in range(args.min_trials if arm.startswith("small") else args.placed_trials if ":" in arm else args.trials):
        sp_mib = SPACERS_MIB[k % len(SPACERS_MIB)]
        spacer = Buf(sp_mib << 20, "hipmalloc" if arm == "hipmalloc" or arm.startswith("hm:") else "whole", 0) if sp_mib else None
        if spacer is not None:
            LIVE.append(spacer)
        trio = []
        for j in range(3):
            trio.append(Buf(nbytes, arm, 0xA110C000 + 16 * k + j, j))
            LIVE.append(trio[-1])
        x, y, z = trio
        copy_in(x.ptr, source.data_ptr(), nbytes)
        clear(y.ptr, nbytes)
        clear(z.ptr, nbytes)
        fwd, inv = make_calls(x.ptr, y.ptr, z.ptr)
        fw, iv = steady(fwd, inv, algorithmic)
        bad, first = differ(z.ptr, source.data_ptr(), nbytes)
        exact = bad == 0
        if not exact:
            print(f"   round trip: {bad} of {nbytes // 16} 16-byte vectors differ, the first at byte {16 * first:#x}", flush=True)
        rows.append((fw, iv))
        print(f"{name:7s} {arm:14s} trial {k:2d} spacer {sp_mib:5d} MiB  x {x.ptr:#x} y-x {(y.ptr - x.ptr) / 2**20:10.1f} MiB z-y {(z.ptr - y.ptr) / 2**20:10.1f} MiB"
              f"  alloc {x.alloc_s + y.alloc_s + z.alloc_s:6.2f} s  fwd {fw:.4f} inv {iv:.4f}  roundtrip {'exact' if exact else 'WRONG'}", flush=True)
        for b in LIVE:
            b.free()
        del LIVE[:]
        assert exact, "round trip differs"
    f = [r[0] for r in rows]
    i = [r[1] for r in rows]
    print(f"== {name} {arm}: fwd {min(f):.4f}..{max(f):.4f} (spread {max(f) - min(f):.4f}, mean {sum(f) / len(f):.4f})   "
          f"inv {min(i):.4f}..{max(i):.4f} (spread {max(i) - min(i):.4f}, mean {sum(i) / len(i):.4f})", flush=True)
    return rows


stream = lambda: torch.cuda.current_stream().cuda_stream
summary = {}

if args.diagnose:
    # Round 5's first run of this probe saw a WRONG round trip on the second trio of the `whole` arm with hipMemcpy / hipMemset
    # moving the data.  Which step was it?  Same sequence, every step checked by a kernel: data in by hipMemcpy and by kernel,
    # the copy compared with its source; then a plain kernel copy x -> y -> z through the mappings and z against the source.
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, device=dev)
    pkg.fill_splitmix64(src, 0xD1A6)
    torch.cuda.synchronize()
    for arm in ("whole", "2m"):
        for k in range(4):
            sp = Buf((2 + 4 * k) << 20, "whole", 0)
            x, y, z = (Buf(n, arm, 0) for _ in range(3))
            ck(h.vmm_copy(x.ptr, src.data_ptr(), n), "hipMemcpy in")
            d_memcpy = differ(x.ptr, src.data_ptr(), n)
            ck(h.vmm_zero(y.ptr, n), "hipMemset")
            ck(h.vmm_copy_by_kernel(z.ptr, y.ptr, n), "kernel copy")
            zeros = torch.zeros(n, dtype=torch.uint8, device=dev)
            d_memset = differ(z.ptr, zeros.data_ptr(), n)
            del zeros
            ck(h.vmm_copy_by_kernel(x.ptr, src.data_ptr(), n), "kernel copy in")
            ck(h.vmm_copy_by_kernel(y.ptr, x.ptr, n), "kernel copy")
            ck(h.vmm_copy_by_kernel(z.ptr, y.ptr, n), "kernel copy")
            d_kernel = differ(z.ptr, src.data_ptr(), n)
            back = torch.empty(n, dtype=torch.uint8, device=dev)
            ck(h.vmm_copy(back.data_ptr(), z.ptr, n), "hipMemcpy out")
            d_out = int((back != src).sum().item())
            del back
            print(f"diagnose {arm:6s} trio {k}: x {x.ptr:#x}  hipMemcpy-in differs in {d_memcpy[0]} vectors (first byte {16 * d_memcpy[1]:#x}), "
                  f"hipMemset leaves {d_memset[0]} non-zero vectors, kernel copies x->y->z differ in {d_kernel[0]}, hipMemcpy-out differs in {d_out} bytes", flush=True)
            for b in (x, y, z, sp):
                b.free()
    del src
    torch.cuda.empty_cache()

if "bc7" in args.workloads.split(","):
    n = int(args.bc7_gib * (1 << 30)) // 16384 * 16384
    src = torch.empty(n, dtype=torch.uint8, device=dev)
    pkg.fill_splitmix64(src, 0x0BC70004)
    bench.bc7_force_modes_device(torch, src, "uniform")
    torch.cuda.synchronize()

    def bc7_calls(x, y, z):
        def fwd():
            rc = lib.dxtlt_transform_bc7_device(x, y, n, None, 0, stream())
            assert rc == 0, _lib.last_error()

        def inv():
            rc = lib.dxtlt_untransform_bc7_device(y, z, n, None, 0, stream())
            assert rc == 0, _lib.last_error()
        return fwd, inv

    for arm in args.arms.split(args.arm_sep):
        summary[("bc7", arm)] = run_arm("bc7", arm, n, src, bc7_calls)
    del src
    torch.cuda.empty_cache()

if "corpus" in args.workloads.split(","):
    fmt, B = "bc3", 16
    st = pkg.Bc3TransformSettings()
    texs = bench.corpus_textures(args.corpus_scale)[::2]
    offs, arena = bench.corpus_layout(texs, B, 256)
    src = torch.empty(arena, dtype=torch.uint8, device=dev)
    pkg.fill_splitmix64(src, 0xC0A90000 + B, 0)
    for (w, hh, blocks), o in zip(texs, offs):
        end = o + blocks * B
        src[end:(end + 255) // 256 * 256].zero_()
    torch.cuda.synchronize()
    fid, block, mode, sa, sc = batch._item_fields(fmt, st)
    lib.dxtlt_transform_batch_device.argtypes = [C.POINTER(batch.DxtltBatchItem), C.c_size_t, C.c_void_p]
    lib.dxtlt_transform_batch_device.restype = C.c_int32

    def items(inp, out, inverse):
        arr = (batch.DxtltBatchItem * len(texs))()
        for k, ((_, _, blocks), o) in enumerate(zip(texs, offs)):
            arr[k].d_input, arr[k].d_output, arr[k].len = inp + o, out + o, blocks * B
            arr[k].format, arr[k].inverse, arr[k].decorrelation_mode = fid, int(inverse), mode
            arr[k].split_alpha_endpoints, arr[k].split_colour_endpoints = int(sa), int(sc)
        return arr

    def corpus_calls(x, y, z):
        fi, ii = items(x, y, False), items(y, z, True)

        def fwd():
            rc = lib.dxtlt_transform_batch_device(fi, len(fi), stream())
            assert rc == 0, _lib.last_error()

        def inv():
            rc = lib.dxtlt_transform_batch_device(ii, len(ii), stream())
            assert rc == 0, _lib.last_error()
        return fwd, inv

    for arm in args.arms.split(args.arm_sep):
        summary[("corpus_bc3", arm)] = run_arm("corpus", arm, arena, src, corpus_calls, sum(t[2] for t in texs) * B)

print("\nsummary (fraction of 8 TB/s; min..max over the trials, spread)")
for (wl, arm), rows in summary.items():
    if not rows:
        print(f"{wl:11s} {arm:9s} failed")
        continue
    f = [r[0] for r in rows]
    i = [r[1] for r in rows]
    print(f"{wl:11s} {arm:14s} fwd {min(f):.4f}..{max(f):.4f} ({max(f) - min(f):.4f})   inv {min(i):.4f}..{max(i):.4f} ({max(i) - min(i):.4f})")
