"""Seeded random campaign, GPU against the oracle, for a time budget: whole-buffer calls and RANGE calls of BC1/2/3 with
random settings, block counts (tiny, around tile multiples, up to a few million), first blocks, and byte offsets 0..191
of both device pointers; BC7 whole buffers and granule ranges; batch calls over 1..40 buffers of mixed formats (BC7
included), directions and pointer offsets.  Guard bytes around every output.  Prints one line per
100 cases and the failing case's parameters (re-runnable: `python tools/fuzz_gpu.py --seed S --only CASE`).
Test tooling: imports oracle/.   usage: python tools/fuzz_gpu.py [--seconds 300] [--seed 1]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import dxt_lossless_transform_amd as pkg  # noqa: E402
from dxt_lossless_transform_amd import bc7 as bc7mod  # noqa: E402
from oracle import oracle_c  # noqa: E402

BLOCK = {"bc1": 8, "bc2": 16, "bc3": 16}
TILE = {"bc1": 512, "bc2": 256, "bc3": 256}


def settings_of(fmt, rng):
    v, sc, sa = int(rng.integers(0, 4)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    if fmt == "bc3":
        return (v, sa, sc), pkg.Bc3TransformSettings(pkg.YCoCgVariant(v), sa, sc)
    cls = pkg.Bc1TransformSettings if fmt == "bc1" else pkg.Bc2TransformSettings
    return (v, False, sc), cls(pkg.YCoCgVariant(v), sc)


def block_count(fmt, rng):
    k = int(rng.integers(0, 10))
    t = TILE[fmt]
    if k < 3:
        return int(rng.integers(0, 200))
    if k < 6:
        return max(0, int(rng.integers(1, 40)) * t + int(rng.integers(-70, 71)))
    if k < 9:
        return int(rng.integers(0, 60_000))
    return int(rng.integers(500_000, 3_000_000))


def guard_ok(h, lo, n, fill):
    return bool((h[:lo] == fill).all() and (h[lo + n:] == fill).all())


def bcn_case(case, rng, dev):
    fmt = ("bc1", "bc2", "bc3")[int(rng.integers(0, 3))]
    (v, sa, sc), st = settings_of(fmt, rng)
    n = block_count(fmt, rng)
    B = BLOCK[fmt]
    a, b = int(rng.integers(0, 192)), int(rng.integers(0, 192))
    if rng.integers(0, 4) == 0:
        a = 0
    if rng.integers(0, 4) == 0:
        b = 0
    x = oracle_c.fill_splitmix64(n * B, 0xF00D + case)
    want = oracle_c.transform(fmt, x, v, sc, sa)
    xd = torch.full((n * B + 256,), 0x11, dtype=torch.uint8, device=dev)
    xd[a:a + n * B] = torch.from_numpy(x).to(dev)
    yd = torch.full((n * B + 256,), 0x22, dtype=torch.uint8, device=dev)
    zd = torch.full((n * B + 256,), 0x33, dtype=torch.uint8, device=dev)
    ranged = bool(rng.integers(0, 2)) and n > 0
    tag = dict(case=case, fmt=fmt, settings=(v, sa, sc), n=n, a=a, b=b, ranged=ranged)
    if not ranged:
        getattr(pkg, f"transform_{fmt}_with_settings")(xd[a:a + n * B], yd[b:b + n * B], st)
        getattr(pkg, f"untransform_{fmt}_with_settings")(yd[b:b + n * B], zd[a:a + n * B], st)
    else:
        # cut [0, n) at random places; every piece is one range call, in a random order
        cuts = sorted(set([0, n] + [int(c) for c in rng.integers(0, n + 1, size=int(rng.integers(1, 5)))]))
        pieces = list(zip(cuts, cuts[1:]))
        tag["cuts"] = cuts
        for i in rng.permutation(len(pieces)):
            lo, hi = pieces[int(i)]
            pkg.transform_range(fmt, False, xd[a + lo * B:a + n * B], yd[b:b + n * B], n, lo, hi - lo, st)
        torch.cuda.synchronize()
        for i in rng.permutation(len(pieces)):
            lo, hi = pieces[int(i)]
            pkg.transform_range(fmt, True, yd[b:b + n * B], zd[a + lo * B:a + n * B], n, lo, hi - lo, st)
    torch.cuda.synchronize()
    yh, zh = yd.cpu().numpy(), zd.cpu().numpy()
    ok = (np.array_equal(yh[b:b + n * B], want) and np.array_equal(zh[a:a + n * B], x)
          and guard_ok(yh, b, n * B, 0x22) and guard_ok(zh, a, n * B, 0x33))
    return ok, tag


def bc7_blocks(n, rng, case):
    x = oracle_c.fill_splitmix64(n * 16, 0xB7 + case).copy()
    kind = int(rng.integers(0, 4))
    if kind == 3:
        return x                                   # raw bytes: modes by trailing zeros, reserved encoding included
    v = x.reshape(-1, 16)
    if kind == 0:
        modes = rng.integers(0, 8, size=n)
    elif kind == 1:
        modes = rng.choice(8, size=n, p=[.02, .25, .02, .13, .02, .02, .52, .02])
    else:
        modes = np.repeat(rng.integers(0, 8, size=n // 97 + 1), 97)[:n]
    m = modes.astype(np.int64)
    v[:, 0] = (((v[:, 0].astype(np.int64) & ~((2 << m) - 1)) | (1 << m)) & 0xFF).astype(np.uint8)
    return x


def bc7_case(case, rng, dev):
    k = int(rng.integers(0, 4))
    n = int(rng.integers(0, 3000)) if k < 2 else int(rng.integers(1, 30)) * 1024 + int(rng.integers(-3, 4)) if k == 2 \
        else int(rng.integers(100_000, 1_500_000))
    n = max(n, 0)
    a, b = int(rng.integers(0, 64)), int(rng.integers(0, 64))
    x = bc7_blocks(n, rng, case)
    want = oracle_c.transform_bc7(x) if n else x
    xd = torch.full((n * 16 + 128,), 0x11, dtype=torch.uint8, device=dev)
    xd[a:a + n * 16] = torch.from_numpy(x).to(dev)
    yd = torch.full((n * 16 + 128,), 0x22, dtype=torch.uint8, device=dev)
    zd = torch.full((n * 16 + 128,), 0x33, dtype=torch.uint8, device=dev)
    ranged = bool(rng.integers(0, 2)) and n >= 2048
    tag = dict(case=case, fmt="bc7", n=n, a=a, b=b, ranged=ranged)
    if not ranged:
        bc7mod.transform_bc7(xd[a:a + n * 16], yd[b:b + n * 16])
        bc7mod.untransform_bc7(yd[b:b + n * 16], zd[a:a + n * 16])
    else:
        g = n // 1024
        cuts = sorted(set([0, n] + [1024 * int(c) for c in rng.integers(0, g + 1, size=int(rng.integers(1, 4)))]))
        tag["cuts"] = cuts
        for lo, hi in zip(cuts, cuts[1:]):
            bc7mod.transform_bc7_range(False, xd[a + lo * 16:a + n * 16], yd[b:b + n * 16], n, lo, hi - lo)
        torch.cuda.synchronize()
        for lo, hi in zip(cuts, cuts[1:]):
            bc7mod.transform_bc7_range(True, yd[b:b + n * 16], zd[a + lo * 16:a + n * 16], n, lo, hi - lo)
    torch.cuda.synchronize()
    yh, zh = yd.cpu().numpy(), zd.cpu().numpy()
    ok = (np.array_equal(yh[b:b + n * 16], want) and np.array_equal(zh[a:a + n * 16], x)
          and guard_ok(yh, b, n * 16, 0x22) and guard_ok(zh, a, n * 16, 0x33))
    return ok, tag


def batch_case(case, rng, dev):
    """one dxtlt_transform_batch_device call over 1..40 buffers: formats BC1-3 and BC7 mixed, both directions, block
    counts 0..20 000, every buffer at its own byte offset 0..63 on both sides"""
    from dxt_lossless_transform_amd import batch

    k = int(rng.integers(1, 41))
    items, checks = [], []
    for i in range(k):
        fmt = ("bc1", "bc2", "bc3", "bc7")[int(rng.integers(0, 4))]
        n = int(rng.integers(0, 20_001)) if rng.integers(0, 4) else int(rng.integers(0, 30))
        inverse = bool(rng.integers(0, 2))
        a, b = int(rng.integers(0, 64)), int(rng.integers(0, 64))
        if fmt == "bc7":
            B, st = 16, None
            x = bc7_blocks(n, rng, case * 64 + i)
            if inverse and n:
                x = oracle_c.transform_bc7(x)
            want = oracle_c.transform_bc7(x, inverse=inverse) if n else x
            tag_s = None
        else:
            B = BLOCK[fmt]
            (v, sa, sc), st = settings_of(fmt, rng)
            x = oracle_c.fill_splitmix64(n * B, 0xBA7 + case * 64 + i)
            want = oracle_c.transform(fmt, x, v, sc, sa, inverse=inverse)
            tag_s = (v, sa, sc)
        xd = torch.full((n * B + 128,), 0x11, dtype=torch.uint8, device=dev)
        xd[a:a + n * B] = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        yd = torch.full((n * B + 128,), 0x22, dtype=torch.uint8, device=dev)
        items.append((fmt, inverse, xd[a:a + n * B], yd[b:b + n * B], st))
        checks.append((want, yd, b, n * B, (fmt, inverse, n, a, b, tag_s)))
    batch.transform_batch(items)
    torch.cuda.synchronize()
    for want, yd, b, nb, tag in checks:
        h = yd.cpu().numpy()
        if not (np.array_equal(h[b:b + nb], want) and guard_ok(h, b, nb, 0x22)):
            return False, dict(case=case, kind="batch", item=tag, ranged=False)
    return True, dict(case=case, kind="batch", ranged=False)


def uniform_batch_case(case, rng, dev):
    """one batch call over 2..24 buffers of ONE format, size, direction and settings (the kernel's equal-size lookups): laid
    out as a regular array (stride >= size, sometimes in reverse order, sometimes so that every stream base sits on a
    128-byte line: the tiled-kernel launch) or at scattered offsets of one arena; guard bytes between the outputs"""
    from dxt_lossless_transform_amd import batch

    fmt = ("bc1", "bc2", "bc3")[int(rng.integers(0, 3))]
    B, t = BLOCK[fmt], TILE[fmt]
    k = int(rng.integers(2, 25))
    kind = int(rng.integers(0, 4))
    n = int(rng.integers(1, 40)) * (t // 2) if kind == 0 else block_count(fmt, rng) % 40_000   # kind 0: whole tiles, aligned
    inverse = bool(rng.integers(0, 2))
    (v, sa, sc), st = settings_of(fmt, rng)
    pad = (0, 128, 4352, 16 * int(rng.integers(0, 300)))[int(rng.integers(0, 4))] if kind != 3 else 0
    stride = n * B + pad
    order = list(range(k))
    if kind == 2:
        order.reverse()
    offs = [o * stride for o in order]
    if kind == 3:       # scattered: distinct, non-uniform gaps
        at, offs = 0, []
        for _ in range(k):
            at += int(rng.integers(0, 5)) * 16
            offs.append(at)
            at += n * B
        rng.shuffle(offs)
        offs = [int(o) for o in offs]
    total = max(offs) + n * B + 256 if offs else 256
    a, b = (0, 0) if kind == 0 else (int(rng.integers(0, 4)) * 8, int(rng.integers(0, 4)) * 8)
    xd = torch.full((total,), 0x11, dtype=torch.uint8, device=dev)
    yd = torch.full((total,), 0x22, dtype=torch.uint8, device=dev)
    items, wants = [], []
    for i in range(k):
        x = oracle_c.fill_splitmix64(n * B, 0xA77 + case * 64 + i)
        if inverse:
            x = oracle_c.transform(fmt, x, v, sc, sa)
        wants.append(oracle_c.transform(fmt, x, v, sc, sa, inverse=inverse))
        lo = offs[i]
        xd[a + lo:a + lo + n * B] = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        items.append((fmt, inverse, xd[a + lo:a + lo + n * B], yd[b + lo:b + lo + n * B], st))
    batch.transform_batch(items)
    torch.cuda.synchronize()
    h = yd.cpu().numpy()
    covered = np.zeros(total, dtype=bool)
    for i in range(k):
        lo = b + offs[i]
        if not np.array_equal(h[lo:lo + n * B], wants[i]):
            return False, dict(case=case, kind="uniform_batch", item=(fmt, inverse, n, kind, pad, i, (v, sa, sc)), ranged=False)
        covered[lo:lo + n * B] = True
    if not (h[~covered] == 0x22).all():
        return False, dict(case=case, kind="uniform_batch", item=(fmt, inverse, n, kind, pad, "guard", (v, sa, sc)), ranged=False)
    return True, dict(case=case, kind="uniform_batch", ranged=False)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--batch-only", action="store_true", help="batch cases only")
    args = ap.parse_args()
    pkg.load()
    dev = torch.device("cuda:0")
    t0 = time.time()
    case = 0
    counts = {"bcn": 0, "bcn_ranged": 0, "bc7": 0, "bc7_ranged": 0, "batch": 0, "uniform_batch": 0}
    while time.time() - t0 < args.seconds:
        rng = np.random.default_rng([args.seed, case])
        if args.only >= 0 and case != args.only:
            case += 1
            continue
        pick = int(rng.integers(0, 16))
        is7, is_batch = pick < 4 and not args.batch_only, pick == 15 or args.batch_only
        is_uniform = is_batch and bool(rng.integers(0, 2))
        ok, tag = (uniform_batch_case if is_uniform else batch_case if is_batch else bc7_case if is7 else bcn_case)(case, rng, dev)
        counts["uniform_batch" if is_uniform else "batch" if is_batch else ("bc7" if is7 else "bcn") + ("_ranged" if tag["ranged"] else "")] += 1
        if not ok:
            print("FAIL", tag, flush=True)
            sys.exit(1)
        case += 1
        if case % 100 == 0:
            print(f"{case} cases ok, {time.time() - t0:.0f} s, {counts}", flush=True)
        if args.only >= 0:
            break
    print(f"done: {case} cases, all exact, seed {args.seed}, {counts}", flush=True)


if __name__ == "__main__":
    main()
