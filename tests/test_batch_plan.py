"""Host logic of the batch call, checked without a GPU through the dxtlt_debug_plan_batch test hook: how many workgroups a
buffer gets and in which tile form (an independent statement of the rules below, written from DESIGN.md section 3 and the
reference's stream layouts -- bc1 transform_with_settings.rs:31-72, bc3 :32-142), and that the two-level workgroup -> entry
index finds the owner of every workgroup, tiny buffers between huge ones included."""
import ctypes as C

import numpy as np
import pytest

TILE_BYTES = {1: 4096, 2: 4096, 3: 4096}   # one 256-lane tile (csrc/bcn_device.h, shift_tile_threads)


def tile_bytes(fmt, sc, inverse):
    """csrc/bcn_device.h batch_tile_threads: 256 lanes x 16 bytes -- except the FORWARD launch of BC1 WITHOUT the colour split, whose
    tiles have 128 lanes (two 2 KiB store runs per tile are the shape the write path dislikes: profiles/r05_bc1_nosplit.txt)"""
    return 2048 if (fmt == 1 and not sc and not inverse) else TILE_BYTES[fmt]


class Planned(C.Structure):
    _fields_ = [("first_wg", C.c_uint32), ("end_wg", C.c_uint32), ("full_tiles", C.c_uint32), ("form", C.c_uint8),
                ("halo_vecs", C.c_uint8), ("shift", C.c_uint8 * 6), ("gbase", C.c_uint64 * 6)]


def streams(fmt, sa, sc):
    """(offset, width) in bytes per block of every stream of the transformed layout, in order"""
    out, off = [], 0
    if fmt == 3:
        for w in ((1, 1) if sa else (2,)):
            out.append((off, w)); off += w
        out.append((off, 6)); off += 6
    if fmt == 2:
        out.append((off, 8)); off += 8
    for w in ((2, 2) if sc else (4,)):
        out.append((off, w)); off += w
    out.append((off, 4))
    return out


def expected(fmt, inverse, sa, sc, src, dst, blocks):
    """the rules: aligned tiles when every stream base sits on a 128-byte line, else halo tiles (windows on 64-byte sectors)
    forward / shifted tiles (base modulo 16) inverse; an edge tile for the blocks behind the last whole tile and, forward, for
    the stream tails the moved-back windows leave out; None when a shift is no multiple of its stream's element width"""
    B = 8 if fmt == 1 else 16
    T = tile_bytes(fmt, sc, inverse) // B
    soa = src if inverse else dst
    mask = 15 if inverse else 63
    bases = [soa + off * blocks for off, _ in streams(fmt, sa, sc)]
    shifts = [b & mask for b in bases]
    for (off, w), d in zip(streams(fmt, sa, sc), shifts):
        if d % (2 if w == 6 else w):
            return None
    tiles, rest = divmod(blocks, T)
    edge = rest != 0 or (not inverse and any(shifts))
    halo_blocks = max((d + w - 1) // w for (off, w), d in zip(streams(fmt, sa, sc), shifts))
    return {"wgs": tiles + (1 if edge else 0), "full_tiles": tiles, "form": int(all(b % 128 == 0 for b in bases)),
            "shifts": shifts, "halo_vecs": 0 if inverse else (halo_blocks + 16 // B - 1) // (16 // B)}


def plan(lib, fmt, inverse, variant, sa, sc, srcs, dsts, blocks):
    n = len(blocks)
    a = lambda v: (C.c_uint64 * n)(*v)
    out = (Planned * max(n, 1))()
    cap = 1 << 20
    index = (C.c_uint8 * cap)()
    lib.dxtlt_debug_plan_batch.restype = C.c_uint32
    lib.dxtlt_debug_plan_batch.argtypes = [C.c_int32] * 5 + [C.c_void_p] * 3 + [C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    wide = C.c_uint32(7)
    total = lib.dxtlt_debug_plan_batch(fmt, int(inverse), variant, int(sa), int(sc), a(srcs), a(dsts), a(blocks), n, out, index, cap,
                                       C.byref(wide))
    plan.last_wide = wide.value
    return total, list(out)[:n], np.frombuffer(index, dtype=np.uint8)


def owner_by_index(index, total, owners, wg, wide):
    """the kernel's lookup: base[wg / 4096] + delta[wg / 64] owns workgroup 64 * (wg / 64); when that entry ends at or before wg,
    bisection over the end_wg of the next wg % 64 entries.  Returns (entry, dependent loads of the search)."""
    n_base = (total + 4095) // 4096
    base = index[: 4 * n_base].view(np.uint32)
    delta = index[4 * n_base:].view(np.uint16) if wide else index[4 * n_base:]
    e = int(base[wg >> 12]) + int(delta[wg >> 6])
    steps = 0
    if owners[e].end_wg <= wg:
        lo, hi = e + 1, min(e + (wg & 63), len(owners) - 1)
        while lo < hi:
            mid = (lo + hi) >> 1
            steps += 1
            if owners[mid].end_wg <= wg:
                lo = mid + 1
            else:
                hi = mid
        e = lo
    return e, steps


@pytest.fixture(scope="module")
def lib(pkg):
    return C.CDLL(pkg._lib.lib_path())


@pytest.mark.parametrize("fmt,sa,sc", [(1, 0, 1), (1, 0, 0), (2, 0, 1), (2, 0, 0), (3, 1, 1), (3, 0, 1), (3, 1, 0), (3, 0, 0)])
@pytest.mark.parametrize("inverse", [False, True])
def test_plan_follows_the_tile_rules(lib, fmt, sa, sc, inverse):
    rng = np.random.default_rng(1000 * fmt + 10 * sa + sc + 7 * inverse)
    B = 8 if fmt == 1 else 16
    T = tile_bytes(fmt, sc, inverse) // B
    blocks = [0, 1, T - 1, T, T + 1, 8 * T, 64 * T - 1, 64 * T, 64 * T + 23] + [int(x) for x in rng.integers(1, 300 * T, 40)]
    # mip-chain counts, counts that keep every stream on its line, and transformed-side pointers off by 0 .. 120 bytes
    blocks += [(4 ** k - 1) // 3 for k in range(4, 11)] + [T * 7 * 32, T * 3 * 64]
    srcs, dsts, at = [], [], 0x7F00_0000_0000
    for i, b in enumerate(blocks):
        lead = [0, 8, 16, 64, 128, 24, 120][i % 7]
        srcs.append(at + (lead if inverse else 0))
        dsts.append(at + (1 << 36) + (0 if inverse else lead))
        at += (b * B + 255 + 256) // 256 * 256
    total, got, _ = plan(lib, fmt, inverse, 1, sa, sc, srcs, dsts, blocks)
    assert total != 0xFFFFFFFF
    at_wg = 0
    for b, s, d, g in zip(blocks, srcs, dsts, got):
        want = expected(fmt, inverse, sa, sc, s, d, b) if b else {"wgs": 0}
        assert want is not None
        assert (g.first_wg, g.end_wg) == (at_wg, at_wg + want["wgs"]), (b, hex(s), hex(d))
        at_wg += want["wgs"]
        if b == 0:
            continue
        assert g.full_tiles == want["full_tiles"] and g.form == want["form"], (b, hex(s), hex(d))
        n_streams = len(want["shifts"])
        assert list(g.shift)[:n_streams] == want["shifts"]
        assert g.halo_vecs == want["halo_vecs"]
        for (off, w), sh, gb in zip(streams(fmt, sa, sc), want["shifts"], list(g.gbase)):
            assert gb == (off * b - sh) % 2 ** 64
    assert total == at_wg


def test_buffers_the_batch_kernel_does_not_take_are_reported(lib):
    # a transformed-side pointer off by 2 bytes: the 4-byte index stream's shift is no multiple of its element width
    total, _, _ = plan(lib, 1, False, 1, 0, 1, [0x7F0000000000], [0x7F1000000002], [4096])
    assert total == 0xFFFFFFFF
    total, _, _ = plan(lib, 3, True, 1, 1, 1, [0x7F0000000001], [0x7F1000000000], [4096])
    assert total == 0xFFFFFFFF
    # 1-byte streams take any shift, but BC3's 2-byte ones do not take an odd one: an odd block count behind an aligned pointer
    # keeps every base a multiple of its width (N, 2N, 8N, 10N, 12N)
    total, got, _ = plan(lib, 3, False, 1, 1, 1, [0x7F0000000000], [0x7F1000000000], [4097])
    assert total == 17 and got[0].form == 0


@pytest.mark.parametrize("inverse", [False, True])
def test_index_finds_the_owner_of_every_workgroup(lib, inverse):
    """huge buffers (many 4096-workgroup spans each), runs of hundreds of one-workgroup buffers (more than 255 entries begin
    inside one span: the byte saturates and the kernel walks), empty buffers in between"""
    rng = np.random.default_rng(5 + inverse)
    T = 256
    blocks = []
    for _ in range(6):
        blocks.append(int(rng.integers(5000, 40000)) * T + int(rng.integers(0, T)))
        blocks += [int(x) for x in rng.integers(1, 3 * T, int(rng.integers(200, 700)))]
        blocks += [0, 0]
        blocks += [int(x) for x in rng.integers(60 * T, 70 * T, 5)]
    srcs, dsts, at = [], [], 0x7E00_0000_0000
    for b in blocks:
        srcs.append(at); dsts.append(at + (1 << 38)); at += (b * 16 + 511) // 256 * 256
    total, got, index = plan(lib, 3, inverse, 1, 1, 1, srcs, dsts, blocks)
    assert total not in (0, 0xFFFFFFFF)
    owners = [g for g in got if g.end_wg > g.first_wg]   # the table holds the buffers that own workgroups, in order
    assert owners[-1].end_wg == total
    truth = np.zeros(total, dtype=np.int64)
    for i, g in enumerate(owners):
        truth[g.first_wg:g.end_wg] = i
    sample = sorted(set(range(0, total, 61)) | set(range(min(total, 20000))) | {total - 1} | {g.first_wg for g in owners} |
                    {g.end_wg - 1 for g in owners})
    assert plan.last_wide == 1, "the case is meant to need the wide index (more than 255 entries begin inside one span)"
    worst = 0
    for wg in sample:
        e, steps = owner_by_index(index, total, owners, wg, True)
        assert e == truth[wg], wg
        worst = max(worst, steps)
    assert 1 <= worst <= 6, worst      # bisection over at most 63 candidates


@pytest.mark.parametrize("inverse", [False, True])
def test_index_stays_narrow_for_a_corpus_and_goes_wide_for_thousands_of_tiny_buffers(lib, inverse):
    """A corpus of mip-chained textures (hundreds to thousands of workgroups each) keeps the byte index -- one cache line per 4096
    workgroups; 70 000 buffers of one to three tiles each switch it to 16-bit deltas instead of saturating (round 4: a walk of
    up to ~3800 dependent entry loads per workgroup).  Every workgroup's owner is found in at most six search steps."""
    import os
    import sys

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    texs = bench.corpus_textures(0.2)
    blocks = [t[2] for t in texs]
    srcs, dsts, at = [], [], 0x7D00_0000_0000
    for b in blocks:
        srcs.append(at); dsts.append(at + (1 << 38)); at += (b * 8 + 255) // 256 * 256
    total, got, index = plan(lib, 1, inverse, 1, 0, 1, srcs, dsts, blocks)
    assert total not in (0, 0xFFFFFFFF) and plan.last_wide == 0
    owners = [g for g in got if g.end_wg > g.first_wg]
    for wg in list(range(0, total, 997)) + [g.first_wg for g in owners] + [g.end_wg - 1 for g in owners]:
        e, steps = owner_by_index(index, total, owners, wg, False)
        assert owners[e].first_wg <= wg < owners[e].end_wg and steps <= 6

    rng = np.random.default_rng(77 + inverse)
    blocks = [int(x) for x in rng.integers(1, 3 * 512, 70000)]      # BC1: 512 blocks per tile
    srcs, dsts, at = [], [], 0x7C00_0000_0000
    for b in blocks:
        srcs.append(at); dsts.append(at + (1 << 38)); at += (b * 8 + 255) // 256 * 256
    total, got, index = plan(lib, 1, inverse, 1, 0, 1, srcs, dsts, blocks)
    assert total not in (0, 0xFFFFFFFF) and plan.last_wide == 1
    owners = [g for g in got if g.end_wg > g.first_wg]
    truth = np.zeros(total, dtype=np.int64)
    for i, g in enumerate(owners):
        truth[g.first_wg:g.end_wg] = i
    worst = 0
    for wg in range(0, total, 7):
        e, steps = owner_by_index(index, total, owners, wg, True)
        assert e == truth[wg], wg
        worst = max(worst, steps)
    assert worst <= 6
