"""Host logic of the single-buffer call, checked without a GPU through the dxtlt_debug_plan_transform test hook: which launches a
range takes -- aligned tiles, forward halo tiles, inverse shifted tiles, edge tiles -- with how many workgroups of how many lanes and
which stream shifts.  The expectation below is an independent statement of the rules, written from DESIGN.md section 3 and the
reference's stream layouts (bc1 transform_with_settings.rs:31-72, bc2 :30-73, bc3 :32-142); the GPU parity tests check what the
launches compute, this checks that the right ones are chosen for every alignment class, size and settings combination."""
import ctypes as C

import numpy as np
import pytest

BLOCK = {1: 8, 2: 16, 3: 16}


class Launch(C.Structure):
    _fields_ = [("kind", C.c_int32), ("threads", C.c_int32), ("workgroups", C.c_uint32), ("full_tiles", C.c_uint32),
                ("range_blocks", C.c_uint64), ("aos_offset", C.c_uint64), ("shift", C.c_uint8 * 6), ("halo_vecs", C.c_uint8),
                ("natural", C.c_uint8), ("gbase", C.c_uint64 * 6)]


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


def aligned_lanes(fmt):
    return 128 if fmt == 1 else 256          # profiles/r01_q_tile_threads_sweep_sc1_stores.txt


def shift_lanes(fmt, inverse, sc):
    return 128 if (fmt == 1 and not inverse and not sc) else 256     # profiles/r05_bc1_nosplit.txt


def shifts_record(fmt, sa, sc, inverse, soa, total, first):
    S = streams(fmt, sa, sc)
    mask = 15 if inverse else 63
    bases = [soa + off * total + w * first for off, w in S]
    d = [b & mask for b in bases]
    natural = all(x % (2 if w == 6 else w) == 0 for (off, w), x in zip(S, d))
    per_vec = 16 // BLOCK[fmt]
    halo_blocks = max((x + w - 1) // w for (off, w), x in zip(S, d))
    return {"shift": d, "natural": int(natural), "halo_vecs": 0 if inverse else (halo_blocks + per_vec - 1) // per_vec,
            "gbase": [(off * total + w * first - x) % 2**64 for (off, w), x in zip(S, d)]}


def expected(fmt, inverse, sa, sc, src, dst, total, first, num, tile_threads=0, force=0):
    """the launches, in order, as dicts"""
    B = BLOCK[fmt]
    soa = src if inverse else dst
    out = []
    for piece in range(0, num, 1 << 31):
        n = min(1 << 31, num - piece)
        f = first + piece
        aos0 = piece * B
        bases = [soa + off * total + w * f for off, w in streams(fmt, sa, sc)]
        if any(b % 128 for b in bases) or (force & 3) == 2:
            lanes = shift_lanes(fmt, inverse, sc)
            T = lanes * 16 // B
            tiles, rest = divmod(n, T)
            rec = shifts_record(fmt, sa, sc, inverse, soa, total, f)
            if force & 0x20:
                rec["natural"] = 0
            tail = rest > 0 or (not inverse and any(rec["shift"]))
            out.append(dict(kind=2 if inverse else 1, threads=lanes, workgroups=tiles + int(tail), full_tiles=tiles, range_blocks=n,
                            aos_offset=aos0, **rec))
            continue
        lanes = tile_threads or aligned_lanes(fmt)
        T = lanes * 16 // B
        tiles = n // T
        if tiles:
            out.append(dict(kind=0, threads=lanes, workgroups=tiles, full_tiles=tiles, range_blocks=tiles * T, aos_offset=aos0))
        e_lanes = shift_lanes(fmt, inverse, sc)
        Te = e_lanes * 16 // B
        at = tiles * T
        while at < n:
            rec = shifts_record(fmt, sa, sc, inverse, soa, total, f + at)
            if force & 0x20:
                rec["natural"] = 0
            out.append(dict(kind=2 if inverse else 1, threads=e_lanes, workgroups=1, full_tiles=0, range_blocks=min(Te, n - at),
                            aos_offset=aos0 + at * B, **rec))
            at += Te
    return out


@pytest.fixture(scope="module")
def lib(pkg):
    l = C.CDLL(pkg._lib.lib_path())
    l.dxtlt_debug_plan_transform.restype = C.c_int32
    l.dxtlt_debug_plan_transform.argtypes = [C.c_int32] * 5 + [C.c_uint64] * 5 + [C.c_void_p, C.c_int32]
    l.dxtlt_set_tuning.argtypes = [C.c_int32, C.c_int32]
    return l


def plan(lib, fmt, inverse, variant, sa, sc, src, dst, total, first, num, cap=64):
    out = (Launch * cap)()
    n = lib.dxtlt_debug_plan_transform(fmt, int(inverse), variant, int(sa), int(sc), src, dst, total, first, num, out, cap)
    return n, list(out)[:max(0, min(n, cap))]


def check(lib, fmt, inverse, sa, sc, src, dst, total, first, num, **kw):
    want = expected(fmt, inverse, sa, sc, src, dst, total, first, num, **kw)
    n, got = plan(lib, fmt, inverse, 1, sa, sc, src, dst, total, first, num)
    tag = (fmt, inverse, sa, sc, hex(src), hex(dst), total, first, num, kw)
    assert n == len(want), (tag, n, want)
    ns = len(streams(fmt, sa, sc))
    for g, w in zip(got, want):
        assert (g.kind, g.threads, g.workgroups, g.full_tiles, g.range_blocks, g.aos_offset) == (
            w["kind"], w["threads"], w["workgroups"], w["full_tiles"], w["range_blocks"], w["aos_offset"]), (tag, w)
        if w["kind"]:
            assert list(g.shift)[:ns] == w["shift"] and g.natural == w["natural"] and g.halo_vecs == w["halo_vecs"], (tag, w)
            assert list(g.gbase)[:ns] == w["gbase"], (tag, w)


ALL = [(1, 0, 1), (1, 0, 0), (2, 0, 1), (2, 0, 0), (3, 1, 1), (3, 0, 1), (3, 1, 0), (3, 0, 0)]


@pytest.mark.parametrize("fmt,sa,sc", ALL)
@pytest.mark.parametrize("inverse", [False, True])
def test_whole_buffers_of_every_alignment_class(lib, fmt, sa, sc, inverse):
    """block counts around every tile boundary, 2^k (every stream on its line), 2^k + e (off it), mip-chain counts; pointers that are
    line-aligned, 16-byte aligned only, and off by 8 / 2 / 1 bytes (the last: shifts that are not natural)"""
    B = BLOCK[fmt]
    T = 4096 // B
    counts = [1, 2, T - 1, T, T + 1, 2 * T, 3 * T + 17, 64 * T, 64 * T + 1, 64 * T + 23, 64 * T + 63, 1 << 20, (1 << 20) + 1,
              (4 ** 9 - 1) // 3, 5463, 1398103]
    for n in counts:
        for lead in (0, 128, 16, 64, 8, 2, 1):
            src, dst = 0x7F00_0000_0000, 0x7F40_0000_0000
            if inverse:
                src += lead
            else:
                dst += lead
            check(lib, fmt, inverse, sa, sc, src, dst, n, 0, n)


@pytest.mark.parametrize("fmt,sa,sc", [(1, 0, 1), (2, 0, 1), (3, 1, 1), (3, 0, 0)])
@pytest.mark.parametrize("inverse", [False, True])
def test_ranges_of_a_larger_array(lib, fmt, sa, sc, inverse):
    """dxtlt_transform_range_device: the shifts are those of the RANGE's first block inside the whole transformed buffer"""
    rng = np.random.default_rng(17 * fmt + 3 * sa + sc + inverse)
    B = BLOCK[fmt]
    T = 4096 // B
    total = 100 * T + 37
    for _ in range(60):
        first = int(rng.integers(0, total))
        num = int(rng.integers(1, total - first + 1))
        if rng.integers(0, 3) == 0:
            first -= first % T
        check(lib, fmt, inverse, sa, sc, 0x7E00_0000_0000, 0x7E80_0000_0000, total, first, num)
    # an aligned array cut on tile boundaries keeps the aligned tiles; a cut one block further does not
    total = 128 * T
    check(lib, fmt, inverse, sa, sc, 0x7E00_0000_0000, 0x7E80_0000_0000, total, 32 * T, 64 * T)
    n, got = plan(lib, fmt, inverse, 1, sa, sc, 0x7E00_0000_0000, 0x7E80_0000_0000, total, 32 * T, 64 * T)
    assert [g.kind for g in got] == [0]
    n, got = plan(lib, fmt, inverse, 1, sa, sc, 0x7E00_0000_0000, 0x7E80_0000_0000, total, 32 * T + 1, 64 * T)
    assert [g.kind for g in got] == [2 if inverse else 1]


def test_more_than_2_to_the_31_blocks_goes_out_in_pieces(lib):
    """HIP refuses 2^32 threads per launch: a 64 GiB BC3 buffer (2^32 blocks) is two launches of 2^31 blocks, the second one
    2^35 bytes further into the block array"""
    n = 1 << 32
    check(lib, 3, False, 1, 1, 0x7000_0000_0000, 0x7800_0000_0000, n, 0, n)
    cnt, got = plan(lib, 3, False, 1, 1, 1, 0x7000_0000_0000, 0x7800_0000_0000, n, 0, n)
    assert cnt == 2 and [g.aos_offset for g in got] == [0, 1 << 35] and all(g.kind == 0 and g.workgroups == 1 << 23 for g in got)
    check(lib, 1, True, 0, 1, 0x7000_0000_0000, 0x7800_0000_0000, n + 5, 0, n + 5)     # odd count: shifted tiles in both pieces


def test_tuning_levers_change_the_plan_as_documented(lib):
    """tile_threads picks the aligned tiles' size (an edge tile is at most 256 lanes' worth: 512-lane BC1 tiles can leave two);
    force_path 2 sends aligned data to the halo / shifted tiles, 0x20 switches their natural-shift form off; bits outside
    dxtlt_tuning_mask() change nothing"""
    try:
        for threads in (64, 128, 256, 512):
            lib.dxtlt_set_tuning(threads, 0)
            for fmt, sa, sc in ALL:
                for inverse in (False, True):
                    check(lib, fmt, inverse, sa, sc, 0x7F00_0000_0000, 0x7F40_0000_0000, 9 * 512 + 300, 0, 9 * 512 + 300, tile_threads=threads)
        for force in (2, 0x20, 2 | 0x20):
            lib.dxtlt_set_tuning(0, force)
            for fmt, sa, sc in ALL:
                for inverse in (False, True):
                    check(lib, fmt, inverse, sa, sc, 0x7F00_0000_0000, 0x7F40_0000_0000, 1 << 16, 0, 1 << 16, force=force)
                    check(lib, fmt, inverse, sa, sc, 0x7F00_0000_0000, 0x7F40_0000_0000, (1 << 16) + 1, 0, (1 << 16) + 1, force=force)
        if lib.dxtlt_tuning_mask() == 0x22:
            for force in (1, 0x10, 0x400, 0x1000, 0x2000):
                lib.dxtlt_set_tuning(0, force)
                check(lib, 3, False, 1, 1, 0x7F00_0000_0000, 0x7F40_0000_0000, 4097, 0, 4097)
    finally:
        lib.dxtlt_set_tuning(0, 0)


def test_refused_arguments_and_capacity(lib):
    out = (Launch * 4)()
    f = lib.dxtlt_debug_plan_transform
    assert f(0, 0, 1, 0, 1, 1 << 40, 1 << 41, 10, 0, 10, out, 4) == -1          # format
    assert f(1, 0, 9, 0, 1, 1 << 40, 1 << 41, 10, 0, 10, out, 4) == -1          # variant
    assert f(1, 0, 1, 0, 1, 1 << 40, 1 << 41, 10, 8, 4, out, 4) == -1           # range past the array
    assert f(1, 0, 1, 0, 1, 1 << 40, 1 << 41, 10, 0, 0, out, 4) == 0            # nothing to do
    assert f(1, 0, 1, 0, 1, 1 << 40, 1 << 41, 1 << 20, 0, 1 << 20, None, 0) == 1   # counting only
