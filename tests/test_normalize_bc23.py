"""BC2 / BC3 block normalisation -- the reference's experimental modules
(/root/reference/src/core/dxt-lossless-transform-bc{2,3}/src/experimental/normalize_blocks/normalize.rs).

CPU: the C oracle replays the reference's unit-test vectors (so this row is PINNED), the device header built for the
host agrees with the oracle's pixel-by-pixel statement on millions of structured blocks (all alpha endpoint pairs, all
colour endpoints), and the "distinct table entries" claims behind the kernels' early exits are checked exhaustively.
GPU (-m gpu): every entry point through the C ABI against the oracle."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RED = [0x00, 0xF8]
A_NONE, A_UNIFORM, A_FILL, A_ZEROMAX = 0, 1, 2, 3
C_NONE, C_COLOR0, C_REPL = 0, 1, 2


def u8(*v):
    return np.array(v, dtype=np.uint8)


def colour_half(mode, c565=RED, keep=(0x12, 0x34, 0, 0, 0, 0)):
    if mode == C_NONE:
        return [*c565, *keep]
    return [*c565, *(c565 if mode == C_REPL else [0, 0]), 0, 0, 0, 0]


# ---------------------------------------------------------------------------------------------------------------
# reference unit tests, BC2 (bc2 normalize.rs tests)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", [C_COLOR0, C_REPL])
def test_ref_bc2_solid_mixed_nonroundtrip_varying_alpha(oracle, mode):
    solid = u8(*([0xFF] * 8), *RED, 0x01, 0x01, 0, 0, 0, 0)
    want = u8(*([0xFF] * 8), *RED, *(RED if mode == C_REPL else [0, 0]), 0, 0, 0, 0)
    assert np.array_equal(oracle.normalize_bc2_blocks(solid, mode), want)
    mixed = u8(*([0xFF] * 8), *RED, 0x1F, 0x00, 0x11, 0x11, 0x11, 0x11)
    assert np.array_equal(oracle.normalize_bc2_blocks(mixed, mode), mixed)
    nonrt = u8(*([0xFF] * 8), *RED, 0x1F, 0x00, 0xAA, 0xAA, 0xAA, 0xAA)
    assert np.array_equal(oracle.normalize_bc2_blocks(nonrt, mode), nonrt)
    varying = u8(*[(x * 32) & 0xFF for x in range(8)], *RED, 0, 0, 0, 0, 0, 0)   # alpha is kept, colour normalised
    want = varying.copy()
    want[10:12] = RED if mode == C_REPL else [0, 0]
    assert np.array_equal(oracle.normalize_bc2_blocks(varying, mode), want)
    two = np.concatenate([u8(*([0xFF] * 8), *RED, 0, 0, 0, 0, 0, 0), mixed])
    want = np.concatenate([u8(*([0xFF] * 8), *RED, *(RED if mode == C_REPL else [0, 0]), 0, 0, 0, 0), mixed])
    assert np.array_equal(oracle.normalize_bc2_blocks(two, mode), want)
    # explicit alpha decodes as nibble * 17 and does not take part in the decision
    px = oracle.decode_block("bc2", varying)
    assert px[0, 3] == 0 and px[3, 3] == 2 * 17 and (px[:, 0] == 255).all()


def test_ref_bc2_all_modes_and_split(oracle):
    blk = u8(*([0xFF] * 8), *RED, 0x01, 0x01, 0, 0, 0, 0)
    x = np.concatenate([blk, blk])
    outs = oracle.normalize_bc2_blocks_all_modes(x)
    for m in range(3):
        assert np.array_equal(outs[m], oracle.normalize_bc2_blocks(x, m))
    # split in place: three blocks, two normalised (bc2 normalize.rs can_normalize_split_blocks_in_place)
    alpha = np.full(24, 0xFF, np.uint8)
    colors = u8(*([0x00, 0xF8, 0x00, 0xF8] * 3))
    indices = np.full(12, 0xAA, np.uint8)
    c, i = oracle.normalize_bc2_split_blocks(alpha[:16], colors[:8], indices[:8], C_COLOR0)
    assert np.array_equal(c, u8(0x00, 0xF8, 0, 0, 0x00, 0xF8, 0, 0)) and (i == 0).all()


# ---------------------------------------------------------------------------------------------------------------
# reference unit tests, BC3 (bc3 normalize.rs tests)
# ---------------------------------------------------------------------------------------------------------------
def alpha_half(mode, alpha=0xFF, original=(0xFF, 0xFF, 0, 0, 0, 0, 0, 0)):
    if mode == A_NONE:
        return list(original)
    if alpha == 255 and mode == A_FILL:
        return [0xFF] * 8
    if alpha == 255 and mode == A_ZEROMAX:
        return [0, 0] + [0xFF] * 6
    return [alpha, 0, 0, 0, 0, 0, 0, 0]


@pytest.mark.parametrize("amode", [A_NONE, A_UNIFORM, A_FILL, A_ZEROMAX])
@pytest.mark.parametrize("cmode", [C_NONE, C_COLOR0, C_REPL])
def test_ref_bc3_opaque_alpha_single_colour(oracle, amode, cmode):
    block = u8(0xFF, 0xFF, 0, 0, 0, 0, 0, 0, *RED, 0x12, 0x34, 0, 0, 0, 0)
    want = u8(*alpha_half(amode), *colour_half(cmode))
    got = oracle.normalize_bc3_blocks(block, amode, cmode)
    assert np.array_equal(got, want)
    px = oracle.decode_block("bc3", got)   # still red, still opaque
    assert (px == np.array([255, 0, 0, 255], dtype=np.uint8)).all()


def test_ref_bc3_other_cases(oracle):
    mixed_colours = u8(0xFF, 0xFF, 0, 0, 0, 0, 0, 0, 0x00, 0xF8, 0xE0, 0x07, 0x11, 0x11, 0x11, 0x11)
    for amode in (A_UNIFORM, A_FILL):
        got = oracle.normalize_bc3_blocks(mixed_colours, amode, C_NONE)
        assert np.array_equal(got, u8(*alpha_half(amode), *mixed_colours[8:]))
    mixed_alpha = u8(0xFF, 0x80, *([0x55] * 6), *RED, 0x12, 0x34, 0, 0, 0, 0)
    for cmode in (C_COLOR0, C_REPL):
        got = oracle.normalize_bc3_blocks(mixed_alpha, A_UNIFORM, cmode)
        assert np.array_equal(got, u8(*mixed_alpha[:8], *colour_half(cmode)))
    counting = u8(0xFF, 0xFF, 0, 0, 0, 0, 0, 0, *range(8, 16))
    got = oracle.normalize_bc3_blocks(counting, A_ZEROMAX, C_NONE)
    assert np.array_equal(got, u8(0, 0, *([0xFF] * 6), *range(8, 16)))
    assert (oracle.decode_block("bc3", got)[:, 3] == 255).all()
    half = u8(128, 128, 0, 0, 0, 0, 0, 0, *range(8, 16))
    for amode in (A_UNIFORM, A_FILL, A_ZEROMAX):
        assert np.array_equal(oracle.normalize_bc3_blocks(half, amode, C_NONE), u8(128, 0, 0, 0, 0, 0, 0, 0, *range(8, 16)))
    assert np.array_equal(oracle.normalize_bc3_blocks(half, A_NONE, C_NONE), half)
    assert (oracle.decode_block("bc3", half)[:, 3] == 128).all()


def test_ref_bc3_all_modes_split_and_in_place(oracle):
    block = u8(0xFF, 0xFF, 0, 0, 0, 0, 0, 0, *RED, 0x12, 0x34, 0, 0, 0, 0)
    x = np.concatenate([block, u8(0xFF, 0x80, *([0x55] * 6), *RED, 0x12, 0x34, 0, 0, 0, 0)])
    outs = oracle.normalize_bc3_blocks_all_modes(x)
    for a in range(4):
        for c in range(3):
            want = oracle.normalize_bc3_blocks(x, a, c)
            assert np.array_equal(outs[a * 3 + c], want)
            b = x.reshape(-1, 16)
            parts = oracle.normalize_bc3_split_blocks(b[:, :2].reshape(-1), b[:, 2:8].reshape(-1), b[:, 8:12].reshape(-1),
                                                      b[:, 12:].reshape(-1), a, c)
            w = want.reshape(-1, 16)
            for got, sl in zip(parts, (slice(0, 2), slice(2, 8), slice(8, 12), slice(12, 16))):
                assert np.array_equal(got.reshape(len(b), -1), w[:, sl])
    l = oracle.lib()
    y = x.copy()
    l.oracle_normalize_bc3_blocks(y.ctypes.data, y.ctypes.data, y.size, A_FILL, C_REPL)
    assert np.array_equal(y, oracle.normalize_bc3_blocks(x, A_FILL, C_REPL))


# ---------------------------------------------------------------------------------------------------------------
# the claims behind the kernels' early exits, and the device header against the oracle
# ---------------------------------------------------------------------------------------------------------------
def test_alpha_table_entries_are_distinct_when_endpoints_are_apart():
    """csrc/bc23_normalize.h: a0 - a1 >= 7 (eight-value mode), or a1 - a0 >= 5 with a0 != 0 and a1 != 255 (six-value
    mode), makes the eight decoded alpha values pairwise different.  Every endpoint pair."""
    a0, a1 = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    eight = a0 > a1
    tab = np.zeros((8, 256, 256), dtype=np.int64)
    tab[0], tab[1] = a0, a1
    for k in range(2, 8):
        tab[k] = np.where(eight, ((8 - k) * a0 + (k - 1) * a1) // 7,
                          ((6 - k) * a0 + (k - 1) * a1) // 5 if k < 6 else (0 if k == 6 else 255))
    distinct = np.ones((256, 256), dtype=bool)
    for i in range(8):
        for j in range(i + 1, 8):
            distinct &= tab[i] != tab[j]
    claimed = np.where(eight, a0 - a1 >= 7, (a1 - a0 >= 5) & (a0 != 0) & (a1 != 255))
    assert distinct[claimed].all()


@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("shim") / "normalize_header_shim.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-shared", "-fPIC", "-Wall", "-Wextra", "-o", so,
                           os.path.join(ROOT, "tests", "cpp", "normalize_header_shim.cpp")])
    lib = ctypes.CDLL(so)
    lib.shim_normalize_bc23_blocks.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int,
                                               ctypes.c_int]
    lib.shim_normalize_bc23_blocks.restype = None
    return lib


def structured_blocks(fmt, rng):
    """Every alpha endpoint pair with several index patterns (BC3) / random alpha (BC2), crossed with colour halves
    that hit every colour case."""
    colour_cases = []
    c0 = rng.integers(0, 65536, 4096, dtype=np.uint32)
    for c1 in (c0, c0 ^ 1, c0 ^ 0x20, c0 ^ 0x800, rng.integers(0, 65536, 4096, dtype=np.uint32)):
        for pat in (0x00000000, 0x55555555, 0xAAAAAAAA, 0xFFFFFFFF, 0x00AA5500, 0x12345678):
            h = np.empty((4096, 8), dtype=np.uint8)
            h[:, 0], h[:, 1], h[:, 2], h[:, 3] = c0 & 255, c0 >> 8, c1 & 255, c1 >> 8
            h[:, 4:] = np.frombuffer(np.uint32(pat).tobytes(), dtype=np.uint8)
            colour_cases.append(h)
    col = np.concatenate(colour_cases)                       # 122 880 colour halves
    n = col.shape[0]
    blocks = np.empty((n, 16), dtype=np.uint8)
    blocks[:, 8:] = col
    if fmt == "bc2":
        blocks[:, :8] = rng.integers(0, 256, (n, 8), dtype=np.uint8)
        return blocks.reshape(-1)
    a0 = (np.arange(n) % 256).astype(np.uint8)
    a1 = ((np.arange(n) // 256 + np.arange(n) * 7) % 256).astype(np.uint8)
    near = rng.integers(0, 2, n).astype(bool)                # half the blocks: neighbouring endpoints (the slow path)
    a1 = np.where(near, (a0.astype(np.int64) + rng.integers(-8, 9, n)) % 256, a1).astype(np.uint8)
    blocks[:, 0], blocks[:, 1] = a0, a1
    idx_kind = rng.integers(0, 4, n)
    single = (rng.integers(0, 8, n).astype(np.uint64) * np.uint64(0x249249249249))
    two_vals = single ^ (rng.integers(0, 2, n).astype(np.uint64) * np.uint64(0x1 << 21))
    rand = rng.integers(0, 1 << 48, n, dtype=np.uint64)
    sixes = np.uint64(0x249249249249) * np.uint64(6) ^ (rng.integers(0, 2, n).astype(np.uint64) * np.uint64(0x1 << 3))
    bits = np.select([idx_kind == 0, idx_kind == 1, idx_kind == 2], [single, two_vals, rand], default=sixes)
    for k in range(6):
        blocks[:, 2 + k] = ((bits >> np.uint64(8 * k)) & np.uint64(0xFF)).astype(np.uint8)
    return blocks.reshape(-1)


@pytest.mark.parametrize("fmt", ["bc2", "bc3"])
def test_device_header_equals_oracle(oracle, shim, fmt):
    rng = np.random.default_rng(0xBC23)
    x = np.ascontiguousarray(structured_blocks(fmt, rng))
    # exhaustive alpha endpoints with a neighbouring-index pattern (BC3): all 65 536 (a0, a1) pairs
    if fmt == "bc3":
        e = np.zeros((65536, 16), dtype=np.uint8)
        e[:, 0], e[:, 1] = np.arange(65536) & 255, np.arange(65536) >> 8
        e[:, 2:8] = np.frombuffer(np.uint64(0x249249249249 * 2 ^ 0b011).tobytes()[:6], dtype=np.uint8)   # values 2 and 1
        e[:, 8:] = x[8:16]
        x = np.concatenate([x, e.reshape(-1)])
    n = x.size // 16
    combos = [(0, c) for c in range(3)] if fmt == "bc2" else [(a, c) for a in range(4) for c in range(3)]
    for a, c in combos:
        got = np.empty_like(x)
        shim.shim_normalize_bc23_blocks(2 if fmt == "bc2" else 3, x.ctypes.data, got.ctypes.data, n, a, c)
        want = oracle.normalize_bc2_blocks(x, c) if fmt == "bc2" else oracle.normalize_bc3_blocks(x, a, c)
        bad = np.flatnonzero((got.reshape(-1, 16) != want.reshape(-1, 16)).any(axis=1))
        assert bad.size == 0, (fmt, a, c, x.reshape(-1, 16)[bad[:3]], got.reshape(-1, 16)[bad[:3]], want.reshape(-1, 16)[bad[:3]])
    # the structured set really exercises the cases
    if fmt == "bc3":
        w = oracle.normalize_bc3_blocks(x, A_UNIFORM, C_COLOR0).reshape(-1, 16)
        xb = x.reshape(-1, 16)
        assert ((w[:, :8] != xb[:, :8]).any(axis=1)).sum() > 20_000 and ((w[:, 8:] != xb[:, 8:]).any(axis=1)).sum() > 20_000


# ---------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------
torch = pytest.importorskip("torch")
SIZES = (1, 2, 255, 256, 257, 4097, 70_001)


@pytest.fixture(scope="module")
def n23(pkg):
    from dxt_lossless_transform_amd import normalize23 as mod

    return mod


def test_blocks(fmt, n, seed):
    rng = np.random.default_rng(seed)
    x = structured_blocks(fmt, rng)
    take = rng.integers(0, x.size // 16, n)
    return np.ascontiguousarray(x.reshape(-1, 16)[take].reshape(-1))


test_blocks.__test__ = False


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["bc2", "bc3"])
def test_gpu_normalize_blocks_bc23(pkg, n23, oracle, fmt):
    dev = torch.device("cuda:0")
    combos = [(0, c) for c in range(3)] if fmt == "bc2" else [(a, c) for a in range(4) for c in range(3)]
    for n in SIZES:
        x = test_blocks(fmt, n, n)
        xd = torch.from_numpy(x).to(dev)
        for a, c in combos:
            want = oracle.normalize_bc2_blocks(x, c) if fmt == "bc2" else oracle.normalize_bc3_blocks(x, a, c)
            yd = torch.full((x.size + 32,), 0x5A, dtype=torch.uint8, device=dev)
            n23.normalize_blocks(fmt, xd, yd[: x.size], n23.ColorNormalizationMode(c), n23.AlphaNormalizationMode(a))
            got = yd.cpu().numpy()
            assert np.array_equal(got[: x.size], want) and (got[x.size:] == 0x5A).all(), (fmt, n, a, c)
            zd = xd.clone()
            n23.normalize_blocks(fmt, zd, zd, n23.ColorNormalizationMode(c), n23.AlphaNormalizationMode(a))
            assert np.array_equal(zd.cpu().numpy(), want), (fmt, n, a, c, "in place")
        if n > 2:   # misaligned views -> byte path
            big = torch.zeros(x.size + 32, dtype=torch.uint8, device=dev)
            src, dst = big[5: 5 + x.size], torch.zeros(x.size + 32, dtype=torch.uint8, device=dev)[2: 2 + x.size]
            src.copy_(xd)
            a, c = combos[-1]
            n23.normalize_blocks(fmt, src, dst, n23.ColorNormalizationMode(c), n23.AlphaNormalizationMode(a))
            want = oracle.normalize_bc2_blocks(x, c) if fmt == "bc2" else oracle.normalize_bc3_blocks(x, a, c)
            assert np.array_equal(dst.cpu().numpy(), want)
    # host pointers
    x = test_blocks(fmt, 30_001, 77)
    a, c = combos[-1]
    y = np.zeros_like(x)
    n23.normalize_blocks(fmt, x, y, n23.ColorNormalizationMode(c), n23.AlphaNormalizationMode(a))
    assert np.array_equal(y, oracle.normalize_bc2_blocks(x, c) if fmt == "bc2" else oracle.normalize_bc3_blocks(x, a, c))
    with pytest.raises(pkg.InvalidLength):
        n23.normalize_blocks(fmt, np.zeros(24, np.uint8), np.zeros(24, np.uint8), n23.ColorNormalizationMode(1))


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["bc2", "bc3"])
def test_gpu_all_modes_bc23(n23, oracle, fmt):
    dev = torch.device("cuda:0")
    count = 3 if fmt == "bc2" else 12
    for n in (1, 257, 20_003):
        x = test_blocks(fmt, n, n + 5)
        want = oracle.normalize_bc2_blocks_all_modes(x) if fmt == "bc2" else oracle.normalize_bc3_blocks_all_modes(x)
        xd = torch.from_numpy(x).to(dev)
        outs = [torch.empty_like(xd) for _ in range(count)]
        n23.normalize_blocks_all_modes(fmt, xd, outs)
        for k in range(count):
            assert np.array_equal(outs[k].cpu().numpy(), want[k]), (fmt, n, k)
        houts = [np.zeros_like(x) for _ in range(count)]
        n23.normalize_blocks_all_modes(fmt, x, houts)
        for k in range(count):
            assert np.array_equal(houts[k], want[k]), (fmt, n, k, "host")


@pytest.mark.gpu
def test_gpu_split_in_place_bc23(n23, oracle):
    dev = torch.device("cuda:0")
    for n in (1, 3, 256, 4099):
        b2 = test_blocks("bc2", n, n + 9).reshape(-1, 16)
        al, c, i = b2[:, :8].reshape(-1).copy(), b2[:, 8:12].reshape(-1).copy(), b2[:, 12:].reshape(-1).copy()
        for mode in range(3):
            wc, wi = oracle.normalize_bc2_split_blocks(al, c, i, mode)
            cd, idd = torch.from_numpy(c).to(dev), torch.from_numpy(i).to(dev)
            n23.normalize_bc2_split_blocks_in_place(None, cd, idd, n23.ColorNormalizationMode(mode))
            assert np.array_equal(cd.cpu().numpy(), wc) and np.array_equal(idd.cpu().numpy(), wi)
            hc, hi = c.copy(), i.copy()
            n23.normalize_bc2_split_blocks_in_place(al, hc, hi, n23.ColorNormalizationMode(mode))
            assert np.array_equal(hc, wc) and np.array_equal(hi, wi)
        b3 = test_blocks("bc3", n, n + 11).reshape(-1, 16)
        parts = [b3[:, :2].reshape(-1).copy(), b3[:, 2:8].reshape(-1).copy(), b3[:, 8:12].reshape(-1).copy(),
                 b3[:, 12:].reshape(-1).copy()]
        for a in range(4):
            for cm in range(3):
                want = oracle.normalize_bc3_split_blocks(*parts, a, cm)
                dparts = [torch.from_numpy(p.copy()).to(dev) for p in parts]
                n23.normalize_bc3_split_blocks_in_place(*dparts, n23.AlphaNormalizationMode(a), n23.ColorNormalizationMode(cm))
                for g, w in zip(dparts, want):
                    assert np.array_equal(g.cpu().numpy(), w), (n, a, cm)
                hparts = [p.copy() for p in parts]
                n23.normalize_bc3_split_blocks_in_place(*hparts, n23.AlphaNormalizationMode(a), n23.ColorNormalizationMode(cm))
                for g, w in zip(hparts, want):
                    assert np.array_equal(g, w), (n, a, cm, "host")


@pytest.mark.gpu
def test_gpu_normalize_bc3_rate(pkg, n23):
    """1 GiB of BC3 blocks: prints the stand-alone kernel's rate on random and on mostly-normalisable data."""
    dev = torch.device("cuda:0")
    x = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    pkg.fill_splitmix64(x, 0x0BC3_4E01)
    y = torch.empty_like(x)
    for label in ("random", "uniform alpha + solid colour"):
        if label != "random":
            b = x.view(-1, 16)
            b[:, 2:8] = 0
            b[:, 12:] = 0
        for _ in range(2):
            n23.normalize_blocks("bc3", x, y, n23.ColorNormalizationMode.COLOR0_ONLY, n23.AlphaNormalizationMode.OPAQUE_FILL_ALL)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(10):
            n23.normalize_blocks("bc3", x, y, n23.ColorNormalizationMode.COLOR0_ONLY, n23.AlphaNormalizationMode.OPAQUE_FILL_ALL)
        ev[1].record()
        torch.cuda.synchronize()
        ms = ev[0].elapsed_time(ev[1]) / 10
        print(f"BC3 normalize_blocks, 1 GiB, {label}: {ms:.3f} ms, {2 * x.numel() / ms / 1e-3 / 8e12:.3f} of 8 TB/s")
