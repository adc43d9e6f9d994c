"""BC1 block normalisation -- the reference's experimental module
(/root/reference/src/core/dxt-lossless-transform-bc1/src/experimental/normalize_blocks/).

CPU part: the C oracle replays every unit test of normalize.rs:505-1076 (these hold concrete input and expected blocks,
so this row of the oracle is PINNED by reference vectors), agrees with the independent numpy statement, and the
composition transform_bc1_with_normalize_blocks == normalise then transform.
GPU part (-m gpu): stand-alone kernels, the fused normalise+transform, and transform_bc1_auto_with_normalization, all
through the C ABI against the oracle."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle_auto
from oracle import oracle_np as onp

NONE, COLOR0, REPL = 0, 1, 2
RED = [0x00, 0xF8]


def u8(*v):
    return np.array(v, dtype=np.uint8)


def crafted_blocks(oracle, n, seed=1):
    """Random BC1 blocks with every normalisation case mixed in by block index modulo 8."""
    x = oracle.fill_splitmix64(n * 8, 0x0BC1_4E00 + seed)
    b = x.reshape(-1, 8)
    k = np.arange(n) % 8
    c0 = b[:, 0].astype(np.uint32) | (b[:, 1].astype(np.uint32) << 8)
    c1 = b[:, 2].astype(np.uint32) | (b[:, 3].astype(np.uint32) << 8)
    lo, hi = np.minimum(c0, c1), np.maximum(c0, c1)

    def set_colours(rows, a, bb):
        b[rows, 0], b[rows, 1] = (a[rows] & 255).astype(np.uint8), (a[rows] >> 8).astype(np.uint8)
        b[rows, 2], b[rows, 3] = (bb[rows] & 255).astype(np.uint8), (bb[rows] >> 8).astype(np.uint8)

    b[k == 1, 4:] = 0x00                      # all pixels = colour 0: solid, always round-trippable
    b[k == 2, 4:] = 0x55                      # all pixels = colour 1
    b[k == 3, 4:] = 0xAA                      # all pixels = first interpolated colour: solid, rarely round-trippable
    r4 = k == 4                               # c0 == c1 (three-colour mode), indices in {0, 1, 2}: solid
    set_colours(r4, c0, c0)
    b[r4, 4:] = b[r4, 4:] & 0x55 | ((b[r4, 4:] >> 1) & 0x55 & ~(b[r4, 4:] & 0x55)) << 1
    r5 = k == 5                               # c0 <= c1 and all indices 3: fully transparent
    set_colours(r5, lo, hi)
    b[r5, 4:] = 0xFF
    r6 = k == 6                               # c0 > c1 and all indices 3: solid second interpolated colour
    set_colours(r6, np.maximum(hi, 1), np.minimum(lo, np.maximum(hi, 1) - 1))
    b[r6, 4:] = 0xFF
    r7 = k == 7                               # c0 <= c1, transparent and opaque pixels mixed: kept
    set_colours(r7, lo, hi)
    b[r7, 4] = 0xF0
    return x


# ---------------------------------------------------------------------------------------------------------------
# CPU: the reference's own unit tests, replayed against the oracle
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", [COLOR0, REPL])
def test_ref_can_normalize_solid_color_block(oracle, mode):  # normalize.rs:508-560
    block = u8(*RED, 0x01, 0x01, 0, 0, 0, 0)
    want = u8(*RED, *(RED if mode == REPL else [0, 0]), 0, 0, 0, 0)
    assert np.array_equal(oracle.normalize_bc1_blocks(block, mode), want)


@pytest.mark.parametrize("mode", [COLOR0, REPL])
def test_ref_can_normalize_transparent_block(oracle, mode):  # normalize.rs:562-598
    block = u8(0x00, 0x80, 0x00, 0xF8, 0xFF, 0xFF, 0xFF, 0xFF)
    assert np.array_equal(oracle.normalize_bc1_blocks(block, mode), np.full(8, 0xFF, np.uint8))


@pytest.mark.parametrize("mode", [COLOR0, REPL])
def test_ref_can_preserve_mixed_color_block(oracle, mode):  # normalize.rs:600-635
    block = u8(*RED, 0x1F, 0x00, 0x11, 0x11, 0x11, 0x11)
    assert np.array_equal(oracle.normalize_bc1_blocks(block, mode), block)


@pytest.mark.parametrize("mode", [COLOR0, REPL])
def test_ref_can_preserve_non_roundtrippable_color_block(oracle, mode):  # normalize.rs:637-674
    block = u8(*RED, 0x1F, 0x00, 0xAA, 0xAA, 0xAA, 0xAA)
    # the reference's comment: decoded 8888 = (170, 0, 85, 255), round-tripped = (173, 0, 82)
    px = oracle.decode_bc1_block(block)
    assert (px == np.array([170, 0, 85, 255], dtype=np.uint8)).all()
    assert np.array_equal(oracle.normalize_bc1_blocks(block, mode), block)


@pytest.mark.parametrize("mode", [COLOR0, REPL])
def test_ref_can_normalize_multiple_blocks(oracle, mode):  # normalize.rs:676-747
    src = u8(*RED, 0, 0, 0, 0, 0, 0, 0x00, 0x80, 0x00, 0xF8, 0xFF, 0xFF, 0xFF, 0xFF)
    want = u8(*RED, *(RED if mode == REPL else [0, 0]), 0, 0, 0, 0, *([0xFF] * 8))
    assert np.array_equal(oracle.normalize_bc1_blocks(src, mode), want)


def test_ref_can_normalize_blocks_all_modes(oracle):  # normalize.rs:750-848
    src = u8(*RED, 0, 0, 0, 0, 0, 0, 0x00, 0x80, 0x00, 0xF8, 0xFF, 0xFF, 0xFF, 0xFF)
    outs, any_n = oracle.normalize_bc1_blocks_all_modes(src)
    assert any_n
    assert np.array_equal(outs[NONE][:8], src[:8])
    assert np.array_equal(outs[COLOR0][:8], u8(*RED, 0, 0, 0, 0, 0, 0))
    assert np.array_equal(outs[REPL][:8], u8(*RED, *RED, 0, 0, 0, 0))
    for o in outs:
        assert (o[8:] == 0xFF).all()


def test_ref_can_normalize_in_place(oracle):  # normalize.rs:851-911
    l = oracle.lib()
    d = u8(0x00, 0xF8, 0x00, 0xF8, 0, 0, 0, 0)
    l.oracle_normalize_bc1_blocks(d.ctypes.data, d.ctypes.data, 8, COLOR0)
    assert np.array_equal(d, u8(0x00, 0xF8, 0, 0, 0, 0, 0, 0))
    t = u8(0x00, 0x00, 0x01, 0x00, 0xFF, 0xFF, 0xFF, 0xFF)
    l.oracle_normalize_bc1_blocks(t.ctypes.data, t.ctypes.data, 8, COLOR0)
    assert (t == 0xFF).all()


@pytest.mark.parametrize("mode,fill", [(COLOR0, 0xAA), (REPL, 0x55)])
def test_ref_can_normalize_split_blocks_in_place(oracle, mode, fill):  # normalize.rs:914-1076
    colors = u8(*([0x00, 0xF8, 0x00, 0xF8] * 3))
    indices = np.full(12, fill, np.uint8)
    l = oracle.lib()
    l.oracle_normalize_bc1_split_blocks_in_place(colors.ctypes.data, indices.ctypes.data, 2, mode)
    second = [0x00, 0xF8] if mode == REPL else [0, 0]
    assert np.array_equal(colors, u8(0x00, 0xF8, *second, 0x00, 0xF8, *second, 0x00, 0xF8, 0x00, 0xF8))
    assert np.array_equal(indices, u8(*([0] * 8), *([fill] * 4)))


# ---------------------------------------------------------------------------------------------------------------
# CPU: the two statements agree; composition; classes all occur in the crafted data
# ---------------------------------------------------------------------------------------------------------------
def test_c_and_numpy_statements_agree(oracle):
    for n in (0, 1, 7, 8, 64, 1000, 20_000):
        x = crafted_blocks(oracle, n, n)
        for mode in (NONE, COLOR0, REPL):
            assert np.array_equal(oracle.normalize_bc1_blocks(x, mode), onp.normalize_bc1_blocks(x, mode)), (n, mode)
    # every 565 colour as a solid block (indices 0) normalises; exhaustive over c0 with a fixed c1
    c0 = np.arange(65536, dtype=np.uint32)
    blocks = np.zeros((65536, 8), dtype=np.uint8)
    blocks[:, 0], blocks[:, 1] = c0 & 255, c0 >> 8
    blocks[:, 2], blocks[:, 3] = 0x34, 0x12
    blocks[:, 4:] = 0
    flat = blocks.reshape(-1)
    got = oracle.normalize_bc1_blocks(flat, COLOR0).reshape(-1, 8)
    assert np.array_equal(got[:, :2], blocks[:, :2]) and (got[:, 2:] == 0).all()
    assert np.array_equal(got.reshape(-1), onp.normalize_bc1_blocks(flat, COLOR0))


def test_palette_entries_are_distinct_when_endpoints_differ():
    """The kernels' classification (csrc/bc1_normalize.h) rests on: c0 != c1 => the four palette entries are pairwise
    different.  Entries are equal only if every channel is equal, so it is enough that in any channel with different
    endpoint values the derived values are pairwise different -- checked here for every pair of 5- and 6-bit values."""
    for bits in (5, 6):
        v = np.arange(1 << bits, dtype=np.int64)
        e = (v << (8 - bits)) | (v >> (2 * bits - 8))
        a, b = np.meshgrid(e, e, indexing="ij")
        diff = a != b
        third1, third2, mid = (2 * a + b) // 3, (a + 2 * b) // 3, (a + b) // 2
        for p, q in ((a, third1), (a, third2), (b, third1), (b, third2), (third1, third2), (a, mid), (b, mid)):
            assert (p != q)[diff].all(), bits


def test_device_header_classification_equals_oracle(oracle, tmp_path):
    """csrc/bc1_normalize.h compiled for the host (tests/cpp/normalize_header_shim.cpp) against the oracle's
    pixel-by-pixel statement: every c0 with structured and random c1, single-value and mixed indices."""
    import ctypes
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "normalize_header_shim.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-shared", "-fPIC", "-Wall", "-Wextra", "-o", so,
                           os.path.join(root, "tests", "cpp", "normalize_header_shim.cpp")])
    shim = ctypes.CDLL(so)
    shim.shim_normalize_blocks.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    shim.shim_normalize_blocks.restype = None

    rng = np.random.default_rng(0xBC1)
    c0 = np.arange(65536, dtype=np.uint32)
    partners = [c0, c0 ^ 1, c0 ^ 0x20, c0 ^ 0x800, (c0 + 1) & 0xFFFF, (c0 - 1) & 0xFFFF, c0 ^ 0xFFFF,
                np.zeros_like(c0), np.full_like(c0, 0xFFFF)] + [rng.integers(0, 65536, 65536, dtype=np.uint32) for _ in range(7)]
    idx_patterns = [0x00000000, 0x55555555, 0xAAAAAAAA, 0xFFFFFFFF, 0x55550000, 0xAAAA5555, 0x00AA5500, 0xFFFFFF00]
    blocks = []
    for c1 in partners:
        for pat in idx_patterns:
            b = np.empty((65536, 8), dtype=np.uint8)
            b[:, 0], b[:, 1], b[:, 2], b[:, 3] = c0 & 255, c0 >> 8, c1 & 255, c1 >> 8
            b[:, 4:] = np.frombuffer(np.uint32(pat).tobytes(), dtype=np.uint8)
            blocks.append(b)
    r = rng.integers(0, 256, (1 << 20, 8), dtype=np.uint8)      # random blocks with few distinct index values
    r[:, 4:] &= rng.choice(np.array([0x00, 0x55, 0xAA, 0xFF, 0x0F], dtype=np.uint8), (1 << 20, 1))
    blocks.append(r)
    x = np.ascontiguousarray(np.concatenate(blocks).reshape(-1))
    for mode in (COLOR0, REPL, 3):
        got = np.empty_like(x)
        shim.shim_normalize_blocks(x.ctypes.data, got.ctypes.data, x.size // 8, mode)
        if mode == 3:   # internal "transparent blocks only" mode == the None output of normalize_blocks_all_modes
            want = oracle.normalize_bc1_blocks_all_modes(x)[0][NONE]
        else:
            want = oracle.normalize_bc1_blocks(x, mode)
        bad = np.flatnonzero((got.reshape(-1, 8) != want.reshape(-1, 8)).any(axis=1))
        assert bad.size == 0, (mode, x.reshape(-1, 8)[bad[:4]], got.reshape(-1, 8)[bad[:4]], want.reshape(-1, 8)[bad[:4]])


def test_crafted_data_hits_every_case(oracle):
    x = crafted_blocks(oracle, 8000)
    y = oracle.normalize_bc1_blocks(x, COLOR0).reshape(-1, 8)
    xb = x.reshape(-1, 8)
    transparent = (y == 0xFF).all(axis=1) & ~(xb == 0xFF).all(axis=1)
    solid = (y[:, 2:] == 0).all(axis=1) & (xb != y).any(axis=1)
    kept = (xb == y).all(axis=1)
    assert transparent.sum() > 500 and solid.sum() > 2000 and kept.sum() > 1500
    # idempotent, and decoded pixels are unchanged by normalisation (it is visually lossless)
    assert np.array_equal(oracle.normalize_bc1_blocks(y.reshape(-1), COLOR0), y.reshape(-1))
    for mode in (COLOR0, REPL):
        z = oracle.normalize_bc1_blocks(x, mode)
        assert np.array_equal(onp.decode_bc1_pixels(z), onp.decode_bc1_pixels(x))


def test_split_and_all_modes_equal_the_block_form(oracle):
    x = crafted_blocks(oracle, 4003, 3)
    b = x.reshape(-1, 8)
    for mode in (NONE, COLOR0, REPL):
        want = oracle.normalize_bc1_blocks(x, mode).reshape(-1, 8)
        c, i = oracle.normalize_bc1_split_blocks(b[:, :4].reshape(-1).copy(), b[:, 4:].reshape(-1).copy(), mode)
        assert np.array_equal(c.reshape(-1, 4), want[:, :4]) and np.array_equal(i.reshape(-1, 4), want[:, 4:])
    outs, any_n = oracle.normalize_bc1_blocks_all_modes(x)
    assert any_n
    for mode in (COLOR0, REPL):
        assert np.array_equal(outs[mode], oracle.normalize_bc1_blocks(x, mode))
    # the `None` output is NOT a copy: fully transparent blocks are rewritten in every output (normalize.rs:447-454)
    none = x.reshape(-1, 8).copy()
    none[(outs[COLOR0].reshape(-1, 8) == 0xFF).all(axis=1)] = 0xFF
    assert np.array_equal(outs[NONE], none.reshape(-1)) and not np.array_equal(outs[NONE], x)
    plain = oracle.fill_splitmix64(8 * 512, 99)   # random blocks: nothing to normalise
    assert oracle.normalize_bc1_blocks_all_modes(plain)[1] is False


def test_transform_with_normalize_is_the_composition(oracle):
    x = crafted_blocks(oracle, 1031, 5)
    for mode in (NONE, COLOR0, REPL):
        n = oracle.normalize_bc1_blocks(x, mode)
        for variant in range(4):
            for split in (False, True):
                got = oracle.transform_bc1_with_normalize_blocks(x, mode, variant, split)
                assert np.array_equal(got, oracle.transform("bc1", n, variant, split))
                assert np.array_equal(oracle.transform("bc1", got, variant, split, inverse=True), n)


# ---------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------
torch = pytest.importorskip("torch")
SIZES = (1, 2, 3, 15, 255, 256, 257, 1023, 1024, 1025, 4097, 100_003, 262_144 + 5)


@pytest.fixture(scope="module")
def norm(pkg):
    from dxt_lossless_transform_amd import normalize as mod

    return mod


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [NONE, COLOR0, REPL])
def test_gpu_normalize_blocks(pkg, norm, oracle, mode):
    dev = torch.device("cuda:0")
    for n in SIZES:
        x = crafted_blocks(oracle, n, n)
        want = oracle.normalize_bc1_blocks(x, mode)
        xd = torch.from_numpy(x).to(dev)
        yd = torch.full((x.size + 64,), 0x5A, dtype=torch.uint8, device=dev)
        norm.normalize_blocks(xd, yd[: x.size], norm.ColorNormalizationMode(mode))
        assert np.array_equal(yd[: x.size].cpu().numpy(), want), (n, mode)
        assert bool((yd[x.size:] == 0x5A).all())
        norm.normalize_blocks(xd, xd, norm.ColorNormalizationMode(mode))      # in place
        assert np.array_equal(xd.cpu().numpy(), want), (n, mode, "in place")
        if n > 2:                                                             # misaligned views: the element path
            big = torch.zeros(x.size + 32, dtype=torch.uint8, device=dev)
            src, dst = big[3: 3 + x.size], torch.zeros(x.size + 16, dtype=torch.uint8, device=dev)[8: 8 + x.size]
            src.copy_(torch.from_numpy(x).to(dev))
            norm.normalize_blocks(src, dst, norm.ColorNormalizationMode(mode))
            assert np.array_equal(dst.cpu().numpy(), want), (n, mode, "misaligned")
    # host pointers, separate and in place
    x = crafted_blocks(oracle, 70_001, 9)
    y = np.zeros_like(x)
    norm.normalize_blocks(x, y, norm.ColorNormalizationMode(mode))
    assert np.array_equal(y, oracle.normalize_bc1_blocks(x, mode))
    z = x.copy()
    norm.normalize_blocks(z, z, norm.ColorNormalizationMode(mode))
    assert np.array_equal(z, y)
    with pytest.raises(pkg.InvalidLength):
        norm.normalize_blocks(np.zeros(12, np.uint8), np.zeros(12, np.uint8), norm.ColorNormalizationMode(mode))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [NONE, COLOR0, REPL])
def test_gpu_normalize_split_in_place(norm, oracle, mode):
    dev = torch.device("cuda:0")
    for n in SIZES:
        x = crafted_blocks(oracle, n, n + 1).reshape(-1, 8)
        c, i = x[:, :4].reshape(-1).copy(), x[:, 4:].reshape(-1).copy()
        wc, wi = oracle.normalize_bc1_split_blocks(c, i, mode)
        cd, idd = torch.from_numpy(c).to(dev), torch.from_numpy(i).to(dev)
        norm.normalize_split_blocks_in_place(cd, idd, norm.ColorNormalizationMode(mode))
        assert np.array_equal(cd.cpu().numpy(), wc) and np.array_equal(idd.cpu().numpy(), wi), (n, mode)
        if n > 4:   # 4-byte aligned but not 16-byte aligned views
            pad_c, pad_i = torch.zeros(c.size + 16, dtype=torch.uint8, device=dev), torch.zeros(i.size + 16, dtype=torch.uint8, device=dev)
            vc, vi = pad_c[4: 4 + c.size], pad_i[12: 12 + i.size]
            vc.copy_(torch.from_numpy(c).to(dev)); vi.copy_(torch.from_numpy(i).to(dev))
            norm.normalize_split_blocks_in_place(vc, vi, norm.ColorNormalizationMode(mode))
            assert np.array_equal(vc.cpu().numpy(), wc) and np.array_equal(vi.cpu().numpy(), wi), (n, mode, "views")
            assert int(pad_c[:4].sum()) == 0 and int(pad_i[:12].sum()) == 0
        hc, hi = c.copy(), i.copy()
        norm.normalize_split_blocks_in_place(hc, hi, norm.ColorNormalizationMode(mode))
        assert np.array_equal(hc, wc) and np.array_equal(hi, wi)


@pytest.mark.gpu
def test_gpu_normalize_all_modes(norm, oracle):
    dev = torch.device("cuda:0")
    for n in SIZES:
        x = crafted_blocks(oracle, n, n + 2)
        want, want_any = oracle.normalize_bc1_blocks_all_modes(x)
        xd = torch.from_numpy(x).to(dev)
        outs = [torch.empty_like(xd) for _ in range(3)]
        assert norm.normalize_blocks_all_modes(xd, outs) == want_any
        for m in range(3):
            assert np.array_equal(outs[m].cpu().numpy(), want[m]), (n, m)
        houts = [np.zeros_like(x) for _ in range(3)]
        assert norm.normalize_blocks_all_modes(x, houts) == want_any
        for m in range(3):
            assert np.array_equal(houts[m], want[m])
    plain = oracle.fill_splitmix64(8 * 4096, 5)
    pd = torch.from_numpy(plain).to(dev)
    assert norm.normalize_blocks_all_modes(pd, [torch.empty_like(pd) for _ in range(3)]) is False


@pytest.mark.gpu
def test_gpu_fused_transform_with_normalize(pkg, norm, oracle):
    dev = torch.device("cuda:0")
    D = norm.Bc1TransformDetailsWithNormalization
    for n in SIZES:
        x = crafted_blocks(oracle, n, n + 3)
        xd = torch.from_numpy(x).to(dev)
        for details in D.all_combinations():
            m, v, s = int(details.color_normalization_mode), details.decorrelation_mode, details.split_colour_endpoints
            want = oracle.transform_bc1_with_normalize_blocks(x, m, v, s)
            yd = torch.full((x.size + 64,), 0x5A, dtype=torch.uint8, device=dev)
            norm.transform_bc1_with_normalize_blocks(xd, yd[: x.size], details)
            assert np.array_equal(yd[: x.size].cpu().numpy(), want), (n, details)
            assert bool((yd[x.size:] == 0x5A).all())
            # the inverse of the plain transform gives back the NORMALISED blocks
            zd = torch.empty_like(xd)
            pkg.untransform_bc1_with_settings(yd[: x.size], zd, details.untransform_settings())
            assert np.array_equal(zd.cpu().numpy(), oracle.normalize_bc1_blocks(x, m))
    # every kernel path: aligned tiles / shifted tiles / element kernel, and the host-pointer entry point
    x = crafted_blocks(oracle, 300_000 + 1, 11)
    xd = torch.from_numpy(x).to(dev)
    det = D(norm.ColorNormalizationMode.REPLICATE_COLOR, 1, True)
    want = oracle.transform_bc1_with_normalize_blocks(x, 2, 1, True)
    try:
        for force in (0, 1, 2):
            pkg.set_tuning(0, force)
            yd = torch.empty_like(xd)
            norm.transform_bc1_with_normalize_blocks(xd, yd, det)
            assert np.array_equal(yd.cpu().numpy(), want), force
    finally:
        pkg.set_tuning(0, 0)
    y = np.zeros_like(x)
    norm.transform_bc1_with_normalize_blocks(x, y, det)
    assert np.array_equal(y, want)
    # normalisation is refused for anything but the BC1 forward transform
    l = pkg.load()
    assert l.dxtlt_transform_bc1_with_normalize_blocks(x.ctypes.data, y.ctypes.data, None, x.size, 3, 1, True) == 2


@pytest.mark.gpu
def test_gpu_fused_large_equals_two_kernels(pkg, norm, oracle):
    """1 GiB: fused normalise+transform == normalise kernel followed by the transform kernel (both checked against the
    oracle above at small sizes); prints the fused kernel's rate."""
    dev = torch.device("cuda:0")
    n = (1 << 30) // 8
    x = torch.empty(n * 8, dtype=torch.uint8, device=dev)
    pkg.fill_splitmix64(x, 0x0BC1_4E01)
    b = x.view(-1, 8)
    k = torch.arange(n, device=dev) % 4
    b[k == 1, 4:] = 0                                  # solid blocks
    rows = (k == 2).nonzero().squeeze(1)
    b[rows, 0:2] = 0
    b[rows, 4:] = 0xFF                                 # c0 = 0 <= c1, all indices 3: transparent
    del rows, k
    det = norm.Bc1TransformDetailsWithNormalization(norm.ColorNormalizationMode.COLOR0_ONLY, 1, True)
    fused, tmp, two = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    norm.transform_bc1_with_normalize_blocks(x, fused, det)
    norm.normalize_blocks(x, tmp, det.color_normalization_mode)
    pkg.transform_bc1_with_settings(tmp, two, det.untransform_settings())
    assert torch.equal(fused, two)
    assert not torch.equal(tmp, x)
    window = slice(123_456 * 8, (123_456 + 65_536) * 8)
    assert np.array_equal(tmp[window].cpu().numpy(), oracle.normalize_bc1_blocks(x[window].cpu().numpy(), 1))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(10):
        norm.transform_bc1_with_normalize_blocks(x, fused, det)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 10
    print(f"fused normalise+transform, 1 GiB: {ms:.3f} ms, {2 * x.numel() / ms / 1e-3 / 8e12:.3f} of 8 TB/s")


@pytest.mark.gpu
@pytest.mark.parametrize("use_all", [False, True])
def test_gpu_auto_with_normalization(pkg, oracle, use_all):
    from tests import cabi

    l = pkg.load()
    f = l.dxtlt_transform_bc1_auto_with_normalization
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(cabi.DltSizeEstimator), C.c_bool,
                  C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.POINTER(C.c_bool), C.POINTER(C.c_uint32)]
    f.restype = C.c_int32

    def run(x, kind, log=None):
        est, py_est = cabi.make_estimator(kind, log)
        y = np.zeros_like(x)
        m, v, s, err = C.c_uint8(9), C.c_uint8(9), C.c_bool(False), C.c_uint32(0)
        rc = f(x.ctypes.data, y.ctypes.data, x.size, C.byref(est), use_all, C.byref(m), C.byref(v), C.byref(s),
               C.byref(err))
        return rc, y, (m.value, v.value, int(s.value)), err.value, py_est

    for n in (1, 9, 700, 20_001):
        x = crafted_blocks(oracle, n, n + 7)
        log = []
        rc, y, choice, err, py_est = run(x, "zlib", log)
        want_choice, want_out, want_calls = oracle_auto.transform_bc1_auto_with_normalization(
            x, lambda b: py_est(bytes(b)), use_all)
        assert rc == 0 and err == 0
        assert choice == tuple(int(c) for c in want_choice), (n, choice, want_choice)
        assert np.array_equal(y, want_out)
        if n == 1:   # block 0 of the crafted data is random: nothing to normalise, the plain auto transform ran
            assert want_choice[0] == 0 and log == [ln for _off, ln in want_calls] and len(log) == (8 if use_all else 4)
        else:
            assert log == want_calls and len(log) == 3 * (8 if use_all else 4)
    # nothing to normalise: the plain transform_bc1_auto runs (4 / 8 estimator calls), mode None is reported
    plain = oracle.fill_splitmix64(8 * 3000, 17)
    log = []
    rc, y, choice, err, py_est = run(plain, "zlib", log)
    (v, _sa, sc), want_out, _ = oracle_auto.transform_auto("bc1", plain, lambda b: py_est(bytes(b)), use_all)
    assert rc == 0 and choice == (0, v, sc) and np.array_equal(y, want_out) and len(log) == (8 if use_all else 4)
    # estimator failures: max_compressed_size propagates; a failing estimate only skips candidates, so the defaults
    # {None, Variant1, split} win and the output is the plain default transform
    x = crafted_blocks(oracle, 500, 23)
    rc, _, _, err, _ = run(x, "fail_max")
    assert rc == 5 and err == 41
    rc, y, choice, err, _ = run(x, "fail_est")
    assert rc == 0 and choice == (0, 1, 1)
    assert np.array_equal(y, oracle.transform("bc1", x, 1, True))
    # the dummy estimator (size = len for every candidate): strict `<` keeps the first candidate tried
    rc, y, choice, _, _ = run(x, "dummy")
    first = oracle_auto.test_order("bc1", use_all)[0]
    assert rc == 0 and choice == (0, first[0], first[2])
    # argument checks
    est, _ = cabi.make_estimator("dummy")
    assert f(x.ctypes.data, y.ctypes.data, 12, C.byref(est), use_all, None, None, None, None) == 1
    assert f(x.ctypes.data, y.ctypes.data, x.size, None, use_all, None, None, None, None) == 2
