"""BC1 / BC2 / BC3 block decoders -- the reference's util modules (decode_bcN_block:
/root/reference/src/core/dxt-lossless-transform-bc1/src/util/bc1_decode.rs:42, -bc2/src/util/bc2_decode.rs:44,
-bc3/src/util/bc3_decode.rs:43) as array operations of include/dxtlt_decode.h.

CPU: the oracle replays the reference's decoder unit vectors (so this row is PINNED); the device header
csrc/bcn_decode.h built for the host equals the oracle on structured and random blocks; its small-divisor tricks are
checked exhaustively.  GPU (-m gpu): every entry point through the C ABI against the oracle."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOCK = {"bc1": 8, "bc2": 16, "bc3": 16}
KIND = {"bc1": 1, "bc2": 2, "bc3": 3}


def u8(*v):
    return np.array(v, dtype=np.uint8)


def px(data):
    return [tuple(int(v) for v in p) for p in np.asarray(data).reshape(-1, 4)]


# ---------------------------------------------------------------------------------------------------------------
# reference unit vectors
# ---------------------------------------------------------------------------------------------------------------
def test_ref_bc1_vectors(oracle):
    """bc1_decode.rs tests: solid red; three-colour mode with every index 3 = transparent black."""
    assert px(oracle.decode_blocks("bc1", u8(0x00, 0xF8, 0x00, 0xF8, 0, 0, 0, 0))) == [(255, 0, 0, 255)] * 16
    assert px(oracle.decode_blocks("bc1", u8(0x00, 0xF0, 0x00, 0xF8, 0xFF, 0xFF, 0xFF, 0xFF))) == [(0, 0, 0, 0)] * 16


def test_ref_bc2_vectors(oracle):
    """bc2_decode.rs tests: red with the sixteen 4-bit alphas 0..15 (x17); red with zero alpha."""
    red = [0x00, 0xF8, 0x00, 0xF8, 0, 0, 0, 0]
    got = px(oracle.decode_blocks("bc2", u8(0x10, 0x32, 0x54, 0x76, 0x98, 0xBA, 0xDC, 0xFE, *red)))
    assert got == [(255, 0, 0, 17 * i) for i in range(16)]
    assert px(oracle.decode_blocks("bc2", u8(*([0] * 8), *red))) == [(255, 0, 0, 0)] * 16


REF_BC3 = [
    # bc3_decode.rs can_decode_bc3_block
    ((0, 0, 0, 255, 255, 255, 255, 255, 255, 255, 18, 0, 0, 0, 0, 250),
     [(255, 255, 255, 0)] * 3 + [(255, 255, 255, 255)] * 9 + [(170, 170, 219, 255)] * 2 + [(85, 85, 183, 255)] * 2),
    # can_decode_bc3_block_with_varying_alpha
    ((41, 1, 253, 178, 0, 0, 0, 0, 10, 0, 0, 0, 0, 0, 77, 0),
     [(0, 0, 82, 18), (0, 0, 82, 6), (0, 0, 82, 29), (0, 0, 82, 1), (0, 0, 82, 29), (0, 0, 82, 1), (0, 0, 82, 41), (0, 0, 82, 41),
      (0, 0, 0, 41), (0, 0, 27, 41), (0, 0, 82, 41), (0, 0, 0, 41)] + [(0, 0, 82, 41)] * 4),
    # can_decode_bc3_block_with_fixed_alpha
    ((221, 0, 0, 0, 0, 0, 0, 0, 10, 0, 0, 0, 0, 0, 212, 0),
     [(0, 0, 82, 221)] * 8 + [(0, 0, 82, 221), (0, 0, 0, 221), (0, 0, 0, 221), (0, 0, 27, 221)] + [(0, 0, 82, 221)] * 4),
]


@pytest.mark.parametrize("block,want", REF_BC3)
def test_ref_bc3_vectors(oracle, block, want):
    assert px(oracle.decode_blocks("bc3", u8(*block))) == want


def test_oracle_array_form_equals_block_form(oracle):
    rng = np.random.default_rng(5)
    x = rng.integers(0, 256, 8 * 100, dtype=np.uint8)
    got = oracle.decode_blocks("bc1", x).reshape(100, 16, 4)
    for i in range(100):
        assert np.array_equal(got[i], oracle.decode_bc1_block(x[8 * i:8 * i + 8]))
    y = x.copy()
    y[8 * 7 + 4] ^= 1       # one index of block 7; random endpoints differ, so its pixels change
    y[8 * 50] ^= 0x10       # an endpoint bit of block 50
    assert oracle.count_pixel_differences("bc1", x, x) == 0
    assert oracle.count_pixel_differences("bc1", x, y) == 2


# ---------------------------------------------------------------------------------------------------------------
# the device header, built for the host
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def shim(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("shim") / "decode_header_shim.so")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-shared", "-fPIC", "-Wall", "-Wextra", "-o", so,
                           os.path.join(ROOT, "tests", "cpp", "normalize_header_shim.cpp")])
    lib = ctypes.CDLL(so)
    lib.shim_decode_blocks.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.shim_decode_blocks.restype = None
    lib.shim_small_division.argtypes, lib.shim_small_division.restype = [ctypes.c_int, ctypes.c_uint32], ctypes.c_uint32
    return lib


def test_small_divisions_are_exact(shim):
    """x / 3 up to 3 * 255, x / 5 up to 4 * 255, x / 7 up to 7 * 255: the ranges the decoders feed them."""
    for d, top in ((3, 765), (5, 1020), (7, 1785)):
        assert all(shim.shim_small_division(d, x) == x // d for x in range(top + 1))


def decode_cases(fmt, rng, n_random=200_000):
    """Random blocks plus blocks that walk the special cases: c0 == c1, c0 < c1, c0 > c1 with every index pattern;
    every alpha endpoint pair (BC3) with an index ramp."""
    bs = BLOCK[fmt]
    parts = [rng.integers(0, 256, bs * n_random, dtype=np.uint8).reshape(-1, bs)]
    c = rng.integers(0, 65536, 20_000, dtype=np.uint32)
    for other in (c, c ^ 1, (c + 1) & 0xFFFF, c ^ 0x8000):
        h = np.empty((c.size, 8), dtype=np.uint8)
        h[:, 0], h[:, 1], h[:, 2], h[:, 3] = c & 255, c >> 8, other & 255, other >> 8
        h[:, 4:] = rng.integers(0, 256, (c.size, 4), dtype=np.uint8)
        if fmt == "bc1":
            parts.append(h)
        else:
            parts.append(np.concatenate([rng.integers(0, 256, (c.size, 8), dtype=np.uint8), h], axis=1))
    if fmt == "bc3":
        e = rng.integers(0, 256, (65536, 16), dtype=np.uint8)
        e[:, 0], e[:, 1] = np.arange(65536) & 255, np.arange(65536) >> 8
        e[:, 2:8] = np.frombuffer(int(0o7654321076543210).to_bytes(6, "little"), dtype=np.uint8)   # indices 0..7 twice
        parts.append(e)
    return np.ascontiguousarray(np.concatenate(parts).reshape(-1))


@pytest.mark.parametrize("fmt", ["bc1", "bc2", "bc3"])
def test_device_header_equals_oracle(oracle, shim, fmt):
    x = decode_cases(fmt, np.random.default_rng(0xDEC0 + KIND[fmt]))
    n = x.size // BLOCK[fmt]
    got = np.empty(64 * n, dtype=np.uint8)
    shim.shim_decode_blocks(KIND[fmt], x.ctypes.data, got.ctypes.data, n)
    want = oracle.decode_blocks(fmt, x)
    bad = np.flatnonzero((got.reshape(n, 64) != want.reshape(n, 64)).any(axis=1))
    assert bad.size == 0, (fmt, x.reshape(n, -1)[bad[:2]], got.reshape(n, 64)[bad[:2]], want.reshape(n, 64)[bad[:2]])


def test_validation_without_a_device(pkg):
    from dxt_lossless_transform_amd import decode as mod

    l = mod._l()
    buf = np.zeros(256, dtype=np.uint8)
    p = buf.ctypes.data
    assert l.dxtlt_decode_bc1_blocks(p, 12, p, 256) == 1
    assert l.dxtlt_decode_bc3_blocks(p, 24, p, 256) == 1
    assert l.dxtlt_decode_bc1_blocks(p, 16, p, 127) == 2        # two blocks need 128 bytes
    assert l.dxtlt_decode_bc2_blocks(None, 16, p, 64) == 2
    assert l.dxtlt_decode_bc2_blocks(p, 0, p, 0) == 0
    out = ctypes.c_uint64(7)
    assert l.dxtlt_count_pixel_differences(4, p, p, 16, ctypes.byref(out)) == 2
    assert l.dxtlt_count_pixel_differences(1, p, p, 12, ctypes.byref(out)) == 1
    assert l.dxtlt_count_pixel_differences(1, p, p, 0, ctypes.byref(out)) == 0 and out.value == 0
    assert l.dxtlt_count_pixel_differences(1, p, p, 8, None) == 2
    with pytest.raises(pkg.InvalidLength):
        mod.decode_blocks("bc1", buf[:12], buf)
    with pytest.raises(pkg.OutputBufferTooSmall):
        mod.decode_blocks("bc2", buf[:64], buf[:255])
    with pytest.raises(pkg.InvalidLength):
        mod.count_pixel_differences("bc1", buf[:16], buf[:24])


# ---------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def dec(pkg):
    from dxt_lossless_transform_amd import decode as mod

    return mod


@pytest.mark.gpu
def test_gpu_reference_vectors(dec):
    out = np.zeros(64, dtype=np.uint8)
    dec.decode_blocks("bc1", u8(0x00, 0xF8, 0x00, 0xF8, 0, 0, 0, 0), out)
    assert px(out) == [(255, 0, 0, 255)] * 16
    dec.decode_blocks("bc1", u8(0x00, 0xF0, 0x00, 0xF8, 0xFF, 0xFF, 0xFF, 0xFF), out)
    assert px(out) == [(0, 0, 0, 0)] * 16
    dec.decode_blocks("bc2", u8(0x10, 0x32, 0x54, 0x76, 0x98, 0xBA, 0xDC, 0xFE, 0x00, 0xF8, 0x00, 0xF8, 0, 0, 0, 0), out)
    assert px(out) == [(255, 0, 0, 17 * i) for i in range(16)]
    for block, want in REF_BC3:
        dec.decode_blocks("bc3", u8(*block), out)
        assert px(out) == want


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["bc1", "bc2", "bc3"])
def test_gpu_decode_equals_oracle(dec, oracle, fmt):
    """Structured + random blocks (every alpha endpoint pair, every colour mode), host and device entry points."""
    import torch

    x = decode_cases(fmt, np.random.default_rng(0xD0 + KIND[fmt]), n_random=100_003)
    n = x.size // BLOCK[fmt]
    want = oracle.decode_blocks(fmt, x)
    got = np.zeros(64 * n, dtype=np.uint8)
    dec.decode_blocks(fmt, x, got)
    bad = np.flatnonzero((got.reshape(n, 64) != want.reshape(n, 64)).any(axis=1))
    assert bad.size == 0, (fmt, x.reshape(n, -1)[bad[:2]], got.reshape(n, 64)[bad[:2]], want.reshape(n, 64)[bad[:2]])
    d = torch.from_numpy(x).cuda()
    o = torch.full((64 * n + 64,), 0xEE, dtype=torch.uint8, device="cuda")
    dec.decode_blocks(fmt, d, o[:64 * n])
    got = o.cpu().numpy()
    assert np.array_equal(got[:64 * n], want) and (got[64 * n:] == 0xEE).all()


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["bc1", "bc2", "bc3"])
@pytest.mark.parametrize("n", [1, 2, 15, 16, 17, 63, 64, 65, 255, 256, 257, 1000, 4099])
def test_gpu_decode_ragged_and_misaligned(dec, oracle, fmt, n):
    import torch

    rng = np.random.default_rng(n + KIND[fmt])
    bs = BLOCK[fmt]
    x = rng.integers(0, 256, bs * n, dtype=np.uint8)
    want = oracle.decode_blocks(fmt, x)
    for in_shift, out_shift in ((0, 0), (1, 0), (0, 4), (3, 7), (8, 16)):
        src = torch.zeros(bs * n + 64, dtype=torch.uint8, device="cuda")
        src[in_shift:in_shift + bs * n] = torch.from_numpy(x).cuda()
        dst = torch.full((64 * n + 128,), 0x77, dtype=torch.uint8, device="cuda")
        dec.decode_blocks(fmt, src[in_shift:in_shift + bs * n], dst[out_shift:out_shift + 64 * n])
        got = dst.cpu().numpy()
        assert np.array_equal(got[out_shift:out_shift + 64 * n], want), (fmt, n, in_shift, out_shift)
        assert (got[:out_shift] == 0x77).all() and (got[out_shift + 64 * n:] == 0x77).all()


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["bc1", "bc2", "bc3"])
def test_gpu_count_pixel_differences(dec, oracle, fmt):
    import torch

    rng = np.random.default_rng(0xD1FF + KIND[fmt])
    bs = BLOCK[fmt]
    n = 70_001
    a = rng.integers(0, 256, bs * n, dtype=np.uint8)
    b = a.copy()
    touched = rng.choice(n, 3000, replace=False)
    for i, blk in enumerate(touched):
        b[bs * blk + int(rng.integers(0, bs))] ^= 1 << (i % 8)
    # byte differences that keep the pixels: swap in an equivalent encoding (c0 == c1, any indices -> indices 0) for BC2/3
    col = bs - 8
    same = np.setdiff1d(np.arange(n), touched)[:500]
    for blk in same:
        o = bs * blk + col
        a[o + 2:o + 4] = a[o:o + 2]
        b[o:o + 4] = a[o:o + 4]
        if fmt == "bc1":
            a[o + 4:o + 8] = 0x55         # c0 == c1 in BC1 is three-colour mode: indices 0 and 1 show the same colour
            b[o + 4:o + 8] = 0x00
        else:
            b[o + 4:o + 8] = rng.integers(0, 256, 4, dtype=np.uint8)   # four equal palette entries: indices are free
    want = oracle.count_pixel_differences(fmt, a, b)
    assert 0 < want <= 3000
    assert dec.count_pixel_differences(fmt, a, b) == want
    assert dec.count_pixel_differences(fmt, a, a) == 0
    da, db = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    assert dec.count_pixel_differences(fmt, da, db) == want
    pad = torch.zeros(bs * n + 8, dtype=torch.uint8, device="cuda")
    pad[3:3 + bs * n] = da
    assert dec.count_pixel_differences(fmt, pad[3:3 + bs * n], db) == want          # unaligned path


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["bc1", "bc2", "bc3"])
def test_gpu_normalisation_keeps_every_pixel(pkg, dec, fmt):
    """The property the reference's normalisation tests assert per block, on 64 MiB of mixed blocks at once:
    decode(normalize(x)) == decode(x) for every mode."""
    import torch
    from dxt_lossless_transform_amd import normalize as n1, normalize23 as n23

    bs = BLOCK[fmt]
    n = (64 << 20) // bs
    x = torch.empty(bs * n, dtype=torch.uint8, device="cuda")
    pkg.fill_splitmix64(x, 0xDEC0DE00 + KIND[fmt])
    v = x.view(n, bs)
    k = torch.arange(n, device="cuda") % 8
    col = bs - 8
    v[k == 1, col + 4:] = 0                                  # one index: solid blocks
    v[k == 2, col + 2:col + 4] = v[k == 2, col:col + 2]      # c0 == c1
    if fmt == "bc1":
        rows = (k == 3).nonzero().squeeze(1)
        v[rows, 0:2] = 0
        v[rows, 4:] = 0xFF                                   # transparent blocks
    if fmt == "bc3":
        v[k == 4, 2:8] = 0                                   # uniform alpha
        v[k == 5, 0:2] = 0xFF
    y = torch.empty_like(x)
    if fmt == "bc1":
        for mode in (n1.ColorNormalizationMode.COLOR0_ONLY, n1.ColorNormalizationMode.REPLICATE_COLOR):
            n1.normalize_blocks(x, y, mode)
            assert not torch.equal(x, y)
            assert dec.count_pixel_differences(fmt, x, y) == 0
    else:
        alpha_modes = [n23.AlphaNormalizationMode.NONE] if fmt == "bc2" else list(n23.AlphaNormalizationMode)
        for am in alpha_modes:
            for cm in (n1.ColorNormalizationMode.COLOR0_ONLY, n1.ColorNormalizationMode.REPLICATE_COLOR):
                n23.normalize_blocks(fmt, x, y, cm, am)
                assert not torch.equal(x, y)
                assert dec.count_pixel_differences(fmt, x, y) == 0
