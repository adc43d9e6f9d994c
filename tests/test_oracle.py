"""The CPU oracle against everything the reference's own tests pin for this path, against the
independently written numpy restatement, and against the committed golden fixtures.  CPU only."""
import hashlib

import numpy as np
import pytest

from helpers import BLOCK, FORMATS, all_settings, golden_digests, golden_vectors, payload, settings_id
from oracle import oracle_np as onp


# ---- known answers held by the reference ------------------------------------------------------------
@pytest.mark.parametrize("fmt", FORMATS)
def test_generator_known_answer(oracle, fmt):
    # bc1 test_prelude.rs:107-119, bc2 :586-606, bc3 :1058-1078
    want = bytes.fromhex(golden_vectors()["known_answers"][f"{fmt}_generator_3"])
    assert oracle.generate_test_data(fmt, 3).tobytes() == want


def test_split_565_known_answer(oracle):
    # common/src/transforms/split_565_color_endpoints/tests.rs:140-152
    ka = golden_vectors()["known_answers"]
    got = oracle.split_565_color_endpoints(bytes.fromhex(ka["split_565_input"]))
    assert got.tobytes() == bytes.fromhex(ka["split_565_output"])
    # the same vector through the BC1 "split colour, no decorrelation" layout: colours section == split output
    blocks = np.zeros(3 * 8, dtype=np.uint8)
    blocks.reshape(3, 8)[:, :4] = np.frombuffer(bytes.fromhex(ka["split_565_input"]), dtype=np.uint8).reshape(3, 4)
    out = oracle.transform("bc1", blocks, 0, True)
    assert out[:12].tobytes() == bytes.fromhex(ka["split_565_output"])


# hand-derived from the formulas at decorrelate.rs:101-127,187-212,274-299
YCOCG_KAT = {
    0xF800: (0xBFD1, 0x5FF1, 0xBFE2), 0x07E0: (0x783F, 0xBC1F, 0x783F), 0x001F: (0xF841, 0x7C21, 0xF842),
    0xFFFF: (0xF820, 0xFC00, 0xF801), 0x1234: (0x0BAD, 0x85CD, 0x0B9B), 0x5678: (0x6CB8, 0xB658, 0x6CB1),
    0x9ABC: (0x45E3, 0xA2E3, 0x45C7), 0xDEF0: (0xC2E6, 0xE166, 0xC2CD), 0x0100: (0x1004, 0x0804, 0x1008),
    0x0302: (0xF79B, 0x7BDB, 0xF7B6),
}


def test_ycocg_known_answers(oracle):
    for c, want in YCOCG_KAT.items():
        assert tuple(oracle.decorrelate(c, v) for v in (1, 2, 3)) == want, hex(c)


def test_ycocg_reference_colour_set_roundtrips(oracle):
    # decorrelate.rs:413-446 and intrinsics/.../avx2.rs:194-266 pin recorrelate(decorrelate(c)) == c
    for c in golden_vectors()["known_answers"]["ycocg_roundtrip_colours"]:
        for v in range(4):
            assert oracle.recorrelate(oracle.decorrelate(c, v), v) == c


def test_ycocg_exhaustive_bijection_and_np_agreement(oracle):
    allc = np.arange(65536, dtype=np.uint16)
    for v in range(4):
        d_np = onp.decorrelate(allc, v)
        d_c = np.array([oracle.decorrelate(int(c), v) for c in allc], dtype=np.uint16)
        assert np.array_equal(d_np, d_c)
        assert len(np.unique(d_c)) == 65536
        assert np.array_equal(onp.recorrelate(d_np, v), allc)
        assert all(oracle.recorrelate(int(x), v) == i for i, x in enumerate(d_c))


def test_survey_layout_vectors(oracle):
    # BC1 generator n=3 and BC3 generator n=2 worked by hand from the layout rules
    g = oracle.generate_test_data("bc1", 3)
    assert oracle.transform("bc1", g, 0, False).tobytes().hex() == "000102030405060708090a0b808182838485868788898a8b"
    assert oracle.transform("bc1", g, 0, True).tobytes().hex() == "000104050809020306070a0b808182838485868788898a8b"
    assert oracle.transform("bc1", g, 1, False).tobytes().hex() == "04109bf7029f89be50e6d705808182838485868788898a8b"
    assert oracle.transform("bc1", g, 1, True).tobytes().hex() == "0410029f50e69bf789bed705808182838485868788898a8b"
    g3 = oracle.generate_test_data("bc3", 2)
    tail = "c0c1c2c3c4c5c6c7"
    aidx = "202122232425262728292a2b"
    assert oracle.transform("bc3", g3, 0, False, False).tobytes().hex() == "00010203" + aidx + "8081828384858687" + tail
    assert oracle.transform("bc3", g3, 0, False, True).tobytes().hex() == "00020103" + aidx + "8081828384858687" + tail
    assert oracle.transform("bc3", g3, 0, True, False).tobytes().hex() == "00010203" + aidx + "8081848582838687" + tail
    assert oracle.transform("bc3", g3, 1, True, True).tobytes().hex() == "00020103" + aidx + "1ebc0c83855b93a2" + tail


# ---- the reference's round-trip-for-every-n harness (bc1 test_prelude.rs:154-317 and twins) ---------------
@pytest.mark.parametrize("fmt", FORMATS)
def test_roundtrip_every_n(oracle, fmt):
    for n in range(0, 130):
        x = oracle.generate_test_data(fmt, n)
        for v, sa, sc in all_settings(fmt):
            y = oracle.transform(fmt, x, v, sc, sa)
            assert y.size == x.size
            assert np.array_equal(oracle.transform(fmt, y, v, sc, sa, inverse=True), x), (n, v, sa, sc)


@pytest.mark.parametrize("fmt", FORMATS)
def test_unaligned_pointers(oracle, fmt):
    # bc1 test_prelude.rs:364-373: input and output offset by one byte
    n = 37
    x = oracle.generate_test_data(fmt, n)
    for v, sa, sc in all_settings(fmt):
        want = oracle.transform(fmt, x, v, sc, sa)
        src = np.zeros(x.size + 1, dtype=np.uint8)
        src[1:] = x
        dst = np.zeros(x.size + 1, dtype=np.uint8)
        f = getattr(oracle.lib(), f"oracle_transform_{fmt}")
        args = (src.ctypes.data + 1, dst.ctypes.data + 1, x.size, v) + ((sa, sc) if fmt == "bc3" else (sc,))
        f(*args)
        assert np.array_equal(dst[1:], want)


# ---- two independent restatements agree ----------------------------------------------------------------
@pytest.mark.parametrize("fmt", FORMATS)
def test_c_oracle_matches_numpy_restatement(oracle, fmt):
    rng = np.random.default_rng(1234)
    for n in (0, 1, 2, 3, 7, 64, 257, 4099):
        x = rng.integers(0, 256, n * BLOCK[fmt], dtype=np.uint8)
        for v, sa, sc in all_settings(fmt):
            a = oracle.transform(fmt, x, v, sc, sa)
            b = onp.transform(fmt, x, v, bool(sc), bool(sa))
            assert np.array_equal(a, b), (n, v, sa, sc)
            assert np.array_equal(onp.transform(fmt, a, v, bool(sc), bool(sa), inverse=True), x)


# ---- committed fixtures ----------------------------------------------------------------------------------
def test_golden_vectors(oracle):
    for e in golden_vectors()["vectors"]:
        x = np.frombuffer(bytes.fromhex(e["input"]), dtype=np.uint8)
        y = oracle.transform(e["fmt"], x, e["variant"], e["split_colour"], e["split_alpha"])
        assert y.tobytes().hex() == e["output"], e
        if e["source"] == "generator":
            assert np.array_equal(oracle.generate_test_data(e["fmt"], e["blocks"]), x)


def test_golden_digests(oracle):
    cache = {}
    for e in golden_digests():
        key = (e["fmt"], e["source"], e.get("seed"), e["blocks"])
        if key not in cache:
            if e["source"] == "splitmix64":
                cache[key] = oracle.fill_splitmix64(e["blocks"] * BLOCK[e["fmt"]], e["seed"])
            else:
                cache[key] = payload(e["fmt"])
        x = cache[key]
        assert hashlib.sha256(x).hexdigest() == e["input_sha256"]
        y = oracle.transform(e["fmt"], x, e["variant"], e["split_colour"], e["split_alpha"])
        assert hashlib.sha256(y).hexdigest() == e["output_sha256"], e
        assert np.array_equal(oracle.transform(e["fmt"], y, e["variant"], e["split_colour"], e["split_alpha"],
                                               inverse=True), x)


# ---- safe-wrapper validation (bc1 safe/transform_with_settings.rs:88-118, tests :225-330) -----------------
@pytest.mark.parametrize("fmt", FORMATS)
def test_safe_wrapper_validation(oracle, fmt):
    b = BLOCK[fmt]
    x = oracle.generate_test_data(fmt, 2)
    rc, _ = oracle.transform_safe(fmt, x[: 2 * b - 1], 2 * b)
    assert rc == oracle.INVALID_LENGTH
    rc, _ = oracle.transform_safe(fmt, x, 2 * b - 1)
    assert rc == oracle.OUTPUT_TOO_SMALL
    # length is checked before size
    rc, _ = oracle.transform_safe(fmt, x[: b + 1], 1)
    assert rc == oracle.INVALID_LENGTH
    rc, out = oracle.transform_safe(fmt, x, 2 * b + 5)
    assert rc == oracle.OK and np.array_equal(out[: 2 * b], oracle.transform(fmt, x))
    rc, back = oracle.transform_safe(fmt, out[: 2 * b], 2 * b, inverse=True)
    assert rc == oracle.OK and np.array_equal(back, x)


# ---- workload generator ----------------------------------------------------------------------------------
def test_splitmix_matches_numpy(oracle):
    a = oracle.fill_splitmix64(4099, 0x0BC10002, 5)
    b = onp.splitmix64(0x0BC10002, 5, 513).astype("<u8").view(np.uint8)[:4099]
    assert np.array_equal(a, b)
    # a range of the stream equals the same range generated stand-alone
    whole = oracle.fill_splitmix64(8 * 100, 7)
    assert np.array_equal(whole[8 * 40:], oracle.fill_splitmix64(8 * 60, 7, 40))


def test_transform_helps_a_generic_compressor_on_real_textures(oracle):
    """The point of the transform (reference README: zstd-16 5.695 -> 4.857 GiB on a BC1 corpus): the transformed real
    textures deflate smaller than the raw block arrays.  zlib stands in for zstd (not installed here); random blocks
    would show ratio ~1, so BASELINE configs[4]'s ratio check is done on the reference's own test textures."""
    import zlib

    for fmt in FORMATS:
        p = payload(fmt)
        raw = len(zlib.compress(p.tobytes(), 6))
        best = min(len(zlib.compress(oracle.transform(fmt, p, v, sc, sa).tobytes(), 6)) for v, sa, sc in all_settings(fmt))
        default = len(zlib.compress(oracle.transform(fmt, p).tobytes(), 6))
        assert best < raw, (fmt, raw, best)
        assert default < raw * 1.02, (fmt, raw, default)  # the default is near the best, never a big loss


def test_simd_ports_equal_scalar_oracle(oracle):
    """The vectorised CPU baselines (AVX2 and AVX-512BW ports of the reference's SIMD strategy, BC1 default settings)
    must be the same function as the scalar oracle: every block count around the 16- and 32-block vector widths,
    several thread counts, every vector level this CPU has."""
    top = oracle.simd_level()
    try:
        for cap in (0, 2, 5):
            level = oracle.simd_set_cap(cap)
            assert level <= min(cap, top)
            for n in list(range(0, 70)) + [255, 256, 257, 100_003]:
                x = oracle.fill_splitmix64(n * 8, 0xA7C2 + n)
                want = oracle.transform("bc1", x, 1, True)
                for threads in (1, 3):
                    got = np.full_like(x, 0xEE)
                    oracle.run_bc1_default_simd(x, got, False, threads)
                    assert np.array_equal(got, want), (level, n, threads)
                    back = np.full_like(x, 0xEE)
                    oracle.run_bc1_default_simd(want, back, True, threads)
                    assert np.array_equal(back, x), (level, n, threads, "inverse")
    finally:
        oracle.simd_set_cap(5)


def test_bc2_default_and_bc3_standard_simd_ports_match_the_scalar_oracle(oracle):
    """The AVX2 ports of BC2 {Variant1, split colours} and BC3 "standard" {None, no splits} (the two paths bench.py's
    cpu_baseline leg also quotes vectorised) equal the scalar oracle byte for byte at every ISA level, around the 8-block
    vector width, over several threads."""
    try:
        for n in (0, 1, 7, 8, 9, 15, 16, 17, 63, 64, 65, 999, 100_003):
            x = oracle.fill_splitmix64(n * 16, 0xB23 + n)
            for kind, fmt, (v, sc, sa) in ((2, "bc2", (1, True, False)), (3, "bc3", (0, False, False))):
                want = oracle.transform(fmt, x, v, sc, sa)
                for cap in (5, 2, 0):
                    oracle.simd_set_cap(cap)
                    for threads in (1, 3):
                        y = np.full_like(x, 0xEE)
                        oracle.run_bc23_simd(kind, x, y, False, threads)
                        assert np.array_equal(y, want), (fmt, n, cap, threads)
                        z = np.full_like(x, 0xEE)
                        oracle.run_bc23_simd(kind, y, z, True, threads)
                        assert np.array_equal(z, x), (fmt, n, cap, threads, "inverse")
    finally:
        oracle.simd_set_cap(5)


def test_mt_range_split_matches_single_thread(oracle):
    for fmt in FORMATS:
        x = oracle.fill_splitmix64(1001 * BLOCK[fmt], 99)
        for inverse in (False, True):
            want = oracle.transform(fmt, x, 1, True, True, inverse=inverse)
            got = np.empty_like(x)
            oracle.run_mt(fmt, x, got, 1, True, True, inverse, 3)
            assert np.array_equal(got, want)
