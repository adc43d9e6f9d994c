"""Array-level RGB565 colour operations (include/dxtlt_color565.h) -- the reference's common-crate entry points
Color565::decorrelate_ycocg_r_ptr / recorrelate_ycocg_r_ptr (decorrelate_batch_ptr.rs:336,378),
recorrelate_ycocg_r_ptr_split (decorrelate_batch_split_ptr.rs:324) and split_color_endpoints
(split_565_color_endpoints/mod.rs:110).

CPU: argument validation needs no device.  GPU (-m gpu): all 65 536 colours x 3 variants against the oracle's scalar
formulas and SURVEY.md section 8(c)'s hand-derived vectors, the reference's 3-pair split vector, in place, misaligned
pointers and ragged counts, through the C ABI (host and device entry points)."""
import numpy as np
import pytest

from test_oracle import YCOCG_KAT


def all_colours():
    return np.arange(65536, dtype="<u2")


def oracle_table(oracle, variant, inverse):
    f = oracle.recorrelate if inverse else oracle.decorrelate
    return np.array([f(int(c), variant) for c in range(65536)], dtype="<u2")


@pytest.fixture(scope="module")
def tables(oracle):
    return {(v, inv): oracle_table(oracle, v, inv) for v in (1, 2, 3) for inv in (False, True)}


def test_oracle_tables_are_inverse_bijections(tables):
    for v in (1, 2, 3):
        fwd, inv = tables[(v, False)], tables[(v, True)]
        assert np.array_equal(inv[fwd], all_colours())
        assert len(np.unique(fwd)) == 65536
        for c, want in YCOCG_KAT.items():
            assert fwd[c] == want[v - 1]


def test_validation_without_a_device(pkg):
    from dxt_lossless_transform_amd import color565 as mod

    l = mod._l()
    buf = np.zeros(64, dtype=np.uint8)
    p = buf.ctypes.data
    assert l.dxtlt_color565_decorrelate_ycocg_r(p, p, 8, 4) == 2
    assert l.dxtlt_color565_recorrelate_ycocg_r(p, p, 8, 9) == 2
    assert l.dxtlt_color565_decorrelate_ycocg_r(p, p, 0, 1) == 0
    assert l.dxtlt_color565_decorrelate_ycocg_r(None, p, 4, 1) == 2
    assert l.dxtlt_color565_recorrelate_ycocg_r_split(p, p, p, 3, 1) == 1
    assert l.dxtlt_color565_recorrelate_ycocg_r_split(p, p, p, 0, 1) == 0
    assert l.dxtlt_split_565_color_endpoints(p, p, 6) == 1
    assert l.dxtlt_split_565_color_endpoints(p, p, 0) == 0
    # variant None on host buffers is a copy and needs no device either (decorrelate_batch_ptr.rs:351-356)
    src = np.arange(32, dtype=np.uint8)
    dst = np.zeros(32, dtype=np.uint8)
    mod.decorrelate_ycocg_r(src, dst, 0)
    assert np.array_equal(src, dst)
    with pytest.raises(pkg.InvalidLength):
        mod.decorrelate_ycocg_r(src[:31], dst, 1)
    with pytest.raises(pkg.OutputBufferTooSmall):
        mod.recorrelate_ycocg_r(src, dst[:30], 1)
    with pytest.raises(pkg.InvalidLength):
        mod.split_color_endpoints(src[:30], dst)
    with pytest.raises(pkg.OutputBufferTooSmall):
        mod.recorrelate_ycocg_r_split(src[:16], src[16:], dst[:31], 1)


# ---------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------
def as_bytes(a):
    return np.ascontiguousarray(a).view(np.uint8)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [1, 2, 3])
def test_gpu_every_colour_every_variant(pkg, tables, variant):
    import torch
    from dxt_lossless_transform_amd import color565 as mod

    src = as_bytes(all_colours())
    # host entry point
    out = np.zeros_like(src)
    mod.decorrelate_ycocg_r(src, out, variant)
    assert np.array_equal(out.view("<u2"), tables[(variant, False)])
    back = np.zeros_like(src)
    mod.recorrelate_ycocg_r(out, back, variant)
    assert np.array_equal(back, src)
    # device entry point, out of place then in place
    d = torch.from_numpy(src.copy()).cuda()
    o = torch.zeros_like(d)
    mod.decorrelate_ycocg_r(d, o, variant)
    assert np.array_equal(o.cpu().numpy().view("<u2"), tables[(variant, False)])
    mod.recorrelate_ycocg_r(o, o, variant)
    assert torch.equal(o, d)
    mod.recorrelate_ycocg_r(d, d, variant)
    assert np.array_equal(d.cpu().numpy().view("<u2"), tables[(variant, True)])


@pytest.mark.gpu
def test_gpu_known_answers_and_none(pkg):
    import torch
    from dxt_lossless_transform_amd import color565 as mod

    cols = np.array(list(YCOCG_KAT), dtype="<u2")
    for v in (1, 2, 3):
        out = np.zeros(cols.size * 2, dtype=np.uint8)
        mod.decorrelate_ycocg_r(as_bytes(cols), out, v)
        assert [int(x) for x in out.view("<u2")] == [YCOCG_KAT[int(c)][v - 1] for c in cols]
    d = torch.from_numpy(as_bytes(cols).copy()).cuda()
    o = torch.zeros_like(d)
    mod.decorrelate_ycocg_r(d, o, 0)
    assert torch.equal(d, o)
    mod.recorrelate_ycocg_r(d, d, 0)
    assert np.array_equal(d.cpu().numpy(), as_bytes(cols))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 63, 64, 65, 1000, 4099, 1 << 20])
@pytest.mark.parametrize("shift", [0, 2, 6])
def test_gpu_ragged_counts_and_misaligned_pointers(pkg, tables, n, shift):
    import torch
    from dxt_lossless_transform_amd import color565 as mod

    rng = np.random.default_rng(n * 7 + shift)
    cols = rng.integers(0, 65536, n, dtype=np.uint16).astype("<u2")
    variant = 1 + (n + shift) % 3
    big = torch.zeros(2 * n + 64, dtype=torch.uint8, device="cuda")
    big_out = torch.full_like(big, 0xA5)
    big[shift:shift + 2 * n] = torch.from_numpy(as_bytes(cols).copy()).cuda()
    out_shift = (shift * 3) % 8
    mod.decorrelate_ycocg_r(big[shift:shift + 2 * n], big_out[out_shift:out_shift + 2 * n], variant)
    got = big_out.cpu().numpy()
    assert np.array_equal(got[out_shift:out_shift + 2 * n].view("<u2"), tables[(variant, False)][cols])
    assert (got[:out_shift] == 0xA5).all() and (got[out_shift + 2 * n:] == 0xA5).all()   # nothing written outside
    mod.recorrelate_ycocg_r(big_out[out_shift:out_shift + 2 * n], big_out[out_shift:out_shift + 2 * n], variant)
    assert np.array_equal(big_out.cpu().numpy()[out_shift:out_shift + 2 * n], as_bytes(cols))


@pytest.mark.gpu
def test_gpu_split_reference_vector(pkg):
    """split_565_color_endpoints/mod.rs:138-165 (the 3-pair known-answer vector)."""
    from dxt_lossless_transform_amd import color565 as mod

    src = np.array([0x00, 0x01, 0x02, 0x03, 0x04, 0x05, 0x06, 0x07, 0x08, 0x09, 0x0A, 0x0B], dtype=np.uint8)
    out = np.zeros_like(src)
    mod.split_color_endpoints(src, out)
    assert out.tolist() == [0x00, 0x01, 0x04, 0x05, 0x08, 0x09, 0x02, 0x03, 0x06, 0x07, 0x0A, 0x0B]


@pytest.mark.gpu
@pytest.mark.parametrize("pairs", [1, 3, 4, 5, 64, 100, 1023, 1024, 4099, 1 << 19])
@pytest.mark.parametrize("shift", [0, 4])
def test_gpu_split_and_interleave_against_oracle(pkg, oracle, tables, pairs, shift):
    import torch
    from dxt_lossless_transform_amd import color565 as mod

    rng = np.random.default_rng(pairs + shift)
    src = rng.integers(0, 256, 4 * pairs, dtype=np.uint8)
    want = oracle.split_565_color_endpoints(src)
    # host entry point
    out = np.zeros_like(src)
    mod.split_color_endpoints(src, out)
    assert np.array_equal(out, want)
    # device entry point at a shifted address
    big = torch.zeros(4 * pairs + 32, dtype=torch.uint8, device="cuda")
    big[shift:shift + 4 * pairs] = torch.from_numpy(src).cuda()
    big_out = torch.full_like(big, 0x5A)
    mod.split_color_endpoints(big[shift:shift + 4 * pairs], big_out[shift:shift + 4 * pairs])
    got = big_out.cpu().numpy()
    assert np.array_equal(got[shift:shift + 4 * pairs], want)
    assert (got[:shift] == 0x5A).all() and (got[shift + 4 * pairs:] == 0x5A).all()
    # the inverse: interleave the two halves, recorrelating (variant None = plain interleave gives the source back)
    half = 2 * pairs
    for variant in (0, 1 + pairs % 3):
        d_out = torch.full_like(big, 0x33)
        halves = big_out[shift:shift + 4 * pairs]
        mod.recorrelate_ycocg_r_split(halves[:half], halves[half:], d_out[shift:shift + 4 * pairs], variant)
        got = d_out.cpu().numpy()
        cols = src.view("<u2")
        exp = cols if variant == 0 else tables[(variant, True)][cols]
        assert np.array_equal(got[shift:shift + 4 * pairs].view("<u2"), exp)
        assert (got[:shift] == 0x33).all() and (got[shift + 4 * pairs:] == 0x33).all()
        h_out = np.zeros_like(src)
        mod.recorrelate_ycocg_r_split(want[:half], want[half:], h_out, variant)
        assert np.array_equal(h_out.view("<u2"), exp)


@pytest.mark.gpu
def test_gpu_three_steps_equal_the_fused_transform(pkg, oracle):
    """The reference's experimental path builds a BC1 transform from these steps; the fused kernel must agree:
    colours of transform(split, variant) == decorrelate(split_color_endpoints(colour words))."""
    import torch
    from dxt_lossless_transform_amd import color565 as mod

    n = 4096 + 37
    blocks = oracle.generate_test_data("bc1", n)
    for variant in (1, 2, 3):
        fused = np.zeros_like(blocks)
        pkg.transform_bc1_with_settings(blocks, fused, pkg.Bc1TransformSettings(variant, True))
        colours = np.ascontiguousarray(blocks.reshape(n, 8)[:, :4]).reshape(-1)
        split = np.zeros_like(colours)
        mod.split_color_endpoints(colours, split)
        mod.decorrelate_ycocg_r(split, split, variant)
        assert np.array_equal(split, fused[:4 * n])
