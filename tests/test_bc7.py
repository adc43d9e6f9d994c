"""BC7 granule-sorted field split, version 2 (docs/BC7_FORMAT.md) -- a format defined by this build; the reference has
no BC7 transform, so parity here means: the two CPU statements agree, the device header's compile-time
field moves equal the oracle's, the GPU equals them bit for bit, and the round trip is exact.  Nothing in this file is
checked against reference behaviour except the mode bit fields (hexpat:286-654), which the statements restate."""
import numpy as np
import pytest

from oracle import oracle_np as onp



def make_blocks(oracle, n, kind, seed=1):
    x = oracle.fill_splitmix64(n * 16, 0x0BC70004 + seed)
    if kind == "uniform":          # modes 0..7 uniformly
        oracle.bc7_force_modes(x)
    elif kind == "mode6":          # a single mode
        x.reshape(-1, 16)[:, 0] = (x.reshape(-1, 16)[:, 0] & 0x80) | 0x40
    elif kind == "skewed":         # mode 6 > 1 > 3 > rest, plus some reserved blocks
        b = x.reshape(-1, 16)
        r = b[:, 15].astype(np.int64)
        mode = np.select([r < 140, r < 200, r < 230, r < 250], [6, 1, 3, 8], default=r % 8)
        keep = b[:, 0].astype(np.int64)
        marker = np.where(mode == 8, 0, 1 << np.minimum(mode, 7))
        b[:, 0] = np.where(mode == 8, 0, (keep & ~((2 << np.minimum(mode, 7)) - 1) & 0xFF) | marker).astype(np.uint8)
    elif kind == "raw":            # arbitrary bytes: mode from the data, byte 0 == 0 happens 1/256 of the time
        pass
    return x


# ---- CPU: the two statements of the format agree ----------------------------------------------------------
@pytest.mark.parametrize("kind", ["uniform", "mode6", "skewed", "raw"])
def test_c_and_numpy_statements_agree(oracle, kind):
    for n in (0, 1, 2, 17, 1023, 1024, 1025, 2100):
        x = make_blocks(oracle, n, kind, n)
        y = oracle.transform_bc7(x)
        assert y.size == x.size
        if n:
            assert np.array_equal(y, onp.transform_bc7(x))
            assert np.array_equal(onp.untransform_bc7(y), x)
        assert np.array_equal(oracle.transform_bc7(y, inverse=True), x)


def test_layout_by_hand(oracle):
    """Mode 6 by hand (hexpat:553-590): marker 0000001, R0 R1 G0 G1 B0 B1 A0 A1 (7 bits each), P0 P1, 63 index bits.
    Record: marker | P0 P1 | index bits | low 3 bits of the eight endpoints | high nibbles of the eight endpoints, the
    endpoints of the record being R - G, G, B - G modulo 128 (version 2's colour decorrelation) and A."""
    block_ep = [0x55, 0x2A, 0x7F, 0x00, 0x13, 0x64, 0x41, 0x3E]    # 7-bit values as the block holds them: R0 R1 G0 G1 B0 B1 A0 A1
    r0, r1, g0, g1, b0, b1, a0, a1 = block_ep
    ep = [(r0 - g0) & 127, (r1 - g1) & 127, g0, g1, (b0 - g0) & 127, (b1 - g1) & 127, a0, a1]   # as the record holds them
    assert ep == [0x56, 0x2A, 0x7F, 0x00, 0x14, 0x64, 0x41, 0x3E]
    p0, p1 = 1, 0
    idx = 0x5A5A_F0F0_1234_5678 & ((1 << 63) - 1)
    b, at = 1 << 6, 7
    for e in block_ep:
        b |= e << at
        at += 7
    b |= p0 << at | p1 << (at + 1)
    at += 2
    b |= idx << at
    assert at + 63 == 128
    r, at = 1 << 6, 7
    r |= p0 << at | p1 << (at + 1)
    at += 2
    r |= idx << at
    at += 63
    for e in ep:
        r |= (e & 7) << at
        at += 3
    for e in ep:
        r |= (e >> 3) << at
        at += 4
    assert at == 128
    blk = np.frombuffer(b.to_bytes(16, "little"), dtype=np.uint8)
    rec = np.frombuffer(r.to_bytes(16, "little"), dtype=np.uint8)
    assert np.array_equal(oracle.bc7_record(blk), rec)
    assert np.array_equal(oracle.bc7_record(rec, inverse=True), blk)
    assert onp.bc7_record_of_block(b, 6) == r and onp.bc7_block_of_record(r, 6) == b
    # the last four record bytes are the high nibbles, two endpoints per byte
    assert rec[12:].tolist() == [(ep[0] >> 3) | (ep[1] >> 3) << 4, (ep[2] >> 3) | (ep[3] >> 3) << 4,
                                 (ep[4] >> 3) | (ep[5] >> 3) << 4, (ep[6] >> 3) | (ep[7] >> 3) << 4]

    # streams: two blocks, mode 6 then the reserved class (byte 0 == 0, moved unchanged), n = 2 < granule: one tail part
    x = np.concatenate([blk, np.arange(100, 116, dtype=np.uint8)])
    x[16] = 0
    y = oracle.transform_bc7(x)
    n = 2
    recs = [rec, x[16:]]
    assert y[0:16].tobytes() == recs[0][1:9].tobytes() + recs[1][1:9].tobytes()           # Q8
    assert y[16:20].tobytes() == recs[0][9:11].tobytes() + recs[1][9:11].tobytes()        # Q2
    for k in range(5):                                                                     # B0..B4
        assert y[10 * n + k * n: 10 * n + (k + 1) * n].tolist() == [recs[0][11 + k], recs[1][11 + k]]
    assert y[15 * n:].tolist() == [rec[0], 0]                                              # F, block order

    # sorting: inside a granule blocks are ordered by class, stably; F stays in block order
    g = oracle.bc7_granule()
    assert g == onp.BC7_GRANULE == 1024
    x = make_blocks(oracle, g + 300, "uniform")
    y = oracle.transform_bc7(x)
    modes = onp.bc7_modes(x.reshape(-1, 16)[:g, 0])
    order = np.argsort(modes, kind="stable")
    recs = np.stack([oracle.bc7_record(x[16 * i:16 * i + 16]) for i in range(g)])
    assert np.array_equal(y[:8 * g].reshape(g, 8), recs[order][:, 1:9])          # main part: streams over g blocks
    assert np.array_equal(y[15 * g:16 * g], recs[:, 0])
    assert np.array_equal(y[16 * g:], oracle.transform_bc7(x[16 * g:]))          # tail part: a transform of its own


def test_device_header_equals_oracle(oracle, tmp_path):
    """csrc/bc7_fields.h (the compile-time bit-field moves of the kernels) built for the host against the oracle's
    one-field-at-a-time statement, every mode, and the forward kernel's shortcut for record byte 0."""
    import ctypes as C
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "bc7_fields_shim.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so,
                           os.path.join(root, "tests", "cpp", "bc7_fields_shim.cpp")])
    l = C.CDLL(so)
    l.shim_bc7_records.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    l.shim_bc7_record_byte0.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    n = 40_000
    for kind in ("uniform", "skewed", "raw"):
        x = make_blocks(oracle, n, kind, 5)
        rec = np.empty_like(x)
        l.shim_bc7_records(x.ctypes.data, rec.ctypes.data, n, 0)
        want = np.concatenate([oracle.bc7_record(x[16 * i:16 * i + 16]) for i in range(0, n, 7)])
        assert np.array_equal(rec.reshape(-1, 16)[::7].reshape(-1), want), kind
        back = np.empty_like(x)
        l.shim_bc7_records(rec.ctypes.data, back.ctypes.data, n, 1)
        assert np.array_equal(back, x), kind
        b0 = np.empty(n, dtype=np.uint8)
        l.shim_bc7_record_byte0(x.ctypes.data, b0.ctypes.data, n)
        assert np.array_equal(b0, rec[::16]), kind


def test_real_bc7_texture(oracle):
    """The reference's BC7 test texture (tests/golden/r2-256-bc7.payload.bin): every mode occurs, the transform round
    trips, and the transformed stream is no worse for a generic compressor (measured: zlib-6 36 778 -> 35 936 bytes,
    lzma 34 492 -> 34 100; docs/BC7_FORMAT.md has the table, including a 1 MiB photo-like texture where the gain is
    16-18 %)."""
    import hashlib
    import lzma
    import os
    import zlib

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "r2-256-bc7.payload.bin")
    p = np.fromfile(path, dtype=np.uint8)
    assert p.size == 65536 and hashlib.sha256(p.tobytes()).hexdigest().startswith("f3bdc75a0198826e")
    modes = onp.bc7_modes(p.reshape(-1, 16)[:, 0])
    assert all(int((modes == m).sum()) > 0 for m in range(8)) and int((modes == 8).sum()) == 0
    t = oracle.transform_bc7(p)
    assert np.array_equal(t, onp.transform_bc7(p))
    assert np.array_equal(oracle.transform_bc7(t, inverse=True), p)
    assert len(zlib.compress(t.tobytes(), 6)) < len(zlib.compress(p.tobytes(), 6))
    assert len(lzma.compress(t.tobytes(), preset=6)) < len(lzma.compress(p.tobytes(), preset=6))


# ---- GPU ---------------------------------------------------------------------------------------------------
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def bc7(pkg):
    from dxt_lossless_transform_amd import bc7 as mod

    return mod


def gpu_fwd_inv(bc7, x, dev):
    xd = torch.from_numpy(x).to(dev)
    yd = torch.full((x.size + 64,), 0x5A, dtype=torch.uint8, device=dev)
    bc7.transform_bc7(xd, yd[: x.size])
    zd = torch.full((x.size + 64,), 0x5A, dtype=torch.uint8, device=dev)
    bc7.untransform_bc7(yd[: x.size], zd[: x.size])
    torch.cuda.synchronize()
    assert bool((yd[x.size:] == 0x5A).all()) and bool((zd[x.size:] == 0x5A).all()), "wrote past the end"
    return yd[: x.size].cpu().numpy(), zd[: x.size].cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["uniform", "mode6", "skewed", "raw"])
def test_gpu_equals_oracle(pkg, bc7, oracle, kind):
    dev = torch.device("cuda:0")
    for n in (1, 2, 63, 64, 65, 255, 256, 1023, 1024, 1025, 2048, 4099, 100_003, 1024 * 1024 + 7):
        x = make_blocks(oracle, n, kind, n)
        y, z = gpu_fwd_inv(bc7, x, dev)
        assert np.array_equal(y, oracle.transform_bc7(x)), (kind, n)
        assert np.array_equal(z, x), (kind, n, "round trip")


@pytest.mark.gpu
@pytest.mark.parametrize("shift", [1, 4, 8, 20])
def test_gpu_misaligned_device_pointers(bc7, oracle, shift):
    """Both pointers off 16-byte alignment (a whole DDS file in HBM has its payload at byte 148): unaligned vector
    accesses, same bytes, nothing written before or behind the output."""
    dev = torch.device("cuda:0")
    for n in (5, 1024, 3 * 1024 + 77):
        x = make_blocks(oracle, n, "skewed", n + shift)
        xd = torch.zeros(x.size + shift, dtype=torch.uint8, device=dev)
        xd[shift:] = torch.from_numpy(x).to(dev)
        yd = torch.full((x.size + shift + 32,), 0x5A, dtype=torch.uint8, device=dev)
        bc7.transform_bc7(xd[shift:], yd[shift:shift + x.size])
        zd = torch.full((x.size + shift + 32,), 0x5A, dtype=torch.uint8, device=dev)
        bc7.untransform_bc7(yd[shift:shift + x.size], zd[shift:shift + x.size])
        torch.cuda.synchronize()
        assert np.array_equal(yd[shift:shift + x.size].cpu().numpy(), oracle.transform_bc7(x)), (n, shift)
        assert np.array_equal(zd[shift:shift + x.size].cpu().numpy(), x), (n, shift)
        for t in (yd, zd):
            assert bool((t[:shift] == 0x5A).all()) and bool((t[shift + x.size:] == 0x5A).all()), (n, shift)


@pytest.mark.gpu
def test_gpu_real_bc7_texture(bc7, oracle):
    import os

    p = np.fromfile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "r2-256-bc7.payload.bin"),
                    dtype=np.uint8)
    y, z = gpu_fwd_inv(bc7, p, torch.device("cuda:0"))
    assert np.array_equal(y, oracle.transform_bc7(p)) and np.array_equal(z, p)


@pytest.mark.gpu
def test_host_pointer_entry_points(pkg, bc7, oracle):
    # up to 1 MiB (65 536 blocks) through the mapped staging pair, above it by two copies
    for n in (0, 1, 1500, 65_535, 65_536, 65_537, 70_001):
        x = make_blocks(oracle, n, "skewed", n)
        y = np.zeros_like(x)
        bc7.transform_bc7(x, y)
        assert np.array_equal(y, oracle.transform_bc7(x))
        z = np.zeros_like(x)
        bc7.untransform_bc7(y, z)
        assert np.array_equal(z, x)
    with pytest.raises(pkg.InvalidLength):
        bc7.transform_bc7(np.zeros(24, dtype=np.uint8), np.zeros(24, dtype=np.uint8))
    with pytest.raises(pkg.OutputBufferTooSmall):
        bc7.transform_bc7(np.zeros(32, dtype=np.uint8), np.zeros(16, dtype=np.uint8))


@pytest.mark.gpu
def test_ranges_compose_and_reject_unaligned_starts(pkg, bc7, oracle):
    """dxtlt_transform_bc7_range_device: granule-aligned ranges written into one whole buffer equal the whole-buffer
    transform (no counters to exchange); a range that starts inside a granule is refused."""
    dev = torch.device("cuda:0")
    g = bc7.sort_granule()
    total = 9 * g + 417
    x = make_blocks(oracle, total, "skewed", 9)
    xd = torch.from_numpy(x).to(dev)
    yd = torch.zeros_like(xd)
    cuts = [0, 2 * g, 3 * g, 8 * g, total]
    for a, b in zip(cuts, cuts[1:]):
        bc7.transform_bc7_range(False, xd[16 * a:], yd, total, a, b - a)
    torch.cuda.synchronize()
    assert np.array_equal(yd.cpu().numpy(), oracle.transform_bc7(x))
    zd = torch.zeros_like(xd)
    for a, b in zip(cuts, cuts[1:]):
        bc7.transform_bc7_range(True, yd, zd[16 * a:], total, a, b - a)
    torch.cuda.synchronize()
    assert torch.equal(zd, xd)
    with pytest.raises(pkg.DeviceError):
        bc7.transform_bc7_range(False, xd[16 * 5:], yd, total, 5, g)
    with pytest.raises(pkg.DeviceError):
        bc7.transform_bc7_range(False, xd, yd, total, 0, g + 1)


@pytest.mark.gpu
def test_four_gib_mode_mixed(pkg, bc7, oracle):
    """BASELINE.json configs[3]: 4 GiB synthetic mode-mixed buffer.  Exact round trip; the F stream against record
    bytes computed with torch; sampled granules against the oracle (a granule's slices are a pure function of the
    granule's blocks); a 256 MiB prefix transformed on its own equals the oracle byte for byte."""
    dev = torch.device("cuda:0")
    n = (4 << 30) // 16
    x = torch.empty(n * 16, dtype=torch.uint8, device=dev)
    pkg.fill_splitmix64(x, 0x0BC70004)
    blocks = x.view(-1, 16)
    m = (blocks[:, 15] & 7).to(torch.int32)
    low_mask = ((2 << m) - 1).to(torch.uint8)
    blocks[:, 0] = (blocks[:, 0] & ~low_mask) | (1 << m).to(torch.uint8)   # same rule as oracle_bc7_force_modes
    del m, low_mask
    y = torch.empty_like(x)
    z = torch.empty_like(x)
    bc7.transform_bc7(x, y)
    bc7.untransform_bc7(y, z)
    torch.cuda.synchronize()
    assert torch.equal(z, x)
    del z
    g = bc7.sort_granule()
    offs, widths = [0, 8, 10, 11, 12, 13, 14, 15], [8, 2, 1, 1, 1, 1, 1, 1]
    for gi in (0, 1, 77_777, n // g // 2, n // g - 1):
        xin = x[16 * g * gi: 16 * g * (gi + 1)].cpu().numpy()
        want = oracle.transform_bc7(xin)                     # one granule on its own: streams over g blocks
        got = np.concatenate([y[o * n + w * g * gi: o * n + w * g * (gi + 1)].cpu().numpy() for o, w in zip(offs, widths)])
        assert np.array_equal(got, want), gi
    # F: byte 0 of the record = byte 0 of the block except for modes 0 and 6 (three / one p-bits move in)
    f = y[15 * n:]
    b0 = blocks[:, 0]
    other = ((b0 & 1) == 0) & ((b0 & 0x7F) != 0x40)
    assert torch.equal(f[other], b0[other])
    k = (256 << 20)
    xs = x[:k].contiguous()
    ys = torch.empty_like(xs)
    bc7.transform_bc7(xs, ys)
    torch.cuda.synchronize()
    assert np.array_equal(ys.cpu().numpy(), oracle.transform_bc7(xs.cpu().numpy()))


@pytest.mark.gpu
def test_large_host_buffers_take_the_chunked_pipeline(pkg, bc7, oracle):
    """>= 96 MiB through dxtlt_transform_bc7: the main part in chunks (upload | kernel | eight per-stream downloads
    overlapped), the tail part as a buffer of its own -- equal to the device path on the same data, exact round trip."""
    n = (130 << 20) // 16 + 777                       # main part of 8320 granules + a tail part
    x = make_blocks(oracle, n, "uniform", 0xB16)
    y, z = np.zeros_like(x), np.zeros_like(x)
    bc7.transform_bc7(x, y)                           # host pointers
    xd = torch.from_numpy(x).to("cuda:0")
    yd = torch.empty_like(xd)
    bc7.transform_bc7(xd, yd)                         # device pointers: one launch pair
    assert np.array_equal(y, yd.cpu().numpy())
    sample = slice(16 * 1024 * 4000, 16 * 1024 * 4003)  # three granules from the middle against the oracle's records
    want = oracle.transform_bc7(x[sample])
    main = n - n % 1024
    for off, w in ((0, 8), (8, 2), (10, 1), (11, 1), (12, 1), (13, 1), (14, 1), (15, 1)):
        got = y[off * main + w * 1024 * 4000: off * main + w * 1024 * 4003]
        assert np.array_equal(got, want[off * 3072: off * 3072 + w * 3072])
    bc7.untransform_bc7(y, z)
    assert np.array_equal(z, x)
