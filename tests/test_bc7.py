"""BC7 mode-split transform, version 0 (docs/BC7_FORMAT.md) -- a format defined by this build; the reference has no BC7
transform, so parity here means: the two CPU statements agree, the GPU equals them bit for bit, and the
round trip is exact.  Nothing in this file is checked against reference behaviour."""
import numpy as np
import pytest

from oracle import oracle_np as onp

HEAD = [9, 9, 11, 11, 5, 7, 7, 11, 15]


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
    for n in (0, 1, 2, 17, 1023, 1024, 1025, 5000):
        x = make_blocks(oracle, n, kind, n)
        y = oracle.transform_bc7(x)
        assert y.size == x.size
        if n:
            assert np.array_equal(y, onp.transform_bc7(x))
            assert np.array_equal(onp.untransform_bc7(y), x)
        assert np.array_equal(oracle.transform_bc7(y, inverse=True), x)


def test_layout_by_hand(oracle):
    # two blocks: mode 6 (byte0 = 0x40) and reserved (byte0 = 0): first = [0x40, 0x00]; mode-6 head = bytes 1..7,
    # tail = bytes 8..15; reserved block: 15 head bytes, no tail
    x = np.zeros(32, dtype=np.uint8)
    x[0] = 0x40
    x[1:16] = np.arange(1, 16)
    x[17:32] = np.arange(101, 116)
    want = bytes([0x40, 0x00]) + bytes(range(1, 8)) + bytes(range(8, 16)) + bytes(range(101, 116))
    assert oracle.transform_bc7(x).tobytes() == want
    # stream sizes: first N, then per mode count*H and count*(15-H)
    y = make_blocks(oracle, 4000, "uniform")
    modes = onp.bc7_modes(y.reshape(-1, 16)[:, 0])
    assert sum(int((modes == m).sum()) * 15 for m in range(9)) + 4000 == y.size


def test_real_bc7_texture(oracle):
    """The reference's BC7 test texture (tests/golden/r2-256-bc7.payload.bin): every mode occurs, the transform round
    trips, and the transformed stream is no worse for a generic compressor (measured: zlib-6 36 778 -> 35 934 bytes,
    lzma 34 492 -> 34 164; the BC1-3 transforms gain more -- BC7 fields are not byte aligned and v0 only regroups
    bytes)."""
    import hashlib
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
def test_gpu_real_bc7_texture(bc7, oracle):
    import os

    p = np.fromfile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "r2-256-bc7.payload.bin"),
                    dtype=np.uint8)
    y, z = gpu_fwd_inv(bc7, p, torch.device("cuda:0"))
    assert np.array_equal(y, oracle.transform_bc7(p)) and np.array_equal(z, p)


@pytest.mark.gpu
def test_host_pointer_entry_points(pkg, bc7, oracle):
    for n in (0, 1, 1500, 70_001):
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
def test_four_gib_mode_mixed(pkg, bc7, oracle):
    """BASELINE.json configs[3]: 4 GiB synthetic mode-mixed buffer.  Exact round trip; the `first` stream and the
    per-mode stream sizes checked against counts computed independently with torch; a 64 MiB prefix of the buffer
    transformed on its own must equal the oracle (whole-buffer comparison is done at 256 MiB)."""
    dev = torch.device("cuda:0")
    n = (4 << 30) // 16
    x = torch.empty(n * 16, dtype=torch.uint8, device=dev)
    pkg.fill_splitmix64(x, 0x0BC70004)
    blocks = x.view(-1, 16)
    m = (blocks[:, 15] & 7).to(torch.int32)
    low_mask = ((2 << m) - 1).to(torch.uint8)
    blocks[:, 0] = (blocks[:, 0] & ~low_mask) | (1 << m).to(torch.uint8)   # same rule as oracle_bc7_force_modes
    y = torch.empty_like(x)
    z = torch.empty_like(x)
    ws = torch.empty(bc7.workspace_bytes(x.numel()), dtype=torch.uint8, device=dev)
    bc7.transform_bc7(x, y, ws)
    bc7.untransform_bc7(y, z, ws)
    torch.cuda.synchronize()
    assert torch.equal(z, x)
    assert torch.equal(y[:n], blocks[:, 0].contiguous())
    counts = torch.bincount(m, minlength=8).tolist()
    # mode-0 head stream = bytes 1..9 of the mode-0 blocks in order: check its first and last record
    sel0 = torch.nonzero(m == 0).flatten()
    first0, last0 = int(sel0[0]), int(sel0[-1])
    assert torch.equal(y[n: n + 9], blocks[first0, 1:10])
    assert torch.equal(y[n + (counts[0] - 1) * 9: n + counts[0] * 9], blocks[last0, 1:10])
    # 256 MiB against the oracle, whole buffer
    k = (256 << 20)
    xs = x[:k].contiguous()
    ys = torch.empty_like(xs)
    bc7.transform_bc7(xs, ys)
    torch.cuda.synchronize()
    assert np.array_equal(ys.cpu().numpy(), oracle.transform_bc7(xs.cpu().numpy()))
