"""dxtlt_transform_batch_device: many device-resident buffers in one call over a pool of internal streams (additive to
the reference's one-buffer-per-call API).  Results must equal the oracle item by item and the call must stay ordered
with the caller's stream on both sides."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
FORMATS = ("bc1", "bc2", "bc3")


def settings_for(pkg, fmt, v, sa, sc):
    if fmt == "bc3":
        return pkg.Bc3TransformSettings(pkg.YCoCgVariant(v), bool(sa), bool(sc))
    cls = pkg.Bc1TransformSettings if fmt == "bc1" else pkg.Bc2TransformSettings
    return cls(pkg.YCoCgVariant(v), bool(sc))


@pytest.mark.gpu
def test_batch_equals_oracle_item_by_item(pkg, oracle):
    from dxt_lossless_transform_amd import batch

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0xBA7C)
    items, expect = [], []
    for k in range(150):
        fmt = FORMATS[k % 3]
        blocks = int(rng.choice([0, 1, 3, 255, 256, 1025, 5463, 21845, 65536, 131073]))   # mip-chain-like odd counts too
        v, sa, sc = int(rng.integers(0, 4)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
        x = oracle.fill_splitmix64(blocks * pkg.BLOCK_BYTES[fmt], 0xBA7C + k)
        inverse = bool(k % 2)
        want = oracle.transform(fmt, x, v, bool(sc), bool(sa), inverse=inverse)
        xd = torch.from_numpy(x).to(dev)
        yd = torch.full((x.size + 32,), 0x5A, dtype=torch.uint8, device=dev)
        items.append((fmt, inverse, xd, yd[: x.size], settings_for(pkg, fmt, v, sa, sc)))
        expect.append((want, yd, x.size))
    batch.transform_batch(items)
    torch.cuda.synchronize()
    for k, (want, yd, n) in enumerate(expect):
        got = yd.cpu().numpy()
        assert np.array_equal(got[:n], want), k
        assert (got[n:] == 0x5A).all(), k
    batch.transform_batch([])   # empty batch is a no-op


@pytest.mark.gpu
def test_batch_index_with_hundreds_of_tiny_buffers_between_large_ones(pkg, oracle):
    """The workgroup -> buffer index is one byte per 64 workgroups on top of a base per 4096 (bcn_launch.h), 16 bits per 64 when
    more than 255 buffers begin inside one 4096-workgroup span; the kernel finds the owner among the entries that begin inside a
    64-workgroup group by bisection.  A large buffer, 700 buffers of one to three tiles of different sizes (no two neighbours alike: no equal-size
    shortcut), another large one, 300 more tiny ones: every buffer against the oracle, guard bytes intact, both directions."""
    from dxt_lossless_transform_amd import batch

    dev = torch.device("cuda:0")
    fmt, B = "bc3", 16
    st = settings_for(pkg, fmt, 1, 1, 1)
    rng = np.random.default_rng(0x1DE7)
    counts = [300_000] + [int(rng.integers(1, 770)) for _ in range(700)] + [1_398_103] + [int(rng.integers(1, 300)) for _ in range(300)]
    for inverse in (False, True):
        xs = [oracle.fill_splitmix64(n * B, 0x1DE7 + 7 * k) for k, n in enumerate(counts)]
        if inverse:
            xs = [oracle.transform(fmt, x, 1, True, True) for x in xs]
        offs, at = [], 0
        for n in counts:
            offs.append(at)
            at += (n * B + 64 + 255) // 256 * 256
        xd = torch.zeros(at, dtype=torch.uint8, device=dev)
        yd = torch.full((at,), 0x5A, dtype=torch.uint8, device=dev)
        hx = np.zeros(at, dtype=np.uint8)
        for x, o in zip(xs, offs):
            hx[o:o + x.size] = x
        xd.copy_(torch.from_numpy(hx))
        batch.transform_batch([(fmt, inverse, xd[o:o + n * B], yd[o:o + n * B], st) for n, o in zip(counts, offs)])
        torch.cuda.synchronize()
        got = yd.cpu().numpy()
        for k, (n, o, x) in enumerate(zip(counts, offs, xs)):
            want = oracle.transform(fmt, x, 1, True, True, inverse=inverse)
            assert np.array_equal(got[o:o + n * B], want), (inverse, k, n)
            assert (got[o + n * B:o + n * B + 64] == 0x5A).all(), (inverse, k, n)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["one_workgroup_each", "one_to_three_tiles"])
def test_batch_of_tens_of_thousands_of_tiny_buffers(pkg, oracle, shape):
    """Thumbnails and the last levels of mip chains: 20 000 BC1 buffers in one call.  `one_workgroup_each`: every buffer smaller
    than a tile -- all own ONE workgroup, the kernel's entry is its workgroup number (uniform lookup, no index).
    `one_to_three_tiles`: different sizes, 60+ buffers begin inside every 64-workgroup group and thousands inside every
    4096-workgroup span -- the index takes its wide (16-bit) form and the owner is found by bisection (round 4: a byte that
    saturated and a walk of up to ~3800 dependent entry loads per workgroup).  Every buffer against the oracle, both
    directions, guard bytes intact."""
    from dxt_lossless_transform_amd import batch

    dev = torch.device("cuda:0")
    fmt, B = "bc1", 8
    st = settings_for(pkg, fmt, 1, 0, 1)
    rng = np.random.default_rng(0x71A7 + len(shape))
    hi = 512 if shape == "one_workgroup_each" else 1500    # BC1 tiles hold 512 blocks
    counts = [int(x) for x in rng.integers(1, hi, 20000)]
    offs, at = [], 0
    for n in counts:
        offs.append(at)
        at += (n * B + 64 + 255) // 256 * 256
    src = oracle.fill_splitmix64(at, 0x71A7)
    for inverse in (False, True):
        hx = np.zeros(at, dtype=np.uint8)
        for n, o in zip(counts, offs):
            x = src[o:o + n * B]
            hx[o:o + n * B] = oracle.transform(fmt, x, 1, True, False) if inverse else x
        xd = torch.from_numpy(hx).to(dev)
        yd = torch.full((at,), 0x5A, dtype=torch.uint8, device=dev)
        batch.transform_batch([(fmt, inverse, xd[o:o + n * B], yd[o:o + n * B], st) for n, o in zip(counts, offs)])
        torch.cuda.synchronize()
        got = yd.cpu().numpy()
        for k, (n, o) in enumerate(zip(counts, offs)):
            want = src[o:o + n * B] if inverse else oracle.transform(fmt, src[o:o + n * B], 1, True, False)
            assert np.array_equal(got[o:o + n * B], want), (shape, inverse, k, n)
            assert (got[o + n * B:o + n * B + 64] == 0x5A).all(), (shape, inverse, k, n)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", FORMATS)
def test_batch_runs_every_tile_form(pkg, oracle, fmt):
    """The batch kernel picks a tile form per buffer (plan_batch_entry): aligned tiles when every stream base sits on a
    128-byte line, forward halo tiles when not and the output pointer is 8-byte aligned, the first shifted form for the
    inverse and for outputs at other addresses.  One batch holds all of them, in both directions, with block counts that
    put the element workgroups of the halo form (the first 64 blocks, the last 64 of the tiles, the rest) to work."""
    from dxt_lossless_transform_amd import batch

    dev = torch.device("cuda:0")
    B = pkg.BLOCK_BYTES[fmt]
    tile = 4096 // B
    st = settings_for(pkg, fmt, 1, 1, 1)
    items, expect = [], []
    cases = [(64 * tile, 0, 0), (64 * tile + 1, 0, 0), (64 * tile - 1, 0, 0), (3 * tile + 65, 0, 8), (3 * tile + 65, 0, 4),
             (3 * tile + 65, 0, 1), (tile + 63, 16, 0), (tile, 0, 24), (2 * tile - 1, 8, 8), (tile - 1, 0, 0), (65, 0, 8)]
    for k, (blocks, in_off, out_off) in enumerate(cases):
        for inverse in (False, True):
            x = oracle.fill_splitmix64(blocks * B, 0xF0 + k)
            if inverse:
                x = oracle.transform(fmt, x, 1, True, True)
            want = oracle.transform(fmt, x, 1, True, True, inverse=inverse)
            xd = torch.zeros(x.size + 64, dtype=torch.uint8, device=dev)
            xd[in_off:in_off + x.size] = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
            yd = torch.full((x.size + 96,), 0x5A, dtype=torch.uint8, device=dev)
            items.append((fmt, inverse, xd[in_off:in_off + x.size], yd[out_off:out_off + x.size], st))
            expect.append((want, yd, out_off, x.size))
    batch.transform_batch(items)
    torch.cuda.synchronize()
    for k, (want, yd, off, n) in enumerate(expect):
        got = yd.cpu().numpy()
        assert np.array_equal(got[off:off + n], want), (fmt, cases[k // 2], k % 2)
        assert (got[:off] == 0x5A).all() and (got[off + n:] == 0x5A).all(), (fmt, cases[k // 2], k % 2)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", FORMATS)
def test_batch_of_equal_buffers_every_lookup(pkg, oracle, fmt):
    """Batches of equal-size buffers take shortcuts in the kernel's buffer lookup: a division instead of the table walk, and
    for a regular array (one stride between the sources, one between the destinations) no table at all.  Every layout --
    back to back, padded stride, reverse order (negative stride), scattered -- and every tile form (block counts on and off
    the 128-byte lines), both directions, against the oracle; guard bytes between the outputs stay untouched."""
    from dxt_lossless_transform_amd import batch

    dev = torch.device("cuda:0")
    B = pkg.BLOCK_BYTES[fmt]
    tile = 4096 // B
    st = settings_for(pkg, fmt, 2, 1, 0)
    count = 19
    for blocks in (8 * tile, 8 * tile + 1, 3 * tile + 65, tile - 1):
        n = blocks * B
        for layout, pad in (("back_to_back", 0), ("padded", 4352), ("reverse", 128), ("scattered", 640)):
            for inverse in (False, True):
                xs = [oracle.fill_splitmix64(n, 0xE9 + 31 * k + blocks) for k in range(count)]
                if inverse:
                    xs = [oracle.transform(fmt, x, 2, False, True) for x in xs]
                want = [oracle.transform(fmt, x, 2, False, True, inverse=inverse) for x in xs]
                stride = n + pad
                order = list(range(count))
                if layout == "reverse":
                    order = order[::-1]
                if layout == "scattered":
                    order = [(7 * k + 3) % count for k in range(count)]
                xd = torch.zeros(count * stride + 64, dtype=torch.uint8, device=dev)
                yd = torch.full((count * stride + 64,), 0x5A, dtype=torch.uint8, device=dev)
                items = []
                for k in range(count):
                    lo = order[k] * stride
                    xd[lo:lo + n] = torch.from_numpy(np.ascontiguousarray(xs[k])).to(dev)
                    items.append((fmt, inverse, xd[lo:lo + n], yd[lo:lo + n], st))
                batch.transform_batch(items)
                torch.cuda.synchronize()
                got = yd.cpu().numpy()
                for k in range(count):
                    lo = order[k] * stride
                    assert np.array_equal(got[lo:lo + n], want[k]), (fmt, blocks, layout, inverse, k)
                    assert (got[lo + n:lo + stride] == 0x5A).all(), (fmt, blocks, layout, inverse, k)


@pytest.mark.gpu
def test_batch_is_ordered_with_the_callers_stream(pkg, oracle):
    """The batch reads what earlier work on the stream produced and later work on the stream sees its output, without
    any synchronisation by the caller."""
    from dxt_lossless_transform_amd import batch

    dev = torch.device("cuda:0")
    n = 1 << 20
    st = pkg.Bc1TransformSettings()
    for _ in range(5):
        xs = [torch.empty(n * 8, dtype=torch.uint8, device=dev) for _ in range(24)]
        ys = [torch.empty_like(x) for x in xs]
        zs = [torch.empty_like(x) for x in xs]
        for k, x in enumerate(xs):
            pkg.fill_splitmix64(x, 0x0D0E + k)                               # producer on the current stream
        batch.transform_batch([("bc1", False, x, y, st) for x, y in zip(xs, ys)])
        batch.transform_batch([("bc1", True, y, z, st) for y, z in zip(ys, zs)])   # consumes the first batch's output
        assert all(torch.equal(z, x) for x, z in zip(xs, zs))
    want = oracle.transform("bc1", xs[3][: 8 * 4096].cpu().numpy(), 1, True)
    head = torch.cat([ys[3][: 2 * 4096], ys[3][2 * n: 2 * n + 2 * 4096], ys[3][4 * n: 4 * n + 4 * 4096]]).cpu().numpy()
    assert np.array_equal(head, want)


@pytest.mark.gpu
def test_batch_validation_is_all_or_nothing(pkg):
    from dxt_lossless_transform_amd.batch import DxtltBatchItem

    l = pkg.load()
    l.dxtlt_transform_batch_device.argtypes = [C.POINTER(DxtltBatchItem), C.c_size_t, C.c_void_p]
    l.dxtlt_transform_batch_device.restype = C.c_int32
    dev = torch.device("cuda:0")
    x = torch.zeros(64, dtype=torch.uint8, device=dev)
    y = torch.full((64,), 7, dtype=torch.uint8, device=dev)
    arr = (DxtltBatchItem * 2)()
    for it in arr:
        it.d_input, it.d_output, it.len, it.format, it.decorrelation_mode = x.data_ptr(), y.data_ptr(), 64, 1, 1
    arr[1].len = 12                                        # second item invalid: nothing may be enqueued
    assert l.dxtlt_transform_batch_device(arr, 2, None) == 1
    torch.cuda.synchronize()
    assert bool((y == 7).all())
    arr[1].len, arr[1].format = 64, 9
    assert l.dxtlt_transform_batch_device(arr, 2, None) == 2
    assert l.dxtlt_transform_batch_device(None, 0, None) == 0
    assert l.dxtlt_transform_batch_device(None, 1, None) == 2


@pytest.mark.gpu
def test_host_batch_equals_oracle_item_by_item(pkg, oracle):
    """dxtlt_transform_batch_host: the reference's call pattern (many small HOST buffers) in one call -- mixed formats,
    directions, settings and sizes, several 64 MiB chunks, guard bytes behind every output."""
    from dxt_lossless_transform_amd import batch

    rng = np.random.default_rng(0x4057)
    items, expect = [], []
    sizes = [0, 1, 3, 255, 256, 1025, 5463, 21845, 65536, 131073, 700_001]
    for k in range(260):
        fmt = FORMATS[k % 3]
        blocks = int(rng.choice(sizes))
        v, sa, sc = int(rng.integers(0, 4)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
        x = oracle.fill_splitmix64(blocks * pkg.BLOCK_BYTES[fmt], 0x4057 + k)
        inverse = bool(k % 2)
        want = oracle.transform(fmt, x, v, bool(sc), bool(sa), inverse=inverse)
        y = np.full(x.size + 32, 0x5A, dtype=np.uint8)
        items.append((fmt, inverse, x, y[: x.size], settings_for(pkg, fmt, v, sa, sc)))
        expect.append((want, y, x.size))
    assert sum(n for _, _, n in expect) > 3 * (64 << 20)      # more than three chunks
    batch.transform_batch_host(items)
    for k, (want, y, n) in enumerate(expect):
        assert np.array_equal(y[:n], want), k
        assert (y[n:] == 0x5A).all(), k
    batch.transform_batch_host([])
    with pytest.raises(pkg.InvalidLength):
        batch.transform_batch_host([("bc1", False, np.zeros(12, np.uint8), np.zeros(12, np.uint8), pkg.Bc1TransformSettings())])
    # a failing item leaves the library usable: the next call works
    x = oracle.fill_splitmix64(4096 * 8, 7)
    y = np.zeros_like(x)
    batch.transform_batch_host([("bc1", False, x, y, pkg.Bc1TransformSettings())])
    assert np.array_equal(y, oracle.transform("bc1", x, 1, True))


@pytest.mark.gpu
def test_host_batch_with_items_larger_than_a_chunk(pkg, oracle):
    """Arena sizes stay bounded whatever the item sizes: an item of two chunks or more (here 136 MiB of BC1
    and 130 MiB of BC7) goes through the single-buffer pipeline, a chunk is closed BEFORE the item that would overfill it
    (40 MiB items: one per chunk once a 30 MiB item sits in it), and the small items around them still batch."""
    from dxt_lossless_transform_amd import batch
    from tests.test_bc7 import make_blocks

    plan = [("bc1", 1 << 20), ("bc3", 30 << 20), ("bc1", 40 << 20), ("bc2", 40 << 20), ("bc1", 136 << 20), ("bc3", 4096 * 16 + 16),
            ("bc7", 130 << 20), ("bc7", 5 * 1024 * 16 + 48), ("bc1", 8), ("bc3", 20 << 20)]
    items, expect = [], []
    for k, (fmt, nbytes) in enumerate(plan):
        inverse = k % 3 == 2
        if fmt == "bc7":
            x = make_blocks(oracle, nbytes // 16, "uniform", k)
            want = oracle.transform_bc7(x, inverse=inverse)
            st = None
        else:
            x = oracle.fill_splitmix64(nbytes, 0xB16 + k)
            want = oracle.transform(fmt, x, 1, True, True, inverse=inverse)
            st = settings_for(pkg, fmt, 1, 1, 1)
        y = np.full(x.size + 32, 0x5A, dtype=np.uint8)
        items.append((fmt, inverse, x, y[: x.size], st))
        expect.append((want, y, x.size))
    batch.transform_batch_host(items)
    for k, (want, y, n) in enumerate(expect):
        assert np.array_equal(y[:n], want), (k, plan[k])
        assert (y[n:] == 0x5A).all(), k


@pytest.mark.gpu
def test_concurrent_host_batches_and_parallel_auto_transforms(pkg, oracle):
    """Four threads at once: two host batches (each with its own pinned arenas, copy threads and streams), one auto
    transform with the estimator on four threads, one plain host call -- every result equal to the oracle's."""
    import threading

    from dxt_lossless_transform_amd import batch
    from oracle import oracle_auto
    import cabi

    lib = cabi.bind(C.CDLL(pkg._lib.lib_path()))
    errors = []

    def guarded(fn):
        def run():
            try:
                fn()
            except BaseException as e:  # noqa: BLE001
                errors.append(repr(e))
        return run

    def host_batch(seed):
        rng = np.random.default_rng(seed)
        items, expect = [], []
        for k in range(120):
            fmt = FORMATS[k % 3]
            blocks = int(rng.integers(0, 40_000))
            v, sa, sc = int(rng.integers(0, 4)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
            x = oracle.fill_splitmix64(blocks * pkg.BLOCK_BYTES[fmt], seed + k)
            y = np.zeros_like(x)
            items.append((fmt, False, x, y, settings_for(pkg, fmt, v, sa, sc)))
            expect.append((oracle.transform(fmt, x, v, bool(sc), bool(sa)), y))
        for _ in range(3):
            batch.transform_batch_host(items)
        for k, (want, y) in enumerate(expect):
            assert np.array_equal(y, want), (seed, k)

    def auto_parallel():
        x = np.tile(np.fromfile(__import__("os").path.join(__import__("helpers").GOLDEN, "r2-256-bc3.payload.bin"), dtype=np.uint8), 8)
        est, py_est = cabi.make_estimator("zlib")
        want_choice, want_out, _ = oracle_auto.transform_auto("bc3", x, lambda b: py_est(bytes(b)), True)
        out = cabi.CoreSettings3()
        for _ in range(3):
            y = np.zeros_like(x)
            r = lib.dltbc3core_transform_auto(x.ctypes.data, x.size, y.ctypes.data, y.size, C.byref(est), cabi.AutoSettings(True), C.byref(out))
            assert r.ErrorCode == 0 and np.array_equal(y, want_out)

    def plain_host_calls():
        x = oracle.fill_splitmix64(8 * 300_001, 99)
        want = oracle.transform("bc1", x, 1, True)
        for _ in range(6):
            y = np.zeros_like(x)
            pkg.transform_bc1_with_settings(x, y)
            assert np.array_equal(y, want)

    pkg.set_auto_estimator_threads(4)
    try:
        threads = [threading.Thread(target=guarded(f)) for f in (lambda: host_batch(0xBA7C0), lambda: host_batch(0xBA7C1), auto_parallel,
                                                                 plain_host_calls)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        pkg.set_auto_estimator_threads(1)
    assert not errors, errors


@pytest.mark.gpu
def test_bc7_items_ride_along_in_both_batch_calls(pkg, oracle):
    """format 7 in dxtlt_transform_batch_device and dxtlt_transform_batch_host: BC7 buffers of every size class (empty,
    tail part only, whole granules, both) next to BC1-3 items, forward and inverse, against the oracle."""
    from dxt_lossless_transform_amd import batch

    rng = np.random.default_rng(0xB7BA)
    dev = torch.device("cuda:0")
    sizes7 = [0, 1, 500, 1023, 1024, 1025, 2048, 3 * 1024 + 77, 40 * 1024, 40 * 1024 + 1, 200_003]
    host_items, dev_items, expect = [], [], []
    for k in range(60):
        if k % 3 == 0:
            fmt, blocks = FORMATS[(k // 3) % 3], int(rng.integers(0, 30_000))
            v, sa, sc = int(rng.integers(0, 4)), int(rng.integers(0, 2)), int(rng.integers(0, 2))
            x = oracle.fill_splitmix64(blocks * pkg.BLOCK_BYTES[fmt], 0xB7BA + k)
            inverse = bool(k % 2)
            want = oracle.transform(fmt, x, v, bool(sc), bool(sa), inverse=inverse)
            settings = settings_for(pkg, fmt, v, sa, sc)
        else:
            fmt, blocks = "bc7", int(sizes7[k % len(sizes7)])
            x = rng.integers(0, 256, 16 * blocks, dtype=np.uint8)      # raw bytes: every class, the reserved one included
            inverse = bool((k // 3) % 2)
            if inverse:
                x = oracle.transform_bc7(x) if blocks else x
            want = oracle.transform_bc7(x, inverse=inverse) if blocks else x
            settings = None
        y = np.full(x.size + 16, 0x5A, dtype=np.uint8)
        host_items.append((fmt, inverse, x, y[: x.size], settings))
        xd = torch.from_numpy(x.copy()).to(dev)
        yd = torch.full((x.size + 16,), 0x5A, dtype=torch.uint8, device=dev)
        dev_items.append((fmt, inverse, xd, yd[: x.size], settings))
        expect.append((want, y, yd, x.size))
    batch.transform_batch_host(host_items)
    prepared = batch.prepare_batch(dev_items)            # the reusable form of transform_batch
    batch.run_prepared_batch(prepared)
    torch.cuda.synchronize()
    for k, (want, y, yd, n) in enumerate(expect):
        assert np.array_equal(y[:n], want) and (y[n:] == 0x5A).all(), ("host", k, host_items[k][0], n)
        h = yd.cpu().numpy()
        assert np.array_equal(h[:n], want) and (h[n:] == 0x5A).all(), ("device", k, dev_items[k][0], n)
    with pytest.raises(pkg.InvalidLength):
        batch.transform_batch_host([("bc7", False, np.zeros(24, np.uint8), np.zeros(24, np.uint8), None)])


@pytest.mark.gpu
def test_many_mixed_batch_calls_in_flight_share_the_table_ring(pkg, oracle):
    """Twelve device batch calls back to back on one stream without a synchronisation in between, every one holding BC7 forward, BC7
    inverse and BC1-3 items (round 5: three table slots per call, so the second such call waited for the first one's kernels; now one
    slot per call, and the ring of four is reused three times over here) -- and BC1 without the colour split, forward, at odd and even
    counts (the batch launch with 128-lane tiles).  Every output against the oracle, guard bytes behind it."""
    from dxt_lossless_transform_amd import batch

    rng = np.random.default_rng(0x51075)
    dev = torch.device("cuda:0")
    calls, expect = [], []
    for c in range(12):
        items = []
        for k in range(10):
            if k < 2:
                fmt, blocks, inverse, settings = "bc7", [3 * 1024 + 5, 1024, 777, 2048 + 1][(c + k) % 4], bool(k), None
                x = rng.integers(0, 256, 16 * blocks, dtype=np.uint8)
                if inverse:
                    x = oracle.transform_bc7(x)
                want = oracle.transform_bc7(x, inverse=inverse)
            else:
                fmt = FORMATS[k % 3]
                blocks = int(rng.integers(1, 40_000)) if k % 2 else 512 * int(rng.integers(1, 60))
                v, sa, sc = (1, 0, 0) if (fmt == "bc1" and k < 8) else (int(rng.integers(0, 4)), int(rng.integers(0, 2)), int(rng.integers(0, 2)))
                inverse = (k == 9)
                x = oracle.fill_splitmix64(blocks * pkg.BLOCK_BYTES[fmt], 0x510 + 16 * c + k)
                want = oracle.transform(fmt, x, v, bool(sc), bool(sa), inverse=inverse)
                settings = settings_for(pkg, fmt, v, sa, sc)
            xd = torch.from_numpy(x.copy()).to(dev)
            yd = torch.full((x.size + 16,), 0x5A, dtype=torch.uint8, device=dev)
            items.append((fmt, inverse, xd, yd[: x.size], settings))
            expect.append((c, k, fmt, want, yd, x.size))
        calls.append(batch.prepare_batch(items))
    torch.cuda.synchronize()
    for prepared in calls:            # nothing between the calls: each finds the previous ones' tables still in use
        batch.run_prepared_batch(prepared)
    torch.cuda.synchronize()
    for c, k, fmt, want, yd, n in expect:
        h = yd.cpu().numpy()
        assert np.array_equal(h[:n], want) and (h[n:] == 0x5A).all(), (c, k, fmt, n)


_SPAWN_FAILURE_SCRIPT = r'''
import os, resource, sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
import dxt_lossless_transform_amd as pkg
from dxt_lossless_transform_amd import batch, bc7

st = pkg.Bc1TransformSettings()
n = 16 << 20
src = [np.random.default_rng(i).integers(0, 256, n, dtype=np.uint8) for i in range(9)]
dst = [np.empty_like(a) for a in src]
items = [("bc1", False, a, b, st) for a, b in zip(src, dst)]
big_in = np.concatenate(src[:4]); big_out = np.empty_like(big_in)

def calls():
    out = {}
    for name, fn in (("batch_host", lambda: batch.transform_batch_host(items)),
                     ("sharded", lambda: pkg.transform_sharded("bc1", False, big_in, big_out, st, 2)),
                     ("bc7_sharded", lambda: bc7.transform_bc7_sharded(big_in, big_out, 2))):
        t0 = time.perf_counter()
        try:
            fn(); out[name] = 0
        except pkg.DeviceError as e:
            out[name] = e.code
        out[name + "_s"] = time.perf_counter() - t0
    return out

first = calls()                       # every lazily started runtime thread exists after this
soft, hard = resource.getrlimit(resource.RLIMIT_NPROC)
resource.setrlimit(resource.RLIMIT_NPROC, (1, hard))      # any further thread creation of this user fails with EAGAIN
starved = calls()
resource.setrlimit(resource.RLIMIT_NPROC, (soft, hard))
after = calls()
print(repr((first, starved, after)))
'''


@pytest.mark.gpu
def test_thread_spawn_failure_is_one_clean_status_everywhere(pkg):
    """The library starts host threads in dxtlt_transform_batch_host, dxtlt_transform_sharded and the BC7 sharded calls.  When
    the process may not create threads (RLIMIT_NPROC), each of them must come back promptly with DXTLT_E_ALLOCATION (6) --
    no hang, no partial work left running -- and work again once the limit is lifted.  (Root is exempt from RLIMIT_NPROC.)"""
    import ast
    import os
    import subprocess
    import sys

    if os.geteuid() == 0:
        pytest.skip("RLIMIT_NPROC does not bind root")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _SPAWN_FAILURE_SCRIPT, root], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    first, starved, after = ast.literal_eval(r.stdout.strip().splitlines()[-1])
    for name in ("batch_host", "sharded", "bc7_sharded"):
        assert first[name] == 0 and after[name] == 0, (name, first, after)
        assert starved[name] == 6, (name, starved)          # DXTLT_E_ALLOCATION, the same code on every path
        assert starved[name + "_s"] < 30, (name, starved)
