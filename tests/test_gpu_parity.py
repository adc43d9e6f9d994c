"""Parity of the HIP path (through the C ABI of libdxtlt_gfx950.so) against the CPU oracle.  Needs an MI355X.

Bar: bit-exact.  Small and medium sizes are compared byte-for-byte with the oracle and with the committed golden
fixtures; BASELINE.json's full sizes (8 GiB BC1 / BC3) use the size-independent properties of the domain: exact
round trip, exact comparison of sampled block windows (the transform is block-independent, so a window of the
output streams is a pure function of the same window of blocks), and per-stream 64-bit sums."""
import hashlib

import numpy as np
import pytest

from helpers import BLOCK, FORMATS, all_settings, golden_digests, golden_vectors, payload, pkg_settings, settings_id

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

TILE = {"bc1": 512, "bc2": 256, "bc3": 256}  # blocks per 256-thread tile (one 16-byte vector per lane)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return torch.device("cuda:0")


def run_device(pkg, fmt, x_host, s, dev, inverse=False):
    x = torch.from_numpy(np.ascontiguousarray(x_host)).to(dev)
    y = torch.full_like(x, 0xA5)
    name = ("untransform" if inverse else "transform") + f"_{fmt}_with_settings"
    getattr(pkg, name)(x, y, pkg_settings(pkg, fmt, s))
    torch.cuda.synchronize()
    return y.cpu().numpy()


def fwd_oracle(oracle, fmt, x, s, inverse=False):
    v, sa, sc = s
    return oracle.transform(fmt, x, v, sc, sa, inverse=inverse)


# ------------------------------------------------------------------------------------------------------------
# every settings combination x sizes around the tile boundaries (reference: run_*_roundtrip_test for every n
# in 1..=max_blocks with max_blocks = 2 x kernel width, bc1 test_prelude.rs:154-317)
# ------------------------------------------------------------------------------------------------------------
def sizes_for(fmt):
    t = TILE[fmt]
    return [1, 2, 3, 15, 16, 17, 63, 64, 65, t - 16, t - 1, t, t + 1, t + 16, 2 * t, 2 * t + 1, 3 * t + 16 * 7,
            5 * t + 37, 16 * t, 33 * t + 16]


@pytest.mark.parametrize("fmt", FORMATS)
def test_all_settings_all_sizes(pkg, oracle, dev, fmt):
    for n in sizes_for(fmt):
        x = oracle.fill_splitmix64(n * BLOCK[fmt], 0xC0FFEE00 + n)
        for s in all_settings(fmt):
            want = fwd_oracle(oracle, fmt, x, s)
            got = run_device(pkg, fmt, x, s, dev)
            assert np.array_equal(got, want), (fmt, n, settings_id(s), "forward")
            back = run_device(pkg, fmt, want, s, dev, inverse=True)
            assert np.array_equal(back, x), (fmt, n, settings_id(s), "inverse")


@pytest.mark.parametrize("fmt", FORMATS)
def test_every_n_up_to_two_tiles_default_settings(pkg, oracle, dev, fmt):
    """Every block count 0..=130 (head/exact/tail of the element kernel) plus every n in a window around two
    tiles; default settings."""
    s = {"bc1": (1, 0, 1), "bc2": (1, 0, 1), "bc3": (1, 1, 1)}[fmt]
    t = TILE[fmt]
    for n in list(range(0, 131)) + list(range(2 * t - 20, 2 * t + 21)):
        x = oracle.generate_test_data(fmt, n)
        got = run_device(pkg, fmt, x, s, dev) if n else np.zeros(0, dtype=np.uint8)
        assert np.array_equal(got, fwd_oracle(oracle, fmt, x, s)), (fmt, n)


@pytest.mark.parametrize("fmt", FORMATS)
def test_every_n_through_four_tiles_round_trip_and_oracle(pkg, oracle, dev, fmt):
    """The reference's harness runs transform -> untransform for EVERY block count 1..=max_blocks (bc1 test_prelude.rs:154-181);
    here every n from 1 to four tiles + 70 (so that every alignment class of the stream bases, every tail length of the
    element path and every head / tail of the halo ranges occurs), default settings and one non-default combination:
    forward == oracle, inverse(forward) == input."""
    t = TILE[fmt]
    combos = [{"bc1": (1, 0, 1), "bc2": (1, 0, 1), "bc3": (1, 1, 1)}[fmt], {"bc1": (3, 0, 0), "bc2": (2, 0, 0), "bc3": (2, 0, 1)}[fmt]]
    whole = oracle.fill_splitmix64((4 * t + 70) * BLOCK[fmt], 0xE7E12)
    for n in range(1, 4 * t + 71):
        x = whole[: n * BLOCK[fmt]]
        s = combos[n & 1]
        want = fwd_oracle(oracle, fmt, x, s)
        got = run_device(pkg, fmt, x, s, dev)
        assert np.array_equal(got, want), (fmt, n, "forward")
        back = run_device(pkg, fmt, want, s, dev, inverse=True)
        assert np.array_equal(back, x), (fmt, n, "inverse")


def test_zero_blocks_is_a_noop(pkg, dev):
    x = torch.empty(0, dtype=torch.uint8, device=dev)
    y = torch.empty(0, dtype=torch.uint8, device=dev)
    pkg.transform_bc1_with_settings(x, y)
    pkg.untransform_bc3_with_settings(x, y)
    pkg.transform_bc2_with_settings(np.zeros(0, dtype=np.uint8), np.zeros(0, dtype=np.uint8))


# ------------------------------------------------------------------------------------------------------------
# pointer alignment (reference: run_*_untransform_unaligned_test offsets both pointers by one byte,
# bc1 test_prelude.rs:364-373)
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fmt", FORMATS)
@pytest.mark.parametrize("shift", [1, 2, 4, 8])
def test_misaligned_device_pointers(pkg, oracle, dev, fmt, shift):
    n = 3 * TILE[fmt] + 11
    x = oracle.fill_splitmix64(n * BLOCK[fmt], 0xA11A + shift)
    for s in all_settings(fmt):
        xs = torch.zeros(x.size + shift, dtype=torch.uint8, device=dev)
        xs[shift:] = torch.from_numpy(x).to(dev)
        ys = torch.zeros(x.size + shift, dtype=torch.uint8, device=dev)
        st = pkg_settings(pkg, fmt, s)
        getattr(pkg, f"transform_{fmt}_with_settings")(xs[shift:], ys[shift:], st)
        torch.cuda.synchronize()
        want = fwd_oracle(oracle, fmt, x, s)
        assert np.array_equal(ys[shift:].cpu().numpy(), want), (fmt, shift, settings_id(s))
        assert int(ys[:shift].sum()) == 0, "wrote before the output pointer"
        zs = torch.zeros(x.size + shift, dtype=torch.uint8, device=dev)
        getattr(pkg, f"untransform_{fmt}_with_settings")(ys[shift:], zs[shift:], st)
        torch.cuda.synchronize()
        assert np.array_equal(zs[shift:].cpu().numpy(), x), (fmt, shift, settings_id(s), "inverse")


@pytest.mark.parametrize("fmt", FORMATS)
@pytest.mark.parametrize("threads", [64, 128, 256, 512])
def test_every_tile_workgroup_size(pkg, oracle, dev, fmt, threads):
    n = 9 * TILE[fmt] + 16 * 3 + 1
    x = oracle.fill_splitmix64(n * BLOCK[fmt], 0x512)
    for s in all_settings(fmt):
        want = fwd_oracle(oracle, fmt, x, s)
        try:
            pkg.set_tuning(threads, 0)
            got = run_device(pkg, fmt, x, s, dev)
            back = run_device(pkg, fmt, got, s, dev, inverse=True)
        finally:
            pkg.set_tuning(0, 0)
        assert np.array_equal(got, want), (fmt, settings_id(s))
        assert np.array_equal(back, x)


@pytest.mark.parametrize("fmt", FORMATS)
def test_shifted_tiles(pkg, oracle, dev, fmt):
    """Odd block counts put the stream bases off 16-byte alignment: the shifted-tile kernels take the body.  Every
    residue of N modulo 16 (BC3 split alphas: a1 starts at byte N), all settings; and the shifted kernels forced on
    aligned data must equal the aligned kernels."""
    t = TILE[fmt]
    for n in [3 * t + r for r in range(1, 17)] + [7 * t + 5, 100 * t + 33]:
        x = oracle.fill_splitmix64(n * BLOCK[fmt], 0x5111F7 + n)
        for s in all_settings(fmt):
            want = fwd_oracle(oracle, fmt, x, s)
            xd = torch.from_numpy(x).to(dev)
            yd = torch.full((x.size + 64,), 0x3C, dtype=torch.uint8, device=dev)
            st = pkg_settings(pkg, fmt, s)
            getattr(pkg, f"transform_{fmt}_with_settings")(xd, yd[: x.size], st)
            torch.cuda.synchronize()
            assert np.array_equal(yd[: x.size].cpu().numpy(), want), (fmt, n, settings_id(s))
            assert bool((yd[x.size:] == 0x3C).all()), "wrote past the end"
            zd = torch.full((x.size + 64,), 0x3C, dtype=torch.uint8, device=dev)
            getattr(pkg, f"untransform_{fmt}_with_settings")(yd[: x.size], zd[: x.size], st)
            torch.cuda.synchronize()
            assert np.array_equal(zd[: x.size].cpu().numpy(), x), (fmt, n, settings_id(s), "inverse")
            assert bool((zd[x.size:] == 0x3C).all())
    n = 6 * t
    x = oracle.fill_splitmix64(n * BLOCK[fmt], 0xF02CE)
    for s in all_settings(fmt):
        try:
            pkg.set_tuning(0, 2)
            got = run_device(pkg, fmt, x, s, dev)
            back = run_device(pkg, fmt, got, s, dev, inverse=True)
        finally:
            pkg.set_tuning(0, 0)
        assert np.array_equal(got, fwd_oracle(oracle, fmt, x, s)), (fmt, settings_id(s))
        assert np.array_equal(back, x)


@pytest.mark.parametrize("fmt", FORMATS)
def test_forward_shifted_tiles_both_forms(pkg, oracle, dev, fmt):
    """The forward halo tiles (whole 16-byte segments only) -- and, in the experiments side build, the first form with typed
    partial segments (switch 0x400) -- must equal the oracle on odd counts, on ranges that start at odd blocks and on an SoA
    pointer that is itself misaligned."""
    t = TILE[fmt]
    B = BLOCK[fmt]
    for force in ((0, 0x400) if pkg.tuning_mask() & 0x400 else (0,)):
        try:
            pkg.set_tuning(0, force)
            for n in (t + 1, 2 * t + 15, 9 * t + 7, 40 * t + 16 + 3):
                x = oracle.fill_splitmix64(n * B, 0xA110 + n)
                for s in all_settings(fmt):
                    assert np.array_equal(run_device(pkg, fmt, x, s, dev), fwd_oracle(oracle, fmt, x, s)), (force, n, settings_id(s))
            # ranges with odd starts, written into one whole buffer; guard bytes around it
            total = 11 * t + 9
            x = oracle.fill_splitmix64(total * B, 0xA111)
            xd = torch.from_numpy(x).to(dev)
            cuts = [0, 3 * t + 1, 3 * t + 1 + 2 * t, 8 * t + 5, total]
            for s in [(1, 1, 1), (2, 0, 1), (0, 1, 0)]:
                st = pkg_settings(pkg, fmt, s)
                for lead in (0, 1, 6):   # the SoA pointer itself off by `lead` bytes
                    buf = torch.full((x.size + 256,), 0x5E, dtype=torch.uint8, device=dev)
                    yd = buf[64 + lead: 64 + lead + x.size]
                    for a, b in zip(cuts, cuts[1:]):
                        pkg.transform_range(fmt, False, xd[a * B:], yd, total, a, b - a, st)
                    torch.cuda.synchronize()
                    assert np.array_equal(yd.cpu().numpy(), fwd_oracle(oracle, fmt, x, s)), (force, settings_id(s), lead)
                    assert bool((buf[:64 + lead] == 0x5E).all()) and bool((buf[64 + lead + x.size:] == 0x5E).all())
        finally:
            pkg.set_tuning(0, 0)


def test_shipped_library_has_no_experiment_switch(pkg, oracle, dev):
    """The shipped library honours two force_path bits (2: halo / shifted tiles always; 0x20: generic LDS accesses), both exact.
    Everything else -- the element kernel (1), the wrong-output timing switch (0x10), store policies, tile orders, old routings --
    lives in the -DDXTLT_EXPERIMENTS side build: here those bits must be absent from the mask and inert."""
    import os
    if os.environ.get("DXTLT_LIB_PATH"):
        pytest.skip("another build of the library is under test")
    assert pkg.tuning_mask() == 0x22
    for fmt in FORMATS:
        n = 9 * TILE[fmt] + 7
        x = oracle.fill_splitmix64(n * BLOCK[fmt], 0x71E + n)
        s = next(iter(all_settings(fmt)))
        try:
            for force in (1, 0x10, 0x10 | 0x400, 0x10 | 2, 0x40, 0x80, 0x100, 0x200, 0x800, 0x1000, 0x2000, 0x7FFFFFFF):
                pkg.set_tuning(0, force)
                assert np.array_equal(run_device(pkg, fmt, x, s, dev), fwd_oracle(oracle, fmt, x, s)), (fmt, hex(force))
        finally:
            pkg.set_tuning(0, 0)


def test_no_write_past_the_end(pkg, oracle, dev):
    """Stream sections are adjacent in one buffer: a wide store of one stream must never spill into the next
    (SURVEY.md 2, AVX-512 crib) nor past the end of the output."""
    for fmt in FORMATS:
        n = 2 * TILE[fmt] + 5
        x = oracle.fill_splitmix64(n * BLOCK[fmt], 0x5A5A)
        xd = torch.from_numpy(x).to(dev)
        yd = torch.full((x.size + 4096,), 0x77, dtype=torch.uint8, device=dev)
        s = (1, 1, 1)
        getattr(pkg, f"transform_{fmt}_with_settings")(xd, yd[: x.size], pkg_settings(pkg, fmt, s))
        torch.cuda.synchronize()
        assert np.array_equal(yd[: x.size].cpu().numpy(), fwd_oracle(oracle, fmt, x, s))
        assert bool((yd[x.size:] == 0x77).all())


# ------------------------------------------------------------------------------------------------------------
# committed fixtures through the GPU
# ------------------------------------------------------------------------------------------------------------
def test_golden_vectors_on_gpu(pkg, dev):
    for e in golden_vectors()["vectors"]:
        x = np.frombuffer(bytes.fromhex(e["input"]), dtype=np.uint8)
        s = (e["variant"], e["split_alpha"], e["split_colour"])
        assert run_device(pkg, e["fmt"], x, s, dev).tobytes().hex() == e["output"], e
        y = np.frombuffer(bytes.fromhex(e["output"]), dtype=np.uint8)
        assert run_device(pkg, e["fmt"], y, s, dev, inverse=True).tobytes().hex() == e["input"], e


def test_golden_digests_on_gpu(pkg, oracle, dev):
    cache = {}
    for e in golden_digests():
        key = (e["fmt"], e["source"], e.get("seed"), e["blocks"])
        if key not in cache:
            if e["source"] == "splitmix64":
                t = torch.empty(e["blocks"] * BLOCK[e["fmt"]], dtype=torch.uint8, device=dev)
                pkg.fill_splitmix64(t, e["seed"])
                cache = {key: t}
            else:
                cache = {key: torch.from_numpy(payload(e["fmt"])).to(dev)}
        x = cache[key]
        y = torch.empty_like(x)
        z = torch.empty_like(x)
        st = pkg_settings(pkg, e["fmt"], (e["variant"], e["split_alpha"], e["split_colour"]))
        getattr(pkg, f"transform_{e['fmt']}_with_settings")(x, y, st)
        getattr(pkg, f"untransform_{e['fmt']}_with_settings")(y, z, st)
        torch.cuda.synchronize()
        assert hashlib.sha256(x.cpu().numpy()).hexdigest() == e["input_sha256"], "fill kernel differs"
        assert hashlib.sha256(y.cpu().numpy()).hexdigest() == e["output_sha256"], e
        assert torch.equal(z, x)


def test_fill_kernel_matches_oracle_fill(pkg, oracle, dev):
    for nbytes, first in ((8 * 1000, 0), (8 * 1000 + 5, 17), (3, 2), (1 << 20, 1 << 33)):
        t = torch.zeros(nbytes + 8, dtype=torch.uint8, device=dev)
        pkg.fill_splitmix64(t[:nbytes], 0x0BC10002, first)
        torch.cuda.synchronize()
        assert np.array_equal(t[:nbytes].cpu().numpy(), oracle.fill_splitmix64(nbytes, 0x0BC10002, first))
        assert int(t[nbytes:].sum()) == 0


# ------------------------------------------------------------------------------------------------------------
# block ranges (multi-GPU shards / chunked staging) and the host-pointer entry points
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fmt", FORMATS)
def test_range_calls_compose_to_the_whole_buffer(pkg, oracle, dev, fmt):
    total = 7 * TILE[fmt] + 16 * 3 + 5
    x = oracle.fill_splitmix64(total * BLOCK[fmt], 0x5AAD)
    xd = torch.from_numpy(x).to(dev)
    cuts = [0, 2 * TILE[fmt], 2 * TILE[fmt] + 16, 5 * TILE[fmt] + 7, total]  # aligned and unaligned shard starts
    for s in [(1, 1, 1), (0, 0, 0), (2, 1, 0), (3, 0, 1)]:
        st = pkg_settings(pkg, fmt, s)
        yd = torch.zeros_like(xd)
        for a, b in zip(cuts, cuts[1:]):
            pkg.transform_range(fmt, False, xd[a * BLOCK[fmt]:], yd, total, a, b - a, st)
        torch.cuda.synchronize()
        want = fwd_oracle(oracle, fmt, x, s)
        assert np.array_equal(yd.cpu().numpy(), want), (fmt, settings_id(s))
        zd = torch.zeros_like(xd)
        for a, b in zip(cuts, cuts[1:]):
            pkg.transform_range(fmt, True, yd, zd[a * BLOCK[fmt]:], total, a, b - a, st)
        torch.cuda.synchronize()
        assert torch.equal(zd, xd)


@pytest.mark.parametrize("fmt", FORMATS)
def test_host_pointer_entry_points(pkg, oracle, dev, fmt):
    for n in (1, 37, TILE[fmt] * 3 + 9):
        x = oracle.fill_splitmix64(n * BLOCK[fmt], 0x4057 + n)
        for s in all_settings(fmt):
            st = pkg_settings(pkg, fmt, s)
            # +1-byte offset views: the reference's unaligned tests
            src = np.zeros(x.size + 1, dtype=np.uint8)
            src[1:] = x
            dst = np.zeros(x.size + 1, dtype=np.uint8)
            getattr(pkg, f"transform_{fmt}_with_settings")(src[1:], dst[1:], st)
            want = fwd_oracle(oracle, fmt, x, s)
            assert np.array_equal(dst[1:], want) and dst[0] == 0
            back = np.zeros_like(x)
            getattr(pkg, f"untransform_{fmt}_with_settings")(dst[1:], back, st)
            assert np.array_equal(back, x)


@pytest.mark.parametrize("fmt", FORMATS)
def test_host_pointer_chunked_pipeline(pkg, oracle, dev, fmt):
    """Buffers of 96 MiB and more take the chunked upload / kernel / download pipeline (two host threads, block-range
    kernels, 16 MiB chunks): several chunks, a ragged last chunk, an odd block count."""
    for nbytes_target in ((96 << 20), (100 << 20) + 48 * BLOCK[fmt] + BLOCK[fmt]):
        n = nbytes_target // BLOCK[fmt]
        x = oracle.fill_splitmix64(n * BLOCK[fmt], 0x919E + n)
        for s in [(1, 1, 1), (0, 0, 0), (3, 0, 1)]:
            st = pkg_settings(pkg, fmt, s)
            y = np.zeros_like(x)
            getattr(pkg, f"transform_{fmt}_with_settings")(x, y, st)
            want = np.empty_like(x)
            oracle.run_mt(fmt, x, want, s[0], bool(s[2]), bool(s[1]), False, 8)
            assert np.array_equal(y, want), (fmt, n, settings_id(s))
            z = np.zeros_like(x)
            getattr(pkg, f"untransform_{fmt}_with_settings")(y, z, st)
            assert np.array_equal(z, x), (fmt, n, settings_id(s), "inverse")


def test_device_calls_are_stream_ordered(pkg, oracle, dev):
    """The *_device entry points only enqueue: work lands on the stream that is current in torch, in order with the
    producer before it (a fill) and the consumer after it (the inverse), with no synchronisation in between; two
    streams run independent transforms at once."""
    n = 6 * 1024 * 1024  # blocks
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    bufs = []
    for i, st in enumerate(streams):
        fmt = ("bc1", "bc3")[i]
        s = (1, 1, 1)
        with torch.cuda.stream(st):
            x = torch.empty(n * BLOCK[fmt], dtype=torch.uint8, device=dev)
            y = torch.empty_like(x)
            z = torch.empty_like(x)
            pkg.fill_splitmix64(x, 0x57EA + i)
            getattr(pkg, f"transform_{fmt}_with_settings")(x, y, pkg_settings(pkg, fmt, s))
            getattr(pkg, f"untransform_{fmt}_with_settings")(y, z, pkg_settings(pkg, fmt, s))
        bufs.append((fmt, s, x, y, z, 0x57EA + i))
    for st in streams:
        st.synchronize()
    for fmt, s, x, y, z, seed in bufs:
        host = oracle.fill_splitmix64(x.numel(), seed)
        assert np.array_equal(x.cpu().numpy(), host)
        want = np.empty_like(host)
        oracle.run_mt(fmt, host, want, s[0], bool(s[2]), bool(s[1]), False, 8)
        assert np.array_equal(y.cpu().numpy(), want)
        assert torch.equal(z, x)


def test_concurrent_callers(pkg, oracle, dev):
    """The reference's functions are reentrant and its CLI calls them from rayon workers, one file per task
    (tools/dxt-lossless-transform-cli/src/commands/transform/mod.rs:166-184).  Eight threads hammer the host-pointer
    entry points with different formats, sizes and settings at once; every result must be exact."""
    import threading

    jobs = []
    for i in range(8):
        fmt = FORMATS[i % 3]
        n = 1000 + 7919 * i + (i % 2) * 300_000
        s = list(all_settings(fmt))[(5 * i) % (16 if fmt == "bc3" else 8)]
        x = oracle.fill_splitmix64(n * BLOCK[fmt], 0x7EAD + i)
        jobs.append((fmt, s, x, fwd_oracle(oracle, fmt, x, s)))
    errors = []

    def work(fmt, s, x, want):
        try:
            st = pkg_settings(pkg, fmt, s)
            for _ in range(6):
                y = np.zeros_like(x)
                getattr(pkg, f"transform_{fmt}_with_settings")(x, y, st)
                if not np.array_equal(y, want):
                    raise AssertionError(f"{fmt} {s}: forward mismatch")
                z = np.zeros_like(x)
                getattr(pkg, f"untransform_{fmt}_with_settings")(y, z, st)
                if not np.array_equal(z, x):
                    raise AssertionError(f"{fmt} {s}: inverse mismatch")
        except Exception as e:  # noqa: BLE001 - collected and re-raised on the main thread
            errors.append(e)

    threads = [threading.Thread(target=work, args=j) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_concurrent_callers_of_every_family(pkg, oracle, dev):
    """The same, across the families that share a thread's staging buffers and scratch: transforms, BC7, block
    normalisation, decoders, colour arrays and the pixel difference count, each from its own thread and all at once."""
    import threading

    from dxt_lossless_transform_amd import bc7, color565, decode, normalize, normalize23

    rng = np.random.default_rng(0xFA111E5)
    n = 40_003
    bc1 = rng.integers(0, 256, 8 * n, dtype=np.uint8)
    bc1.reshape(n, 8)[::3, 4:] = 0                                   # normalisable blocks
    bc3 = rng.integers(0, 256, 16 * n, dtype=np.uint8)
    bc3.reshape(n, 16)[::4, 2:8] = 0
    bc7_blocks = oracle.generate_bc7_mode_mixed(n, 0xB7) if hasattr(oracle, "generate_bc7_mode_mixed") else rng.integers(
        0, 256, 16 * n, dtype=np.uint8)
    cols = rng.integers(0, 256, 2 * 60_001, dtype=np.uint8)
    want = {
        "norm1": oracle.normalize_bc1_blocks(bc1, 1),
        "norm3": oracle.normalize_bc3_blocks(bc3, 1, 2),
        "dec1": oracle.decode_blocks("bc1", bc1),
        "dec3": oracle.decode_blocks("bc3", bc3),
        "fwd3": oracle.transform("bc3", bc3, 1, True, True),
        "diff": oracle.count_pixel_differences("bc1", bc1, oracle.normalize_bc1_blocks(bc1, 2)),
    }
    errors = []

    def guarded(fn):
        def run():
            try:
                for _ in range(5):
                    fn()
            except Exception as e:  # noqa: BLE001 - collected and re-raised on the main thread
                errors.append(e)
        return run

    def t_norm1():
        out = np.zeros_like(bc1)
        normalize.normalize_blocks(bc1, out, normalize.ColorNormalizationMode.COLOR0_ONLY)
        assert np.array_equal(out, want["norm1"])

    def t_norm3():
        out = np.zeros_like(bc3)
        normalize23.normalize_blocks("bc3", bc3, out, normalize.ColorNormalizationMode.REPLICATE_COLOR,
                                     normalize23.AlphaNormalizationMode.UNIFORM_ALPHA_ZERO_INDICES)
        assert np.array_equal(out, want["norm3"])

    def t_dec():
        out = np.zeros(64 * n, dtype=np.uint8)
        decode.decode_blocks("bc1", bc1, out)
        assert np.array_equal(out, want["dec1"])
        decode.decode_blocks("bc3", bc3, out)
        assert np.array_equal(out, want["dec3"])

    def t_diff():
        assert decode.count_pixel_differences("bc1", bc1, want["norm1"]) == 0
        assert decode.count_pixel_differences("bc1", bc1, oracle.normalize_bc1_blocks(bc1, 2)) == want["diff"] == 0

    def t_cols():
        out = np.zeros_like(cols)
        color565.decorrelate_ycocg_r(cols, out, 3)
        back = np.zeros_like(cols)
        color565.recorrelate_ycocg_r(out, back, 3)
        assert np.array_equal(back, cols) and not np.array_equal(out, cols)

    def t_bc7():
        y, z = np.zeros_like(bc7_blocks), np.zeros_like(bc7_blocks)
        bc7.transform_bc7(bc7_blocks, y)
        bc7.untransform_bc7(y, z)
        assert np.array_equal(z, bc7_blocks)

    def t_fwd3():
        y = np.zeros_like(bc3)
        pkg.transform_bc3_with_settings(bc3, y)
        assert np.array_equal(y, want["fwd3"])

    threads = [threading.Thread(target=guarded(f)) for f in (t_norm1, t_norm3, t_dec, t_diff, t_cols, t_bc7, t_fwd3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


@pytest.mark.parametrize("fmt", FORMATS)
def test_sharded_entry_point_on_one_gpu(pkg, oracle, dev, fmt):
    n = 9 * 2048 + 123
    x = oracle.fill_splitmix64(n * BLOCK[fmt], 0x5AAD0 + n)
    s = (1, 1, 1)
    st = pkg_settings(pkg, fmt, s)
    y = np.zeros_like(x)
    pkg.transform_sharded(fmt, False, x, y, st, 0)
    assert np.array_equal(y, fwd_oracle(oracle, fmt, x, s))
    z = np.zeros_like(x)
    pkg.transform_sharded(fmt, True, y, z, st, 0)
    assert np.array_equal(z, x)


def test_randomised_dispatch(pkg, oracle, dev):
    """400 seeded random cases: format, settings, block count (0..40 000), and independent byte offsets of the input and
    output device pointers (0..127: the halo tiles move their windows back to 64-byte boundaries, so stream bases of
    every residue mod 64 must come up) -- every combination of aligned-tile / shifted-tile / element-kernel dispatch with
    ragged tails -- forward bytes against the oracle, inverse against the input, and guard bytes on both sides."""
    rng = np.random.default_rng(0xD15BA7C4)
    for case in range(400):
        fmt = FORMATS[int(rng.integers(0, 3))]
        settings = list(all_settings(fmt))
        s = settings[int(rng.integers(0, len(settings)))]
        n = int(rng.integers(0, 40_001)) if case % 4 else int(rng.integers(0, 40)) * TILE[fmt]
        a, b = int(rng.integers(0, 128)) if case % 3 else 0, int(rng.integers(0, 128)) if case % 5 else 0
        nbytes = n * BLOCK[fmt]
        x = oracle.fill_splitmix64(nbytes, 0xFA22 + case)
        st = pkg_settings(pkg, fmt, s)
        xd = torch.full((nbytes + 256,), 0x11, dtype=torch.uint8, device=dev)
        xd[a:a + nbytes] = torch.from_numpy(x).to(dev)
        yd = torch.full((nbytes + 256,), 0x22, dtype=torch.uint8, device=dev)
        getattr(pkg, f"transform_{fmt}_with_settings")(xd[a:a + nbytes], yd[b:b + nbytes], st)
        zd = torch.full((nbytes + 256,), 0x33, dtype=torch.uint8, device=dev)
        getattr(pkg, f"untransform_{fmt}_with_settings")(yd[b:b + nbytes], zd[a:a + nbytes], st)
        torch.cuda.synchronize()
        tag = (case, fmt, settings_id(s), n, a, b)
        yh, zh = yd.cpu().numpy(), zd.cpu().numpy()
        assert np.array_equal(yh[b:b + nbytes], fwd_oracle(oracle, fmt, x, s)), tag
        assert np.array_equal(zh[a:a + nbytes], x), tag
        assert (yh[:b] == 0x22).all() and (yh[b + nbytes:] == 0x22).all(), tag
        assert (zh[:a] == 0x33).all() and (zh[a + nbytes:] == 0x33).all(), tag


def test_host_pointers_beyond_four_gib(pkg, oracle, dev):
    """A single host buffer just over 2^32 bytes through the drop-in entry point (chunked pipeline, 32 MiB chunks,
    per-stream slices at offsets above 4 GiB): sampled windows against the oracle and an exact round trip."""
    fmt, s = "bc3", (1, 1, 1)
    n = (1 << 32) // 16 + 3 * 1024 + 1  # odd block count on top
    x = oracle.fill_splitmix64(n * 16, 0x4A1B)
    st = pkg_settings(pkg, fmt, s)
    y = np.empty_like(x)
    pkg.transform_bc3_with_settings(x, y, st)
    table = pkg.stream_table(fmt, st)
    win = 4096
    for first in (0, n - win, (1 << 32) // 16 - win // 2, n // 3):
        want = fwd_oracle(oracle, fmt, x[first * 16:(first + win) * 16], s)
        got = np.empty_like(want)
        for off, w in table:
            got[off * win: off * win + w * win] = y[off * n + w * first: off * n + w * (first + win)]
        assert np.array_equal(got, want), first
    z = np.empty_like(x)
    pkg.untransform_bc3_with_settings(y, z, st)
    assert np.array_equal(z, x)


def test_mixed_bc1_bc3_archive(pkg, oracle, dev):
    """BASELINE.json configs[4] in miniature on the visible device(s): an archive of alternating BC1 / BC3 textures
    (16 MiB each, ragged sizes too), each transformed with its format's default settings through the sharded entry point
    (contiguous block ranges per device, per-stream placement on the host), compared byte for byte with the CPU oracle;
    then restored."""
    rng = np.random.default_rng(55)
    for i in range(8):
        fmt = "bc1" if i % 2 == 0 else "bc3"
        blocks = (16 << 20) // BLOCK[fmt] + int(rng.integers(0, 3)) * int(rng.integers(1, 5000))
        x = oracle.fill_splitmix64(blocks * BLOCK[fmt], 0x0A5C0005 + i)
        st = pkg_settings(pkg, fmt, (1, 1, 1))
        y = np.zeros_like(x)
        pkg.transform_sharded(fmt, False, x, y, st, 0)
        want = np.empty_like(x)
        oracle.run_mt(fmt, x, want, 1, True, True, False, 8)
        assert np.array_equal(y, want), (i, fmt, blocks)
        z = np.zeros_like(x)
        pkg.transform_sharded(fmt, True, y, z, st, 0)
        assert np.array_equal(z, x), (i, fmt, blocks, "inverse")


@pytest.mark.parametrize("fmt,shards", [("bc1", 3), ("bc3", 5), ("bc2", 2)])
def test_sharded_entry_point_many_shards_pipelined(pkg, oracle, dev, fmt, shards):
    """dxtlt_transform_sharded with several shards, each large enough (>= 96 MiB) to take the chunked pipeline with a
    non-zero base block: every visible device is used, more shards than devices are dealt round robin.  Whole-buffer
    comparison with the oracle, then the inverse."""
    blocks = shards * ((112 << 20) // BLOCK[fmt]) + 12_345
    x = oracle.fill_splitmix64(blocks * BLOCK[fmt], 0x5AAD + shards)
    st = pkg_settings(pkg, fmt, (1, 1, 1))
    y = np.zeros_like(x)
    pkg.transform_sharded(fmt, False, x, y, st, shards)
    want = np.empty_like(x)
    oracle.run_mt(fmt, x, want, 1, True, True, False, 8)
    assert np.array_equal(y, want)
    # the per-device shard contexts (stream + buffers) are kept across calls; releasing them in between must not matter
    pkg.load().dxtlt_release_thread_resources()
    z = np.zeros_like(x)
    pkg.transform_sharded(fmt, True, y, z, st, shards)
    assert np.array_equal(z, x)
    y2 = np.zeros_like(x)
    pkg.transform_sharded(fmt, False, x, y2, st, max(1, shards - 1))      # reuse with another shard size
    assert np.array_equal(y2, want)


def test_sharded_entry_point_on_every_visible_device(pkg, oracle, dev):
    """More than one device visible (the driver's 8-GPU node): the same call over all of them, one shard per device."""
    if torch.cuda.device_count() < 2:
        pytest.skip("one device visible")
    fmt = "bc3"
    blocks = (1 << 30) // 16 + 77
    x = oracle.fill_splitmix64(blocks * 16, 0xD371CE5)
    st = pkg_settings(pkg, fmt, (1, 1, 1))
    y = np.zeros_like(x)
    pkg.transform_sharded(fmt, False, x, y, st, 0)
    want = np.empty_like(x)
    oracle.run_mt(fmt, x, want, 1, True, True, False, 8)
    assert np.array_equal(y, want)
    z = np.zeros_like(x)
    pkg.transform_sharded(fmt, True, y, z, st, 0)
    assert np.array_equal(z, x)


def test_mixed_archive_over_every_visible_device(pkg, oracle, dev):
    """BASELINE.json configs[4] in its defining shape, activated by >= 2 visible devices (the driver's 8-GPU node; skips on
    a one-GPU box): alternating 256 MiB BC1 / BC3 textures, one per device and format, every texture through
    dxtlt_transform_sharded over ALL devices (contiguous block ranges, host concatenation), plus a BC7 texture through
    dxtlt_transform_bc7_sharded.  Whole-buffer equality with the oracle, exact round trips, and one shard per device in the
    stats."""
    n_dev = torch.cuda.device_count()
    if n_dev < 2:
        pytest.skip("one device visible")
    from dxt_lossless_transform_amd import bc7

    tex = 256 << 20
    for i in range(2 * n_dev):
        fmt = "bc1" if i % 2 == 0 else "bc3"
        st = pkg_settings(pkg, fmt, (1, 1, 1))
        x = oracle.fill_splitmix64(tex, 0x0A5C0005 + i)
        y = np.zeros_like(x)
        pkg.transform_sharded(fmt, False, x, y, st, 0)
        stats = pkg.sharded_last_stats()
        assert sorted(s["device"] for s in stats) == list(range(n_dev)), stats
        assert sum(s["blocks"] for s in stats) == tex // pkg.BLOCK_BYTES[fmt]
        want = np.empty_like(x)
        oracle.run_mt(fmt, x, want, 1, True, True, False, 8)
        assert np.array_equal(y, want), (i, fmt)
        z = np.zeros_like(x)
        pkg.transform_sharded(fmt, True, y, z, st, 0)
        assert np.array_equal(z, x), (i, fmt)
    x = oracle.fill_splitmix64(tex + 16 * 333, 0x0A5C0707)      # whole granules and a tail part
    oracle.bc7_force_modes(x)
    y, z = np.zeros_like(x), np.zeros_like(x)
    bc7.transform_bc7_sharded(x, y, 0)
    assert np.array_equal(y, oracle.transform_bc7(x))
    bc7.transform_bc7_sharded(y, z, 0, inverse=True)
    assert np.array_equal(z, x)


def test_real_textures(pkg, oracle, dev):
    """assets/tests/r2-256-bc{1,2,3}.dds payloads: every settings combination, forward bytes and round trip
    (reference: debug_bcN roundtrip commands)."""
    for fmt in FORMATS:
        p = payload(fmt)
        for s in all_settings(fmt):
            got = run_device(pkg, fmt, p, s, dev)
            assert np.array_equal(got, fwd_oracle(oracle, fmt, p, s)), (fmt, settings_id(s))
            assert np.array_equal(run_device(pkg, fmt, got, s, dev, inverse=True), p)


# ------------------------------------------------------------------------------------------------------------
# medium size: exact against the oracle, whole buffer
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fmt", FORMATS)
def test_one_gib_exact(pkg, oracle, dev, fmt):
    nbytes = 1 << 30
    s = (1, 1, 1)
    st = pkg_settings(pkg, fmt, s)
    x = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    pkg.fill_splitmix64(x, 0x0BC10002)
    y = torch.empty_like(x)
    getattr(pkg, f"transform_{fmt}_with_settings")(x, y, st)
    torch.cuda.synchronize()
    host_x = oracle.fill_splitmix64(nbytes, 0x0BC10002)
    want = np.empty_like(host_x)
    oracle.run_mt(fmt, host_x, want, s[0], bool(s[2]), bool(s[1]), False, 8)
    assert np.array_equal(y.cpu().numpy(), want)
    z = torch.empty_like(x)
    getattr(pkg, f"untransform_{fmt}_with_settings")(y, z, st)
    torch.cuda.synchronize()
    assert torch.equal(z, x)


# ------------------------------------------------------------------------------------------------------------
# BASELINE.json full sizes: configs[1] (BC1, 8 GiB) and configs[2] (BC3, 8 GiB)
# ------------------------------------------------------------------------------------------------------------
def _window_check(pkg, oracle, fmt, s, x, y, total_blocks, first, count):
    """Exact check of blocks [first, first+count): gather the window's slice of every stream from the device
    output and compare with the oracle run on the window alone."""
    B = BLOCK[fmt]
    st = pkg_settings(pkg, fmt, s)
    xin = x[first * B:(first + count) * B].cpu().numpy()
    want = fwd_oracle(oracle, fmt, xin, s)
    got = np.empty_like(want)
    for off, w in pkg.stream_table(fmt, st):
        sl = y[off * total_blocks + w * first: off * total_blocks + w * (first + count)].cpu().numpy()
        got[off * count: off * count + w * count] = sl
    assert np.array_equal(got, want), (fmt, first, count)


@pytest.mark.parametrize("fmt,seed,s,drop", [
    ("bc1", 0x0BC10002, (1, 1, 1), 0), ("bc3", 0x0BC30003, (1, 1, 1), 0),   # BASELINE.json configs[1], [2]: default settings
    ("bc1", 0x0BC10002, (2, 0, 0), 0),      # one non-default combination per format at the full size
    ("bc3", 0x0BC30003, (3, 0, 1), 0),
    ("bc2", 0x0BC20002, (1, 0, 1), 0),      # BC2 at 8 GiB once
    ("bc3", 0x0BC30003, (1, 1, 1), 5),      # 2^29 - 5 blocks: every stream base misaligned (halo tiles, shifted inverse)
])
def test_eight_gib_properties(pkg, oracle, dev, fmt, seed, s, drop):
    B = BLOCK[fmt]
    nbytes = (8 << 30) - drop * B
    total = nbytes // B
    st = pkg_settings(pkg, fmt, s)
    x = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    pkg.fill_splitmix64(x, seed)
    y = torch.empty_like(x)
    getattr(pkg, f"transform_{fmt}_with_settings")(x, y, st)
    torch.cuda.synchronize()
    # sampled windows, including both ends and a 2^32-byte-offset crossing
    win = 64 * 1024
    firsts = [0, total - win, (1 << 32) // B - win // 2, total // 2 + 12345, total // 3, 7 * (total // 8) + 1]
    for f in firsts:
        _window_check(pkg, oracle, fmt, s, x, y, total, f, win)
    # index streams are copied verbatim: their 64-bit sum equals the sum of the index fields of the input
    xi = x.view(torch.int32).view(-1, B // 4)
    idx_sum_in = int(xi[:, -1].to(torch.int64).sum())
    idx_sum_out = int(y[(B - 4) * total:].view(torch.int32).to(torch.int64).sum())
    assert idx_sum_in == idx_sum_out
    del xi
    # exact round trip at full size
    z = torch.empty_like(x)
    getattr(pkg, f"untransform_{fmt}_with_settings")(y, z, st)
    torch.cuda.synchronize()
    assert torch.equal(z, x)


@pytest.mark.parametrize("fmt,seed", [("bc1", 0x0BC10640), ("bc3", 0x0BC30640)])
def test_sixty_four_gib_round_trip(pkg, oracle, dev, fmt, seed):
    """Sized for the 288 GB of one MI355X rather than for a CPU's caches: one 64 GiB buffer (2^33 BC1 blocks, 32 M
    workgroups in one launch; stream bases and block indices well past 2^32).  Sampled windows against the oracle at
    both ends and across the 2^32-, 2^33-, 2^34- and 2^35-byte offsets, then an exact round trip."""
    nbytes = 64 << 30
    torch.cuda.empty_cache()   # memory cached from earlier tests counts as used in mem_get_info
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 3 * nbytes + (8 << 30):
        pytest.skip(f"needs three 64 GiB buffers, {free >> 30} GiB free")
    B = BLOCK[fmt]
    total = nbytes // B
    s = (1, 1, 1)
    st = pkg_settings(pkg, fmt, s)
    x = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    pkg.fill_splitmix64(x, seed)
    y = torch.empty_like(x)
    getattr(pkg, f"transform_{fmt}_with_settings")(x, y, st)
    torch.cuda.synchronize()
    win = 32 * 1024
    firsts = [0, total - win, total // 2 - win // 2, total // 3 + 1, 5 * (total // 7)]
    firsts += [(1 << p) // B - win // 2 for p in (32, 33, 34, 35)]
    for f in firsts:
        _window_check(pkg, oracle, fmt, s, x, y, total, f, win)
    z = torch.empty_like(x)
    getattr(pkg, f"untransform_{fmt}_with_settings")(y, z, st)
    torch.cuda.synchronize()
    step = 1 << 30   # torch.equal on the whole buffer would allocate another 64 GiB
    for lo in range(0, nbytes, step):
        assert torch.equal(z[lo:lo + step], x[lo:lo + step]), (fmt, lo)


def test_element_wise_kernels_past_2_32_lanes(pkg, oracle, dev):
    """72 GiB at 16 bytes per lane is 1.125 * 2^32 lanes: more threads than one HIP launch dimension holds, so the
    element-wise kernels run on a two-dimensional grid (csrc/launch_grid.h).  BC1 normalisation: sampled windows
    against the oracle around the 2^32-lane row boundary and at both ends, and no decoded pixel changes anywhere;
    colour decorrelation in place and back gives the source."""
    from dxt_lossless_transform_amd import color565, decode, normalize

    nbytes = 72 << 30
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 2 * nbytes + (8 << 30):
        pytest.skip(f"needs two 72 GiB buffers, {free >> 30} GiB free")
    x = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    pkg.fill_splitmix64(x, 0x72)
    # make a quarter of the blocks normalisable: one index value per block
    v = x.view(-1, 8)
    step = 1 << 27
    for lo in range(0, v.shape[0], step):
        v[lo:lo + step:4, 4:] = 0
    y = torch.empty_like(x)
    mode = normalize.ColorNormalizationMode.COLOR0_ONLY
    normalize.normalize_blocks(x, y, mode)
    torch.cuda.synchronize()
    win = 1 << 20
    for at in (0, nbytes - win, (64 << 30) - win // 2, (64 << 30) + (1 << 30), 36 << 30):
        want = oracle.normalize_bc1_blocks(x[at:at + win].cpu().numpy(), int(mode))
        assert np.array_equal(y[at:at + win].cpu().numpy(), want), at
    assert decode.count_pixel_differences("bc1", x, y) == 0
    changed = sum(int((x[lo:lo + (1 << 30)] != y[lo:lo + (1 << 30)]).any()) for lo in (0, 65 << 30, 71 << 30))
    assert changed == 3          # the kernel did rewrite blocks in the first, a middle and the last GiB
    # colour arrays: y <- decorrelate(x) (out of place), then recorrelate in place
    color565.decorrelate_ycocg_r(x, y, 2)
    for at in (0, nbytes - win, (64 << 30) - win // 2, (64 << 30) + 12345 * 16):
        cols = x[at:at + win].cpu().numpy().view("<u2")
        want = np.array([oracle.decorrelate(int(c), 2) for c in cols[:4096]], dtype="<u2")
        assert np.array_equal(y[at:at + 8192].cpu().numpy().view("<u2"), want), at
    color565.recorrelate_ycocg_r(y, y, 2)
    torch.cuda.synchronize()
    for lo in range(0, nbytes, 1 << 30):
        assert torch.equal(y[lo:lo + (1 << 30)], x[lo:lo + (1 << 30)]), lo
