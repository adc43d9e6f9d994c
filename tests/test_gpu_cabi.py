"""The reference-shaped C APIs (core dltbcNcore_*, stable dltbcN_*) end to end on the GPU, against the oracle."""
import ctypes as C

import zlib

import numpy as np
import pytest

import cabi
from helpers import BLOCK, all_settings, payload
from oracle import oracle_auto

pytestmark = pytest.mark.gpu

CORE_S = {1: cabi.CoreSettings2, 2: cabi.CoreSettings2, 3: cabi.CoreSettings3}
FMT = {1: "bc1", 2: "bc2", 3: "bc3"}
STABLE_OF_CORE = {1: 0, 2: 1, 3: 2, 0: 3}  # api-common reexports/color_565.rs:65-91


@pytest.fixture(scope="module")
def lib(pkg):
    import torch

    assert torch.cuda.is_available()
    return cabi.bind(C.CDLL(pkg._lib.lib_path()))


def core_settings(n, s):
    v, sa, sc = s
    st = CORE_S[n]()
    st.DecorrelationMode = v
    st.SplitColourEndpoints = bool(sc)
    if n == 3:
        st.SplitAlphaEndpoints = bool(sa)
    return st


@pytest.mark.parametrize("n", [1, 2, 3])
def test_core_transform_untransform(lib, oracle, n):
    fmt = FMT[n]
    for blocks in (2, 777, 5000):
        x = oracle.generate_test_data(fmt, blocks) if blocks < 1000 else oracle.fill_splitmix64(blocks * BLOCK[fmt], 5)
        for s in all_settings(fmt):
            st = core_settings(n, s)
            y = np.zeros(x.size + 3, dtype=np.uint8)  # output may be larger than the input
            r = getattr(lib, f"dltbc{n}core_transform")(x.ctypes.data, x.size, y.ctypes.data, y.size, st)
            assert r.ErrorCode == 0
            want = oracle.transform(fmt, x, s[0], s[2], s[1])
            assert np.array_equal(y[: x.size], want) and not y[x.size:].any()
            z = np.zeros_like(x)
            r = getattr(lib, f"dltbc{n}core_untransform")(y.ctypes.data, x.size, z.ctypes.data, z.size, st)
            assert r.ErrorCode == 0 and np.array_equal(z, x)


@pytest.mark.parametrize("n", [1, 2, 3])
def test_host_buffers_either_side_of_the_mapped_staging_limit(lib, oracle, n):
    """host pointers: up to 1 MiB the kernel reads and writes mapped pinned staging buffers itself, above that the buffer
    travels by two copies (and from 96 MiB by the chunked pipeline, tests/test_gpu_parity.py); same bytes either way, odd
    block counts and a misaligned caller buffer included"""
    fmt = FMT[n]
    B = BLOCK[fmt]
    s = list(all_settings(fmt))[-1]
    st = core_settings(n, s)
    for nbytes in (B, (1 << 20) - B, 1 << 20, (1 << 20) + B, (3 << 20) + 5 * B):
        buf = np.zeros(nbytes + 1, dtype=np.uint8)
        x = buf[1:]                                # the caller's input at an odd address
        x[:] = oracle.fill_splitmix64(nbytes, 0x51 + n)
        y = np.full(nbytes + 16, 0xEE, dtype=np.uint8)
        r = getattr(lib, f"dltbc{n}core_transform")(x.ctypes.data, x.size, y.ctypes.data, nbytes, st)
        assert r.ErrorCode == 0
        assert np.array_equal(y[:nbytes], oracle.transform(fmt, x, s[0], s[2], s[1])) and (y[nbytes:] == 0xEE).all()
        z = np.full(nbytes + 16, 0xDD, dtype=np.uint8)
        r = getattr(lib, f"dltbc{n}core_untransform")(y.ctypes.data, nbytes, z.ctypes.data, nbytes, st)
        assert r.ErrorCode == 0 and np.array_equal(z[:nbytes], x) and (z[nbytes:] == 0xDD).all()


@pytest.mark.parametrize("n", [1, 2, 3])
def test_stable_manual_builder(lib, oracle, n):
    fmt = FMT[n]
    p = f"dltbc{n}_"
    x = oracle.fill_splitmix64(3001 * BLOCK[fmt], 77)
    b = getattr(lib, p + "new_ManualTransformBuilder")()
    y = np.zeros_like(x)
    # defaults: Variant1 + split (bc1-api manual_transform_builder.rs:24-36; BC3: + split alphas)
    assert getattr(lib, p + "ManualTransformBuilder_Transform")(x.ctypes.data, x.size, y.ctypes.data, y.size, b).ErrorCode == 0
    assert np.array_equal(y, oracle.transform(fmt, x, 1, True))
    for v, sa, sc in all_settings(fmt):
        getattr(lib, p + "ManualTransformBuilder_SetDecorrelationMode")(b, STABLE_OF_CORE[v])
        getattr(lib, p + "ManualTransformBuilder_SetSplitColourEndpoints")(b, bool(sc))
        if n == 3:
            lib.dltbc3_ManualTransformBuilder_SetSplitAlphaEndpoints(b, bool(sa))
        c = getattr(lib, p + "clone_ManualTransformBuilder")(b)
        assert getattr(lib, p + "ManualTransformBuilder_Transform")(x.ctypes.data, x.size, y.ctypes.data, y.size, b).ErrorCode == 0
        assert np.array_equal(y, oracle.transform(fmt, x, v, sc, sa if n == 3 else True)), (v, sa, sc)
        z = np.zeros_like(x)
        assert getattr(lib, p + "ManualTransformBuilder_Untransform")(y.ctypes.data, y.size, z.ctypes.data, z.size, c).ErrorCode == 0
        assert np.array_equal(z, x)
        getattr(lib, p + "free_ManualTransformBuilder")(c)
    getattr(lib, p + "ManualTransformBuilder_ResetToDefaults")(b)
    assert getattr(lib, p + "ManualTransformBuilder_Transform")(x.ctypes.data, x.size, y.ctypes.data, y.size, b).ErrorCode == 0
    assert np.array_equal(y, oracle.transform(fmt, x, 1, True))
    getattr(lib, p + "free_ManualTransformBuilder")(b)


@pytest.mark.parametrize("n", [1, 2, 3])
@pytest.mark.parametrize("use_all", [False, True])
def test_core_auto_matches_reference_algorithm(lib, oracle, n, use_all):
    fmt = FMT[n]
    x = payload(fmt)
    for kind in ("dummy", "dummy0", "zlib"):
        log = []
        est, py_est = cabi.make_estimator(kind, log)
        want_choice, want_out, want_calls = oracle_auto.transform_auto(fmt, x, lambda b: py_est(bytes(b)), use_all)
        y = np.zeros_like(x)
        out = CORE_S[n]()
        r = getattr(lib, f"dltbc{n}core_transform_auto")(x.ctypes.data, x.size, y.ctypes.data, y.size, C.byref(est),
                                                         cabi.AutoSettings(use_all), C.byref(out))
        assert r.ErrorCode == 0
        got = (out.DecorrelationMode, int(out.SplitAlphaEndpoints) if n == 3 else 0, int(out.SplitColourEndpoints))
        assert got == want_choice, (kind, got, want_choice)
        assert np.array_equal(y, want_out)
        assert log == [ln for _, ln in want_calls]  # same sections, same order
        z = np.zeros_like(x)
        assert getattr(lib, f"dltbc{n}core_untransform")(y.ctypes.data, y.size, z.ctypes.data, z.size, out).ErrorCode == 0
        assert np.array_equal(z, x)
    # every byte the estimator is shown, candidate by candidate: the sections come from the fused candidate kernel's
    # arena (one read of the input), the oracle's from one full transform per candidate
    for blocks in (1, 2, 3, 255, 256, 257, 4099):
        xs = np.ascontiguousarray(np.resize(x, blocks * (8 if n == 1 else 16)))
        log, seen = [], []
        est, py_est = cabi.make_estimator("crc", log)

        def spy(b):
            seen.append((len(b), zlib.crc32(bytes(b))))
            return py_est(bytes(b))

        want_choice, want_out, _ = oracle_auto.transform_auto(fmt, xs, spy, use_all)
        y = np.zeros_like(xs)
        out = CORE_S[n]()
        r = getattr(lib, f"dltbc{n}core_transform_auto")(xs.ctypes.data, xs.size, y.ctypes.data, y.size, C.byref(est),
                                                         cabi.AutoSettings(use_all), C.byref(out))
        assert r.ErrorCode == 0 and log == seen, (blocks, use_all)
        got = (out.DecorrelationMode, int(out.SplitAlphaEndpoints) if n == 3 else 0, int(out.SplitColourEndpoints))
        assert got == want_choice and np.array_equal(y, want_out), blocks
    if not use_all and n != 3:
        # with a constant estimator the strict `<` keeps the FIRST candidate: None / NoSplit (settings.rs:81-86)
        assert oracle_auto.transform_auto(fmt, x, len, False)[0] == (0, 0, 0)


@pytest.mark.parametrize("n", [1, 2, 3])
def test_core_auto_estimator_failures(lib, n):
    fmt = FMT[n]
    x = payload(fmt)
    y = np.zeros_like(x)
    out = CORE_S[n]()
    for kind in ("fail_max", "fail_est"):
        est, _ = cabi.make_estimator(kind)
        r = getattr(lib, f"dltbc{n}core_transform_auto")(x.ctypes.data, x.size, y.ctypes.data, y.size, C.byref(est),
                                                         cabi.AutoSettings(False), C.byref(out))
        assert r.ErrorCode == 7  # SizeEstimationError


@pytest.mark.parametrize("n", [1, 2, 3])
def test_stable_auto_builder(lib, oracle, n):
    fmt = FMT[n]
    p = f"dltbc{n}_"
    x = payload(fmt)
    from tools import zstd_ratio
    for use_all, kind in ((False, "zlib"), (True, "zlib")) + (((False, "zstd"), (True, "zstd")) if zstd_ratio.available() else ()):
        est, py_est = cabi.make_estimator(kind)
        want_choice, want_out, _ = oracle_auto.transform_auto(fmt, x, lambda b: py_est(bytes(b)), use_all)
        ab = getattr(lib, p + "new_AutoTransformBuilder")(C.byref(est))
        assert getattr(lib, p + "AutoTransformBuilder_SetUseAllDecorrelationModes")(ab, use_all).ErrorCode == 0
        y = np.zeros_like(x)
        mb = C.c_void_p()
        r = getattr(lib, p + "AutoTransformBuilder_Transform")(ab, x.ctypes.data, x.size, y.ctypes.data, y.size, C.byref(mb))
        assert r.ErrorCode == 0 and mb.value
        assert np.array_equal(y, want_out)
        # the returned manual builder carries the chosen settings: it untransforms the result...
        z = np.zeros_like(x)
        assert getattr(lib, p + "ManualTransformBuilder_Untransform")(y.ctypes.data, y.size, z.ctypes.data, z.size, mb).ErrorCode == 0
        assert np.array_equal(z, x)
        # ...and transforming again with it reproduces the auto output
        y2 = np.zeros_like(x)
        assert getattr(lib, p + "ManualTransformBuilder_Transform")(x.ctypes.data, x.size, y2.ctypes.data, y2.size, mb).ErrorCode == 0
        assert np.array_equal(y2, y)
        getattr(lib, p + "free_ManualTransformBuilder")(mb)
        getattr(lib, p + "free_AutoTransformBuilder")(ab)
    est, _ = cabi.make_estimator("fail_est")
    ab = getattr(lib, p + "new_AutoTransformBuilder")(C.byref(est))
    mb = C.c_void_p(1)
    r = getattr(lib, p + "AutoTransformBuilder_Transform")(ab, x.ctypes.data, x.size, y.ctypes.data, y.size, C.byref(mb))
    assert r.ErrorCode == 4 and mb.value is None  # SizeEstimationFailed, output builder NULL
    getattr(lib, p + "free_AutoTransformBuilder")(ab)


@pytest.mark.gpu
def test_release_thread_resources_between_calls(pkg, oracle):
    """dxtlt_release_thread_resources frees every per-thread device resource (staging buffers, BC7 scratch,
    normalisation flag, batch tables); the next call of each family allocates again and still computes the same."""
    import torch

    from dxt_lossless_transform_amd import batch, bc7, normalize

    l = pkg.load()
    l.dxtlt_release_thread_resources.argtypes, l.dxtlt_release_thread_resources.restype = [], None
    x1 = oracle.fill_splitmix64(8 * 5001, 1)
    x7 = oracle.fill_splitmix64(16 * 3001, 7)
    dev = torch.device("cuda:0")
    xd = torch.from_numpy(x1).to(dev)
    for _ in range(3):
        y = np.zeros_like(x1)
        pkg.transform_bc1_with_settings(x1, y)
        assert np.array_equal(y, oracle.transform("bc1", x1, 1, True))
        y7 = np.zeros_like(x7)
        bc7.transform_bc7(x7, y7)
        assert np.array_equal(y7, oracle.transform_bc7(x7))
        outs = [np.zeros_like(x1) for _ in range(3)]
        normalize.normalize_blocks_all_modes(x1, outs)
        assert np.array_equal(outs[1], oracle.normalize_bc1_blocks(x1, 1))
        yd = torch.empty_like(xd)
        batch.transform_batch([("bc1", False, xd, yd, pkg.Bc1TransformSettings())] * 2)
        assert np.array_equal(yd.cpu().numpy(), oracle.transform("bc1", x1, 1, True))
        torch.cuda.synchronize()
        l.dxtlt_release_thread_resources()


# ---- opt-in: the estimator on several host threads (dxtlt_set_auto_estimator_threads) --------------------------------
@pytest.fixture
def estimator_threads(pkg):
    yield pkg.set_auto_estimator_threads
    pkg.set_auto_estimator_threads(1)


@pytest.mark.parametrize("n", [1, 2, 3])
def test_parallel_estimator_makes_the_same_choice(lib, pkg, oracle, estimator_threads, n):
    """threads > 1: every distinct section once, concurrently -- same settings and bytes as the reference's sequence
    (oracle_auto), with an estimator that sees every byte (CRC) and with zlib."""
    fmt = FMT[n]
    out = CORE_S[n]()
    rng = np.random.default_rng(40 + n)
    for blocks in (1, 37, 4099):
        x = rng.integers(0, 256, blocks * BLOCK[fmt], dtype=np.uint8)
        for use_all in (False, True):
            for kind in ("crc", "zlib"):
                est, py_est = cabi.make_estimator(kind)
                want_choice, want_out, _ = oracle_auto.transform_auto(fmt, x, lambda b: py_est(bytes(b)), use_all)
                for threads in (1, 4):
                    estimator_threads(threads)
                    assert pkg.get_auto_estimator_threads() == threads
                    y = np.zeros_like(x)
                    r = getattr(lib, f"dltbc{n}core_transform_auto")(x.ctypes.data, x.size, y.ctypes.data, y.size, C.byref(est),
                                                                     cabi.AutoSettings(use_all), C.byref(out))
                    assert r.ErrorCode == 0, (fmt, blocks, use_all, kind, threads)
                    assert np.array_equal(y, want_out), (fmt, blocks, use_all, kind, threads)


@pytest.mark.parametrize("n", [1, 2, 3])
def test_parallel_estimator_calls_once_per_distinct_section_and_concurrently(lib, pkg, estimator_threads, n):
    made = cabi.zstd_c_estimator(1)
    if made is None:
        pytest.skip("no gcc / libzstd for the C estimator")
    est, zlib_c = made
    fmt = FMT[n]
    out = CORE_S[n]()
    x = np.tile(payload(fmt), 512)                    # 16-32 MiB: estimator calls of milliseconds, long enough to overlap
    results = {}
    for use_all in (False, True):
        for threads in (1, 6):
            estimator_threads(threads)
            zlib_c.zest_reset()
            y = np.zeros_like(x)
            r = getattr(lib, f"dltbc{n}core_transform_auto")(x.ctypes.data, x.size, y.ctypes.data, y.size, C.byref(est),
                                                             cabi.AutoSettings(use_all), C.byref(out))
            assert r.ErrorCode == 0
            results[(use_all, threads)] = (y.copy(), zlib_c.zest_calls(), zlib_c.zest_max_concurrency())
        seq, par = results[(use_all, 1)], results[(use_all, 6)]
        assert np.array_equal(seq[0], par[0]), (fmt, use_all)          # same choice, same bytes
        candidates = (8 if use_all else 4) if n != 3 else (16 if use_all else 8)
        assert seq[1] == candidates * (2 if n == 3 else 1) and seq[2] == 1
        assert par[1] == (8 if use_all else 4) + (2 if n == 3 else 0)  # distinct sections only
        # and they did overlap (threads start tens of microseconds apart, every call takes milliseconds); a loaded box
        # gets two more tries before this is called a failure
        overlap = par[2]
        for _ in range(2):
            if overlap > 1:
                break
            estimator_threads(6)
            zlib_c.zest_reset()
            y = np.zeros_like(x)
            r = getattr(lib, f"dltbc{n}core_transform_auto")(x.ctypes.data, x.size, y.ctypes.data, y.size, C.byref(est),
                                                             cabi.AutoSettings(use_all), C.byref(out))
            assert r.ErrorCode == 0
            overlap = zlib_c.zest_max_concurrency()
        assert overlap > 1


def test_per_thread_cap_keeps_the_estimator_on_the_calling_thread(lib, pkg, estimator_threads):
    """dxtlt_set_auto_estimator_threads_for_this_thread(1): what the Rust glue holds around its auto calls, because its
    estimator type is not `Sync` (rust/core-bodies/gfx950_glue.rs, SerialEstimatorCalls).  With the process-wide setting at 6
    the capped thread's callbacks must still all run on that thread, one at a time, in the reference's sequence of calls; the
    cap is per thread (another thread is not capped) and the setter returns the previous cap."""
    import threading

    lib.dxtlt_set_auto_estimator_threads_for_this_thread.argtypes = [C.c_int32]
    lib.dxtlt_set_auto_estimator_threads_for_this_thread.restype = C.c_int32
    n, fmt = 3, "bc3"
    out = CORE_S[n]()
    x = np.tile(payload(fmt), 64)
    seen = []

    def record(_ctx, _p, _n, _o, _ol, out_size):
        seen.append(threading.get_ident())
        out_size[0] = 100 + len(seen)
        return 0

    def max_size(_ctx, _n, out_size):
        out_size[0] = 0
        return 0

    est = cabi.DltSizeEstimator(None, cabi.MAXFN(max_size), cabi.ESTFN(record))
    estimator_threads(6)
    try:
        assert lib.dxtlt_set_auto_estimator_threads_for_this_thread(1) == 0
        y = np.zeros_like(x)
        r = lib.dltbc3core_transform_auto(x.ctypes.data, x.size, y.ctypes.data, y.size, C.byref(est), cabi.AutoSettings(True), C.byref(out))
        assert r.ErrorCode == 0
        assert set(seen) == {threading.get_ident()} and len(seen) == 16 * 2     # the reference's sequence: every candidate, both sections
        other = {}
        t = threading.Thread(target=lambda: other.setdefault("cap", lib.dxtlt_set_auto_estimator_threads_for_this_thread(0)))
        t.start(); t.join()
        assert other["cap"] == 0                                               # a fresh thread has no cap
        assert lib.dxtlt_set_auto_estimator_threads_for_this_thread(0) == 1
        del seen[:]
        r = lib.dltbc3core_transform_auto(x.ctypes.data, x.size, y.ctypes.data, y.size, C.byref(est), cabi.AutoSettings(True), C.byref(out))
        assert r.ErrorCode == 0 and len(seen) == 8 + 2                         # uncapped again: distinct sections only, on worker threads
    finally:
        lib.dxtlt_set_auto_estimator_threads_for_this_thread(0)


def test_parallel_estimator_reports_estimator_failures(lib, pkg, estimator_threads):
    estimator_threads(4)
    for n in (1, 3):
        x = payload(FMT[n])
        y = np.zeros_like(x)
        out = CORE_S[n]()
        est, _ = cabi.make_estimator("fail_est")
        r = getattr(lib, f"dltbc{n}core_transform_auto")(x.ctypes.data, x.size, y.ctypes.data, y.size, C.byref(est),
                                                         cabi.AutoSettings(False), C.byref(out))
        assert r.ErrorCode == 7  # SizeEstimationError
