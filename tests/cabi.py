"""ctypes view of the reference-shaped C APIs (include/dltbc{1,2,3}core.h, include/dltbc{1,2}.h,
include/dlt_size_estimator.h) for the tests."""
from __future__ import annotations

import ctypes as C
import zlib

MAXFN = C.CFUNCTYPE(C.c_uint32, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t))
ESTFN = C.CFUNCTYPE(C.c_uint32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t))


class DltSizeEstimator(C.Structure):
    _fields_ = [("Context", C.c_void_p), ("MaxCompressedSize", MAXFN), ("EstimateCompressedSize", ESTFN)]


class CoreSettings2(C.Structure):  # Dltbc1/2TransformSettings, core layout
    _fields_ = [("SplitColourEndpoints", C.c_bool), ("DecorrelationMode", C.c_uint8)]


class CoreSettings3(C.Structure):
    _fields_ = [("SplitAlphaEndpoints", C.c_bool), ("SplitColourEndpoints", C.c_bool), ("DecorrelationMode", C.c_uint8)]


class AutoSettings(C.Structure):
    _fields_ = [("UseAllModes", C.c_bool)]


class Result(C.Structure):
    _fields_ = [("ErrorCode", C.c_int32)]


def bind(lib):
    vp, sz = C.c_void_p, C.c_size_t
    for n, S in ((1, CoreSettings2), (2, CoreSettings2), (3, CoreSettings3)):
        for d in ("transform", "untransform"):
            f = getattr(lib, f"dltbc{n}core_{d}")
            f.argtypes, f.restype = [vp, sz, vp, sz, S], Result
        f = getattr(lib, f"dltbc{n}core_transform_auto")
        f.argtypes, f.restype = [vp, sz, vp, sz, C.POINTER(DltSizeEstimator), AutoSettings, C.POINTER(S)], Result
    for n in (1, 2, 3):
        p = f"dltbc{n}_"
        getattr(lib, p + "new_ManualTransformBuilder").argtypes = []
        getattr(lib, p + "new_ManualTransformBuilder").restype = vp
        getattr(lib, p + "free_ManualTransformBuilder").argtypes = [vp]
        getattr(lib, p + "free_ManualTransformBuilder").restype = None
        getattr(lib, p + "clone_ManualTransformBuilder").argtypes = [vp]
        getattr(lib, p + "clone_ManualTransformBuilder").restype = vp
        getattr(lib, p + "ManualTransformBuilder_SetDecorrelationMode").argtypes = [vp, C.c_uint8]
        getattr(lib, p + "ManualTransformBuilder_SetDecorrelationMode").restype = None
        getattr(lib, p + "ManualTransformBuilder_SetSplitColourEndpoints").argtypes = [vp, C.c_bool]
        getattr(lib, p + "ManualTransformBuilder_SetSplitColourEndpoints").restype = None
        getattr(lib, p + "ManualTransformBuilder_ResetToDefaults").argtypes = [vp]
        getattr(lib, p + "ManualTransformBuilder_ResetToDefaults").restype = None
        for d in ("Transform", "Untransform"):
            f = getattr(lib, p + "ManualTransformBuilder_" + d)
            f.argtypes, f.restype = [vp, sz, vp, sz, vp], Result
        getattr(lib, p + "new_AutoTransformBuilder").argtypes = [C.POINTER(DltSizeEstimator)]
        getattr(lib, p + "new_AutoTransformBuilder").restype = vp
        getattr(lib, p + "free_AutoTransformBuilder").argtypes = [vp]
        getattr(lib, p + "free_AutoTransformBuilder").restype = None
        f = getattr(lib, p + "AutoTransformBuilder_SetUseAllDecorrelationModes")
        f.argtypes, f.restype = [vp, C.c_bool], Result
        f = getattr(lib, p + "AutoTransformBuilder_Transform")
        f.argtypes, f.restype = [vp, vp, sz, vp, sz, C.POINTER(vp)], Result
        f = getattr(lib, p + "error_message")
        f.argtypes, f.restype = [C.c_int32], C.c_char_p
    lib.dltbc3_ManualTransformBuilder_SetSplitAlphaEndpoints.argtypes = [vp, C.c_bool]
    lib.dltbc3_ManualTransformBuilder_SetSplitAlphaEndpoints.restype = None
    return lib


def make_estimator(kind: str, log=None):
    """kind: 'dummy' (size = len, like the reference's C dummy estimator, bc1 c_api/transform_auto.rs:200-231),
    'zlib' (zlib level 1 size), 'zstd' (the system libzstd at level 1 through tools/zstd_ratio.py: what the reference's
    estimator crate does, extensions/compressors/dxt-lossless-transform-zstd/src/lib.rs:146-200, with zstd 1.5.7),
    'fail_max' / 'fail_est' (callback errors)."""
    if kind == "zstd":
        from tools import zstd_ratio

    def py_estimate(buf: bytes) -> int:
        if kind == "zlib":
            return len(zlib.compress(buf, 1))
        if kind == "zstd":
            return zstd_ratio.compressed_size(buf, 1) if buf else 0
        if kind == "crc":          # every byte of the section matters; the log records what the estimator was shown
            return zlib.crc32(buf) & 0xFFFFF
        return len(buf)

    @MAXFN
    def max_fn(ctx, n, out):
        if kind == "fail_max":
            return 41
        out[0] = n + 64 if kind in ("zlib", "crc", "zstd") else (0 if kind == "dummy0" else n)
        return 0

    @ESTFN
    def est_fn(ctx, inp, n, scratch, scratch_len, out):
        if kind == "fail_est":
            return 42
        data = C.string_at(inp, n) if n else b""
        if log is not None:
            log.append((n, zlib.crc32(data)) if kind == "crc" else n)
        out[0] = py_estimate(data)
        return 0

    est = DltSizeEstimator(None, max_fn, est_fn)
    est._keep = (max_fn, est_fn)
    return est, py_estimate


def zstd_c_estimator(level: int = 1):
    """(DltSizeEstimator, lib) over tests/cpp/zstd_estimator.c -- a thread-safe C estimator on the system libzstd that
    counts its calls and its highest concurrency; None when gcc or libzstd is missing."""
    import os
    import subprocess

    here = os.path.dirname(os.path.abspath(__file__))
    src, so = os.path.join(here, "cpp", "zstd_estimator.c"), os.path.join(here, "cpp", "libzest.so")
    try:
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", src, "-o", so, "-ldl"])
        lib = C.CDLL(so)
    except (OSError, subprocess.CalledProcessError):
        return None
    lib.zest_init.restype = C.c_int
    if lib.zest_init() != 0:
        return None
    max_fn = C.cast(lib.zest_max_compressed_size, MAXFN)
    est_fn = C.cast(lib.zest_len_estimate if level is None else lib.zest_estimate, ESTFN)   # level None: size = len, in C
    est = DltSizeEstimator(C.c_void_p(level or 0), max_fn, est_fn)
    est._keep = (max_fn, est_fn, lib)
    return est, lib
