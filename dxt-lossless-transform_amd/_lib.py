"""ctypes binding of libdxtlt_gfx950.so (the C ABI declared in include/dxtlt_gfx950.h).

There is no CPU fallback: if the library is missing this module raises, and if no HIP device is present
the library's entry points return DXTLT_E_NO_DEVICE, which the wrappers turn into DeviceError.
"""
from __future__ import annotations

import ctypes as C
import os

from . import _build

OK, E_INVALID_LENGTH, E_INVALID_ARGUMENT, E_NO_DEVICE, E_DEVICE = 0, 1, 2, 3, 4

_lib = None


class LibraryMissingError(RuntimeError):
    pass


def _declare(l: C.CDLL) -> None:
    vp, sz, u8, b, i32, u64 = C.c_void_p, C.c_size_t, C.c_uint8, C.c_bool, C.c_int32, C.c_uint64
    for n in ("bc1", "bc2"):
        for d in ("transform", "untransform"):
            f = getattr(l, f"dxtlt_{d}_{n}_with_settings")
            f.argtypes, f.restype = [vp, vp, sz, u8, b], i32
            f = getattr(l, f"dxtlt_{d}_{n}_with_settings_device")
            f.argtypes, f.restype = [vp, vp, sz, u8, b, vp], i32
    for d in ("transform", "untransform"):
        f = getattr(l, f"dxtlt_{d}_bc3_with_settings")
        f.argtypes, f.restype = [vp, vp, sz, u8, b, b], i32
        f = getattr(l, f"dxtlt_{d}_bc3_with_settings_device")
        f.argtypes, f.restype = [vp, vp, sz, u8, b, b, vp], i32
    l.dxtlt_transform_range_device.argtypes = [i32, b, vp, vp, u64, u64, u64, u8, b, b, vp]
    l.dxtlt_transform_range_device.restype = i32
    l.dxtlt_transform_sharded.argtypes = [i32, b, vp, vp, sz, u8, b, b, i32]
    l.dxtlt_transform_sharded.restype = i32
    l.dxtlt_fill_splitmix64_device.argtypes = [vp, sz, u64, u64, vp]
    l.dxtlt_fill_splitmix64_device.restype = i32
    l.dxtlt_last_error.argtypes, l.dxtlt_last_error.restype = [], C.c_char_p
    l.dxtlt_device_count.argtypes, l.dxtlt_device_count.restype = [], i32
    l.dxtlt_set_tuning.argtypes, l.dxtlt_set_tuning.restype = [i32, i32], None
    if hasattr(l, "dxtlt_tuning_mask"):     # (absent from the round-4 library, which same-box A/B runs load through DXTLT_LIB_PATH)
        l.dxtlt_tuning_mask.argtypes, l.dxtlt_tuning_mask.restype = [], i32
    l.dxtlt_version.argtypes, l.dxtlt_version.restype = [], C.c_char_p
    l.dxtlt_set_auto_estimator_threads.argtypes, l.dxtlt_set_auto_estimator_threads.restype = [i32], None
    l.dxtlt_get_auto_estimator_threads.argtypes, l.dxtlt_get_auto_estimator_threads.restype = [], i32


def lib_path() -> str:
    # DXTLT_LIB_PATH: another build of the same library (tools/asan_host_check.sh points it at the sanitizer build)
    return os.environ.get("DXTLT_LIB_PATH") or _build.LIB_PATH


def load() -> C.CDLL:
    """Load the in-tree shared library.  Raises LibraryMissingError when it has not been built."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise LibraryMissingError(
                f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
            )
        # If torch is (or will be) in the process, let its HIP runtime be the one both sides use:
        # the library's DT_NEEDED libamdhip64.so.7 resolves to an already-loaded image with that SONAME.
        try:
            import torch  # noqa: F401
        except Exception:  # pragma: no cover - torch is plumbing, not a requirement
            pass
        l = C.CDLL(path, mode=C.RTLD_GLOBAL)
        _declare(l)
        _lib = l
    return _lib


def last_error() -> str:
    return load().dxtlt_last_error().decode("utf-8", "replace")
