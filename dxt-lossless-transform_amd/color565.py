"""Array-level RGB565 colour operations -- host-side mirror of the reference's common crate
(``Color565::decorrelate_ycocg_r_ptr`` / ``recorrelate_ycocg_r_ptr`` decorrelate_batch_ptr.rs:336,378,
``recorrelate_ycocg_r_ptr_split`` decorrelate_batch_split_ptr.rs:324, ``split_color_endpoints``
split_565_color_endpoints/mod.rs:110) over include/dxtlt_color565.h.

Buffers are the little-endian bytes of the colours: 1-D ``uint8`` numpy / bytes-like host buffers or CUDA ``torch.uint8``
tensors (enqueued on torch's current stream).  No CPU fallback."""
from __future__ import annotations

import ctypes as C

from . import _lib

_declared = False


def _l():
    global _declared
    l = _lib.load()
    if not _declared:
        vp, sz, i32, u8 = C.c_void_p, C.c_size_t, C.c_int32, C.c_uint8
        sig = {
            "dxtlt_color565_decorrelate_ycocg_r": [vp, vp, sz, u8],
            "dxtlt_color565_recorrelate_ycocg_r": [vp, vp, sz, u8],
            "dxtlt_color565_recorrelate_ycocg_r_split": [vp, vp, vp, sz, u8],
            "dxtlt_split_565_color_endpoints": [vp, vp, sz],
            "dxtlt_color565_decorrelate_ycocg_r_device": [vp, vp, sz, u8, vp],
            "dxtlt_color565_recorrelate_ycocg_r_device": [vp, vp, sz, u8, vp],
            "dxtlt_color565_recorrelate_ycocg_r_split_device": [vp, vp, vp, sz, u8, vp],
            "dxtlt_split_565_color_endpoints_device": [vp, vp, sz, vp],
        }
        for name, args in sig.items():
            getattr(l, name).argtypes, getattr(l, name).restype = args, i32
        _declared = True
    return l


def _check(rc: int) -> None:
    from . import DeviceError

    if rc != _lib.OK:
        raise DeviceError(rc, _lib.last_error())


def _bufs(items, writable):
    from . import _Buf

    bufs = [_Buf(x, w) for x, w in zip(items, writable)]
    if len({b.device for b in bufs}) != 1:
        raise TypeError("all buffers must be host buffers or all be tensors on one device")
    return bufs, bufs[0].device


def _run(name, device, *args):
    l = _l()
    if device is None:
        _check(getattr(l, name)(*args))
        return
    import torch

    with torch.cuda.device(device):
        _check(getattr(l, name + "_device")(*args, torch.cuda.current_stream(device).cuda_stream))


def _ycocg(name, src, dst, variant):
    from . import InvalidLength, OutputBufferTooSmall

    (s, d), device = _bufs((src, dst), (False, True))
    if s.nbytes % 2 != 0:
        raise InvalidLength(s.nbytes)
    if d.nbytes < s.nbytes:
        raise OutputBufferTooSmall(s.nbytes, d.nbytes)
    _run(name, device, s.ptr, d.ptr, s.nbytes // 2, int(variant))


def decorrelate_ycocg_r(src, dst, variant) -> None:
    """decorrelate_batch_ptr.rs:336; ``dst`` may be ``src``.  ``variant`` = YCoCgVariant (0 = None: a copy)."""
    _ycocg("dxtlt_color565_decorrelate_ycocg_r", src, dst, variant)


def recorrelate_ycocg_r(src, dst, variant) -> None:
    """decorrelate_batch_ptr.rs:378; ``dst`` may be ``src``."""
    _ycocg("dxtlt_color565_recorrelate_ycocg_r", src, dst, variant)


def recorrelate_ycocg_r_split(src0, src1, dst, variant) -> None:
    """decorrelate_batch_split_ptr.rs:324: dst[2k] = recorrelate(src0[k]), dst[2k+1] = recorrelate(src1[k])."""
    from . import InvalidLength, OutputBufferTooSmall

    (a, b, d), device = _bufs((src0, src1, dst), (False, False, True))
    if a.nbytes % 2 != 0 or a.nbytes != b.nbytes:
        raise InvalidLength(a.nbytes)
    if d.nbytes < 2 * a.nbytes:
        raise OutputBufferTooSmall(2 * a.nbytes, d.nbytes)
    _run("dxtlt_color565_recorrelate_ycocg_r_split", device, a.ptr, b.ptr, d.ptr, a.nbytes, int(variant))


def split_color_endpoints(colors, colors_out) -> None:
    """split_565_color_endpoints/mod.rs:110: (c0, c1) pairs -> all c0, then all c1."""
    from . import InvalidLength, OutputBufferTooSmall

    (s, d), device = _bufs((colors, colors_out), (False, True))
    if s.nbytes % 4 != 0:
        raise InvalidLength(s.nbytes)
    if d.nbytes < s.nbytes:
        raise OutputBufferTooSmall(s.nbytes, d.nbytes)
    _run("dxtlt_split_565_color_endpoints", device, s.ptr, d.ptr, s.nbytes)
