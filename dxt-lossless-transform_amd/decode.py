"""BC1 / BC2 / BC3 block decoders -- host-side mirror of the reference's util modules (``decode_bcN_block``:
bc1_decode.rs:42, bc2_decode.rs:44, bc3_decode.rs:43) as array operations over include/dxtlt_decode.h.

``decode_blocks`` writes one ``Decoded4x4Block`` (64 bytes: sixteen r, g, b, a pixels, row-major) per block;
``count_pixel_differences`` is the reference tests' "decode before == decode after" assertion as a count.
1-D ``uint8`` numpy / bytes-like host buffers or CUDA ``torch.uint8`` tensors (torch's current stream).  No CPU fallback."""
from __future__ import annotations

import ctypes as C

from . import _lib

DECODED_BLOCK_BYTES = 64
_FMT = {"bc1": 1, "bc2": 2, "bc3": 3}
_BLOCK = {"bc1": 8, "bc2": 16, "bc3": 16}
_declared = False


def _l():
    global _declared
    l = _lib.load()
    if not _declared:
        vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int32
        for f in _FMT:
            getattr(l, f"dxtlt_decode_{f}_blocks").argtypes = [vp, sz, vp, sz]
            getattr(l, f"dxtlt_decode_{f}_blocks_device").argtypes = [vp, sz, vp, sz, vp]
            getattr(l, f"dxtlt_decode_{f}_blocks").restype = getattr(l, f"dxtlt_decode_{f}_blocks_device").restype = i32
        l.dxtlt_count_pixel_differences.argtypes = [i32, vp, vp, sz, C.POINTER(C.c_uint64)]
        l.dxtlt_count_pixel_differences_device.argtypes = [i32, vp, vp, sz, vp, vp]
        l.dxtlt_count_pixel_differences.restype = l.dxtlt_count_pixel_differences_device.restype = i32
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


def decode_blocks(fmt: str, blocks, pixels) -> None:
    """``pixels`` receives 64 bytes per block (decoded_4x4_block.rs:56)."""
    from . import InvalidLength, OutputBufferTooSmall

    (s, d), device = _bufs((blocks, pixels), (False, True))
    if s.nbytes % _BLOCK[fmt] != 0:
        raise InvalidLength(s.nbytes)
    need = s.nbytes // _BLOCK[fmt] * DECODED_BLOCK_BYTES
    if d.nbytes < need:
        raise OutputBufferTooSmall(need, d.nbytes)
    l = _l()
    if device is None:
        _check(getattr(l, f"dxtlt_decode_{fmt}_blocks")(s.ptr, s.nbytes, d.ptr, d.nbytes))
        return
    import torch

    with torch.cuda.device(device):
        _check(getattr(l, f"dxtlt_decode_{fmt}_blocks_device")(s.ptr, s.nbytes, d.ptr, d.nbytes,
                                                               torch.cuda.current_stream(device).cuda_stream))


def count_pixel_differences(fmt: str, blocks_a, blocks_b) -> int:
    """Number of blocks whose sixteen decoded pixels differ between two block arrays of equal length."""
    from . import InvalidLength

    (a, b), device = _bufs((blocks_a, blocks_b), (False, False))
    if a.nbytes % _BLOCK[fmt] != 0 or a.nbytes != b.nbytes:
        raise InvalidLength(a.nbytes)
    l = _l()
    if device is None:
        out = C.c_uint64(0)
        _check(l.dxtlt_count_pixel_differences(_FMT[fmt], a.ptr, b.ptr, a.nbytes, C.byref(out)))
        return int(out.value)
    import torch

    with torch.cuda.device(device):
        count = torch.zeros(1, dtype=torch.int64, device=f"cuda:{device}")
        _check(l.dxtlt_count_pixel_differences_device(_FMT[fmt], a.ptr, b.ptr, a.nbytes, count.data_ptr(),
                                                      torch.cuda.current_stream(device).cuda_stream))
        return int(count.item())
