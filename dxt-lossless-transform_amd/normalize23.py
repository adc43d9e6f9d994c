"""BC2 / BC3 block normalisation -- host-side mirror of the reference's experimental modules
(``dxt_lossless_transform_bc{2,3}::experimental::normalize_blocks``) over include/dxtlt_bc23_normalize.h.  numpy /
bytes-like host buffers or CUDA ``torch.uint8`` tensors (enqueued on torch's current stream).  No CPU fallback."""
from __future__ import annotations

import ctypes as C
import enum
from typing import Sequence

from . import _lib
from .normalize import ColorNormalizationMode  # same three values in all formats


class AlphaNormalizationMode(enum.IntEnum):
    """bc3 normalize.rs:117-139"""

    NONE = 0
    UNIFORM_ALPHA_ZERO_INDICES = 1
    OPAQUE_FILL_ALL = 2
    OPAQUE_ZERO_ALPHA_MAX_INDICES = 3


_declared = False


def _l():
    global _declared
    l = _lib.load()
    if not _declared:
        vp, sz, i32, u8 = C.c_void_p, C.c_size_t, C.c_int32, C.c_uint8
        pp = C.POINTER(vp)
        sig = {
            "dxtlt_bc2_normalize_blocks": [vp, vp, sz, u8],
            "dxtlt_bc2_normalize_blocks_device": [vp, vp, sz, u8, vp],
            "dxtlt_bc3_normalize_blocks": [vp, vp, sz, u8, u8],
            "dxtlt_bc3_normalize_blocks_device": [vp, vp, sz, u8, u8, vp],
            "dxtlt_bc2_normalize_split_blocks_in_place": [vp, vp, vp, sz, u8],
            "dxtlt_bc2_normalize_split_blocks_in_place_device": [vp, vp, sz, u8, vp],
            "dxtlt_bc3_normalize_split_blocks_in_place": [vp, vp, vp, vp, sz, u8, u8],
            "dxtlt_bc3_normalize_split_blocks_in_place_device": [vp, vp, vp, vp, sz, u8, u8, vp],
            "dxtlt_bc2_normalize_blocks_all_modes": [vp, pp, sz],
            "dxtlt_bc2_normalize_blocks_all_modes_device": [vp, pp, sz, vp],
            "dxtlt_bc3_normalize_blocks_all_modes": [vp, pp, sz],
            "dxtlt_bc3_normalize_blocks_all_modes_device": [vp, pp, sz, vp],
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
    devices = {b.device for b in bufs}
    if len(devices) != 1:
        raise TypeError("all buffers must be host buffers or all be tensors on one device")
    return bufs, bufs[0].device


def _on_device(device, call):
    import torch

    with torch.cuda.device(device):
        _check(call(torch.cuda.current_stream(device).cuda_stream))


def normalize_blocks(fmt: str, input, output, color_mode: ColorNormalizationMode,
                     alpha_mode: AlphaNormalizationMode = AlphaNormalizationMode.NONE) -> None:
    """bc2 normalize.rs:35 / bc3 normalize.rs:36; ``output`` may be ``input`` (in place)."""
    from . import InvalidLength, OutputBufferTooSmall

    assert fmt in ("bc2", "bc3")
    (src, dst), device = _bufs((input, output), (False, True))
    if src.nbytes % 16 != 0:
        raise InvalidLength(src.nbytes)
    if dst.nbytes < src.nbytes:
        raise OutputBufferTooSmall(src.nbytes, dst.nbytes)
    l = _l()
    if fmt == "bc2":
        if device is None:
            _check(l.dxtlt_bc2_normalize_blocks(src.ptr, dst.ptr, src.nbytes, int(color_mode)))
        else:
            _on_device(device, lambda s: l.dxtlt_bc2_normalize_blocks_device(src.ptr, dst.ptr, src.nbytes, int(color_mode), s))
    else:
        if device is None:
            _check(l.dxtlt_bc3_normalize_blocks(src.ptr, dst.ptr, src.nbytes, int(alpha_mode), int(color_mode)))
        else:
            _on_device(device, lambda s: l.dxtlt_bc3_normalize_blocks_device(src.ptr, dst.ptr, src.nbytes, int(alpha_mode),
                                                                           int(color_mode), s))


def normalize_blocks_all_modes(fmt: str, input, outputs: Sequence) -> None:
    """bc2 normalize.rs:193 (3 outputs, by colour mode) / bc3 normalize.rs:419 (12 outputs, [alpha_mode * 3 + colour_mode])."""
    from . import InvalidLength, OutputBufferTooSmall

    assert fmt in ("bc2", "bc3")
    count = 3 if fmt == "bc2" else 12
    if len(outputs) != count:
        raise ValueError(f"{count} output buffers are required")
    bufs, device = _bufs((input, *outputs), (False,) + (True,) * count)
    src, outs = bufs[0], bufs[1:]
    if src.nbytes % 16 != 0:
        raise InvalidLength(src.nbytes)
    for o in outs:
        if o.nbytes < src.nbytes:
            raise OutputBufferTooSmall(src.nbytes, o.nbytes)
    ptrs = (C.c_void_p * count)(*[o.ptr for o in outs])
    l = _l()
    name = f"dxtlt_{fmt}_normalize_blocks_all_modes"
    if device is None:
        _check(getattr(l, name)(src.ptr, ptrs, src.nbytes))
    else:
        _on_device(device, lambda s: getattr(l, name + "_device")(src.ptr, ptrs, src.nbytes, s))


def normalize_bc2_split_blocks_in_place(alpha, colors, indices, color_mode: ColorNormalizationMode) -> None:
    """bc2 normalize.rs:382.  ``alpha`` (8 bytes per block) is accepted for signature parity and may be None."""
    from . import InvalidLength

    (c, x), device = _bufs((colors, indices), (True, True))
    if c.nbytes % 4 != 0 or c.nbytes != x.nbytes:
        raise InvalidLength(c.nbytes)
    l = _l()
    if device is None:
        _check(l.dxtlt_bc2_normalize_split_blocks_in_place(None, c.ptr, x.ptr, c.nbytes // 4, int(color_mode)))
    else:
        _on_device(device, lambda s: l.dxtlt_bc2_normalize_split_blocks_in_place_device(c.ptr, x.ptr, c.nbytes // 4,
                                                                                      int(color_mode), s))


def normalize_bc3_split_blocks_in_place(alpha_endpoints, alpha_indices, color_endpoints, color_indices,
                                        alpha_mode: AlphaNormalizationMode, color_mode: ColorNormalizationMode) -> None:
    """bc3 normalize.rs:539: sections of 2, 6, 4 and 4 bytes per block, all modified in place."""
    from . import InvalidLength

    (ae, ai, ce, ci), device = _bufs((alpha_endpoints, alpha_indices, color_endpoints, color_indices), (True,) * 4)
    n = ae.nbytes // 2
    if ae.nbytes != 2 * n or ai.nbytes != 6 * n or ce.nbytes != 4 * n or ci.nbytes != 4 * n:
        raise InvalidLength(ae.nbytes)
    l = _l()
    if device is None:
        _check(l.dxtlt_bc3_normalize_split_blocks_in_place(ae.ptr, ai.ptr, ce.ptr, ci.ptr, n, int(alpha_mode), int(color_mode)))
    else:
        _on_device(device, lambda s: l.dxtlt_bc3_normalize_split_blocks_in_place_device(
            ae.ptr, ai.ptr, ce.ptr, ci.ptr, n, int(alpha_mode), int(color_mode), s))
