"""BC1 block normalisation -- host-side mirror of the reference's experimental module
(``dxt_lossless_transform_bc1::experimental::normalize_blocks``, normalize.rs / transform.rs / mod.rs) over the C ABI of
include/dxtlt_bc1_normalize.h.  Same buffer conventions as the BC1-3 functions: numpy/bytes-like host buffers or CUDA
``torch.uint8`` tensors (device path, enqueued on torch's current stream).  No CPU fallback."""
from __future__ import annotations

import ctypes as C
import dataclasses
import enum
from typing import Iterator, Sequence

from . import _lib

_declared = False


class ColorNormalizationMode(enum.IntEnum):
    """normalize.rs:487-500"""

    NONE = 0
    COLOR0_ONLY = 1
    REPLICATE_COLOR = 2


@dataclasses.dataclass(frozen=True)
class Bc1TransformDetailsWithNormalization:
    """experimental/normalize_blocks/mod.rs:98-113; defaults :126-135"""

    color_normalization_mode: ColorNormalizationMode = ColorNormalizationMode.NONE
    decorrelation_mode: int = 1  # core YCoCgVariant numbering, Variant1
    split_colour_endpoints: bool = True

    @staticmethod
    def all_combinations() -> Iterator["Bc1TransformDetailsWithNormalization"]:
        # mod.rs:170-184: mode-major, then decorrelation (None, Variant1..3), then split true / false
        for m in ColorNormalizationMode:
            for v in (0, 1, 2, 3):
                for s in (True, False):
                    yield Bc1TransformDetailsWithNormalization(m, v, s)

    def untransform_settings(self):
        """`impl From<Bc1TransformDetailsWithNormalization> for Bc1UntransformSettings` (mod.rs:115-122)."""
        from . import Bc1TransformSettings, YCoCgVariant

        return Bc1TransformSettings(YCoCgVariant(self.decorrelation_mode), self.split_colour_endpoints)


def _l():
    global _declared
    l = _lib.load()
    if not _declared:
        vp, sz, i32, u8, b = C.c_void_p, C.c_size_t, C.c_int32, C.c_uint8, C.c_bool
        l.dxtlt_bc1_normalize_blocks.argtypes, l.dxtlt_bc1_normalize_blocks.restype = [vp, vp, sz, u8], i32
        l.dxtlt_bc1_normalize_blocks_device.argtypes = [vp, vp, sz, u8, vp]
        l.dxtlt_bc1_normalize_blocks_device.restype = i32
        l.dxtlt_bc1_normalize_split_blocks_in_place.argtypes = [vp, vp, sz, u8]
        l.dxtlt_bc1_normalize_split_blocks_in_place.restype = i32
        l.dxtlt_bc1_normalize_split_blocks_in_place_device.argtypes = [vp, vp, sz, u8, vp]
        l.dxtlt_bc1_normalize_split_blocks_in_place_device.restype = i32
        l.dxtlt_bc1_normalize_blocks_all_modes.argtypes = [vp, C.POINTER(vp), sz, C.POINTER(b)]
        l.dxtlt_bc1_normalize_blocks_all_modes.restype = i32
        l.dxtlt_bc1_normalize_blocks_all_modes_device.argtypes = [vp, C.POINTER(vp), sz, vp, vp]
        l.dxtlt_bc1_normalize_blocks_all_modes_device.restype = i32
        l.dxtlt_transform_bc1_with_normalize_blocks.argtypes = [vp, vp, vp, sz, u8, u8, b]
        l.dxtlt_transform_bc1_with_normalize_blocks.restype = i32
        l.dxtlt_transform_bc1_with_normalize_blocks_device.argtypes = [vp, vp, sz, u8, u8, b, vp]
        l.dxtlt_transform_bc1_with_normalize_blocks_device.restype = i32
        _declared = True
    return l


def _check(rc: int) -> None:
    from . import DeviceError

    if rc != _lib.OK:
        raise DeviceError(rc, _lib.last_error())


def _stream(device) -> int:
    import torch

    return torch.cuda.current_stream(device).cuda_stream


def normalize_blocks(input, output, color_mode: ColorNormalizationMode) -> None:
    """normalize.rs:38.  ``output`` may be the same buffer as ``input`` (in place)."""
    from . import InvalidLength, OutputBufferTooSmall, _Buf

    src, dst = _Buf(input, False), _Buf(output, True)
    if src.nbytes % 8 != 0:
        raise InvalidLength(src.nbytes)
    if dst.nbytes < src.nbytes:
        raise OutputBufferTooSmall(src.nbytes, dst.nbytes)
    if (src.device is None) != (dst.device is None):
        raise TypeError("input and output must both be host buffers or both be device tensors")
    if src.device is None:
        _check(_l().dxtlt_bc1_normalize_blocks(src.ptr, dst.ptr, src.nbytes, int(color_mode)))
    else:
        import torch

        with torch.cuda.device(src.device):
            _check(_l().dxtlt_bc1_normalize_blocks_device(src.ptr, dst.ptr, src.nbytes, int(color_mode),
                                                          _stream(src.device)))


def normalize_split_blocks_in_place(colors, indices, color_mode: ColorNormalizationMode) -> None:
    """normalize.rs:286: colours (4 bytes per block) and indices (4 bytes per block), both modified in place."""
    from . import InvalidLength, _Buf

    c, x = _Buf(colors, True), _Buf(indices, True)
    if c.nbytes % 4 != 0 or c.nbytes != x.nbytes:
        raise InvalidLength(c.nbytes)
    if (c.device is None) != (x.device is None):
        raise TypeError("colors and indices must both be host buffers or both be device tensors")
    if c.device is None:
        _check(_l().dxtlt_bc1_normalize_split_blocks_in_place(c.ptr, x.ptr, c.nbytes // 4, int(color_mode)))
    else:
        import torch

        with torch.cuda.device(c.device):
            _check(_l().dxtlt_bc1_normalize_split_blocks_in_place_device(c.ptr, x.ptr, c.nbytes // 4, int(color_mode),
                                                                         _stream(c.device)))


def normalize_blocks_all_modes(input, outputs: Sequence) -> bool:
    """normalize.rs:417: one output per ColorNormalizationMode, in enum order.  Returns whether any block was
    normalised.  Device tensors: synchronises the current stream to read the flag."""
    from . import InvalidLength, OutputBufferTooSmall, _Buf

    if len(outputs) != len(ColorNormalizationMode):
        raise ValueError("one output buffer per ColorNormalizationMode is required")
    src = _Buf(input, False)
    outs = [_Buf(o, True) for o in outputs]
    if src.nbytes % 8 != 0:
        raise InvalidLength(src.nbytes)
    for o in outs:
        if o.nbytes < src.nbytes:
            raise OutputBufferTooSmall(src.nbytes, o.nbytes)
        if (o.device is None) != (src.device is None):
            raise TypeError("input and outputs must all be host buffers or all be device tensors")
    ptrs = (C.c_void_p * 3)(*[o.ptr for o in outs])
    if src.device is None:
        flag = C.c_bool(False)
        _check(_l().dxtlt_bc1_normalize_blocks_all_modes(src.ptr, ptrs, src.nbytes, C.byref(flag)))
        return bool(flag.value)
    import torch

    with torch.cuda.device(src.device):
        any_word = torch.zeros(1, dtype=torch.int32, device=input.device)
        _check(_l().dxtlt_bc1_normalize_blocks_all_modes_device(src.ptr, ptrs, src.nbytes, any_word.data_ptr(),
                                                                _stream(src.device)))
        return bool(any_word.item() != 0)


def transform_bc1_with_normalize_blocks(input, output,
                                        details: Bc1TransformDetailsWithNormalization = Bc1TransformDetailsWithNormalization(),
                                        ) -> None:
    """transform.rs:65 (the reference's ``work_ptr`` scratch buffer has no counterpart: normalisation is fused into
    the transform kernel).  Undo with ``untransform_bc1_with_settings(..., details.untransform_settings())``."""
    from . import InvalidLength, OutputBufferTooSmall, _Buf

    src, dst = _Buf(input, False), _Buf(output, True)
    if src.nbytes % 8 != 0:
        raise InvalidLength(src.nbytes)
    if dst.nbytes < src.nbytes:
        raise OutputBufferTooSmall(src.nbytes, dst.nbytes)
    if (src.device is None) != (dst.device is None):
        raise TypeError("input and output must both be host buffers or both be device tensors")
    args = (int(details.color_normalization_mode), int(details.decorrelation_mode), bool(details.split_colour_endpoints))
    if src.device is None:
        _check(_l().dxtlt_transform_bc1_with_normalize_blocks(src.ptr, dst.ptr, None, src.nbytes, *args))
    else:
        import torch

        with torch.cuda.device(src.device):
            _check(_l().dxtlt_transform_bc1_with_normalize_blocks_device(src.ptr, dst.ptr, src.nbytes, *args,
                                                                         _stream(src.device)))
