"""Many device-resident buffers in one call (dxtlt_transform_batch_device, include/dxtlt_gfx950.h): one kernel launch per
(format, direction) present in the batch, enqueued on torch's current stream.  fmt "bc7" (this build's own format,
docs/BC7_FORMAT.md; settings ignored, pass None) rides along: its granules in one launch, its tail parts in another."""
from __future__ import annotations

import ctypes as C
from typing import Sequence, Tuple

from . import _lib


def _item_fields(fmt, settings):
    """(format id, block bytes, mode, split_alpha, split_colour) of one batch item"""
    from . import _FMT_ID, BLOCK_BYTES, _settings_tuple

    if fmt == "bc7":
        return 7, 16, 0, False, False
    mode, sa, sc = _settings_tuple(fmt, settings)
    return _FMT_ID[fmt], BLOCK_BYTES[fmt], mode, sa, sc


class DxtltBatchItem(C.Structure):
    _fields_ = [("d_input", C.c_void_p), ("d_output", C.c_void_p), ("len", C.c_uint64), ("format", C.c_uint8),
                ("inverse", C.c_uint8), ("decorrelation_mode", C.c_uint8), ("split_alpha_endpoints", C.c_uint8),
                ("split_colour_endpoints", C.c_uint8), ("reserved", C.c_uint8 * 3)]


def prepare_batch(items: Sequence[Tuple[str, bool, object, object, object]]):
    """The C item array of a device batch (and the tensors it points into), for callers that run the same batch again:
    building it costs ~2 us per item in Python, the call itself 0.1 us per item."""
    from . import InvalidLength, OutputBufferTooSmall, _Buf

    arr = (DxtltBatchItem * len(items))()
    device, keep = None, []
    for k, (fmt, inverse, src, dst, settings) in enumerate(items):
        s, d = _Buf(src, False), _Buf(dst, True)
        if s.device is None or d.device is None:
            raise TypeError("transform_batch takes device tensors")
        device = s.device if device is None else device
        if s.device != device or d.device != device:
            raise ValueError("all tensors of a batch must live on one device")
        fid, block, mode, sa, sc = _item_fields(fmt, settings)
        if s.nbytes % block != 0:
            raise InvalidLength(s.nbytes)
        if d.nbytes < s.nbytes:
            raise OutputBufferTooSmall(s.nbytes, d.nbytes)
        arr[k].d_input, arr[k].d_output, arr[k].len = s.ptr, d.ptr, s.nbytes
        arr[k].format, arr[k].inverse, arr[k].decorrelation_mode = fid, int(bool(inverse)), mode
        arr[k].split_alpha_endpoints, arr[k].split_colour_endpoints = int(bool(sa)), int(bool(sc))
        keep.append((s, d))
    return arr, device, keep


def run_prepared_batch(prepared) -> None:
    import torch

    from . import DeviceError

    arr, device, _keep = prepared
    if len(arr) == 0:
        return
    l = _lib.load()
    l.dxtlt_transform_batch_device.argtypes = [C.POINTER(DxtltBatchItem), C.c_size_t, C.c_void_p]
    l.dxtlt_transform_batch_device.restype = C.c_int32
    with torch.cuda.device(device):
        rc = l.dxtlt_transform_batch_device(arr, len(arr), torch.cuda.current_stream().cuda_stream)
    if rc != _lib.OK:
        raise DeviceError(rc, _lib.last_error())


def transform_batch(items: Sequence[Tuple[str, bool, object, object, object]]) -> None:
    """items: (fmt, inverse, input tensor, output tensor, settings) with CUDA uint8 tensors on one device."""
    if not items:
        return
    run_prepared_batch(prepare_batch(items))


def prepare_batch_host(items: Sequence[Tuple[str, bool, object, object, object]]):
    """The C item array of a host batch (and the buffers it points into), for callers that run the same batch again."""
    from . import InvalidLength, OutputBufferTooSmall, _Buf

    arr = (DxtltBatchItem * len(items))()
    keep = []
    for k, (fmt, inverse, src, dst, settings) in enumerate(items):
        s, d = _Buf(src, False), _Buf(dst, True)
        if s.device is not None or d.device is not None:
            raise TypeError("transform_batch_host takes host buffers")
        fid, block, mode, sa, sc = _item_fields(fmt, settings)
        if s.nbytes % block != 0:
            raise InvalidLength(s.nbytes)
        if d.nbytes < s.nbytes:
            raise OutputBufferTooSmall(s.nbytes, d.nbytes)
        arr[k].d_input, arr[k].d_output, arr[k].len = s.ptr, d.ptr, s.nbytes
        arr[k].format, arr[k].inverse, arr[k].decorrelation_mode = fid, int(bool(inverse)), mode
        arr[k].split_alpha_endpoints, arr[k].split_colour_endpoints = int(bool(sa)), int(bool(sc))
        keep.append((s, d))
    return arr, keep


def run_prepared_batch_host(prepared) -> None:
    from . import DeviceError

    arr, _keep = prepared
    l = _lib.load()
    l.dxtlt_transform_batch_host.argtypes = [C.POINTER(DxtltBatchItem), C.c_size_t]
    l.dxtlt_transform_batch_host.restype = C.c_int32
    rc = l.dxtlt_transform_batch_host(arr, len(arr))
    if rc != _lib.OK:
        raise DeviceError(rc, _lib.last_error())


def transform_batch_host(items: Sequence[Tuple[str, bool, object, object, object]]) -> None:
    """items: (fmt, inverse, input, output, settings) with HOST buffers (numpy uint8 / bytes / bytearray): packed side by
    side, one upload, one launch per (format, direction) and one download per ~64 MiB chunk, unpacked
    (dxtlt_transform_batch_host)."""
    if not items:
        return
    run_prepared_batch_host(prepare_batch_host(items))
