"""BC7 granule-sorted field split, version 2 -- a format defined by this build (docs/BC7_FORMAT.md); the reference has
no BC7 transform.  Thin Python layer over include/dxtlt_bc7.h, same buffer conventions as the BC1-3 functions."""
from __future__ import annotations

import ctypes as C

from . import _lib

_declared = False


def _l():
    global _declared
    l = _lib.load()
    if not _declared:
        vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int32
        for n in ("dxtlt_transform_bc7", "dxtlt_untransform_bc7"):
            getattr(l, n).argtypes, getattr(l, n).restype = [vp, vp, sz], i32
        for n in ("dxtlt_transform_bc7_device", "dxtlt_untransform_bc7_device"):
            getattr(l, n).argtypes, getattr(l, n).restype = [vp, vp, sz, vp, sz, vp], i32
        l.dxtlt_bc7_workspace_bytes.argtypes, l.dxtlt_bc7_workspace_bytes.restype = [sz], sz
        l.dxtlt_transform_bc7_range_device.argtypes = [C.c_bool, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, vp]
        l.dxtlt_transform_bc7_range_device.restype = i32
        l.dxtlt_bc7_sort_granule.argtypes, l.dxtlt_bc7_sort_granule.restype = [], C.c_uint32
        _declared = True
    return l


def workspace_bytes(nbytes: int) -> int:
    return int(_l().dxtlt_bc7_workspace_bytes(nbytes))


def _run(inverse: bool, input, output, workspace=None) -> None:
    from . import DeviceError, InvalidLength, OutputBufferTooSmall, _Buf

    src, dst = _Buf(input, False), _Buf(output, True)
    if src.nbytes % 16 != 0:
        raise InvalidLength(src.nbytes)
    if dst.nbytes < src.nbytes:
        raise OutputBufferTooSmall(src.nbytes, dst.nbytes)
    if (src.device is None) != (dst.device is None):
        raise TypeError("input and output must both be host buffers or both be device tensors")
    l = _l()
    name = "dxtlt_untransform_bc7" if inverse else "dxtlt_transform_bc7"
    if src.device is None:
        rc = getattr(l, name)(src.ptr, dst.ptr, src.nbytes)
    else:
        import torch

        with torch.cuda.device(src.device):
            stream = torch.cuda.current_stream().cuda_stream
            rc = getattr(l, name + "_device")(src.ptr, dst.ptr, src.nbytes, None, 0, stream)   # since version 1: no workspace
    if rc != _lib.OK:
        raise DeviceError(rc, _lib.last_error())


def transform_bc7(input, output, workspace=None) -> None:
    _run(False, input, output, workspace)


def untransform_bc7(input, output, workspace=None) -> None:
    _run(True, input, output, workspace)


def sort_granule() -> int:
    return int(_l().dxtlt_bc7_sort_granule())


def transform_bc7_range(inverse: bool, src, dst, total_blocks: int, first_block: int, num_blocks: int) -> None:
    """dxtlt_transform_bc7_range_device on torch CUDA tensors: the AoS-side tensor starts at block `first_block` (a
    multiple of the sort granule), the SoA-side tensor is the whole transformed buffer."""
    import torch

    from . import DeviceError, OutputBufferTooSmall, _Buf

    s, d = _Buf(src, False), _Buf(dst, True)
    if s.device is None or d.device is None:
        raise TypeError("transform_bc7_range takes device tensors")
    aos, soa = (d, s) if inverse else (s, d)
    if aos.nbytes < num_blocks * 16 or soa.nbytes < total_blocks * 16:
        raise OutputBufferTooSmall(max(num_blocks, total_blocks) * 16, min(aos.nbytes, soa.nbytes))
    with torch.cuda.device(s.device):
        stream = torch.cuda.current_stream().cuda_stream
        rc = _l().dxtlt_transform_bc7_range_device(bool(inverse), s.ptr, d.ptr, total_blocks, first_block, num_blocks, stream)
    if rc != _lib.OK:
        raise DeviceError(rc, _lib.last_error())


def _declare_sharded(l):
    if not getattr(l, "_bc7_sharded_declared", False):
        vp, sz, i32, u64, u64p = C.c_void_p, C.c_size_t, C.c_int32, C.c_uint64, C.POINTER(C.c_uint64)
        for n in ("dxtlt_transform_bc7_sharded", "dxtlt_untransform_bc7_sharded"):
            getattr(l, n).argtypes, getattr(l, n).restype = [vp, vp, sz, i32], i32
        l.dxtlt_bc7_shard_pieces.argtypes = [u64, u64, u64, u64p, u64p, u64p]
        l.dxtlt_bc7_shard_pieces.restype = i32
        l._bc7_sharded_declared = True
    return l


def transform_bc7_sharded(input, output, num_shards: int = 0, inverse: bool = False) -> None:
    """Host buffers, block range sharded over the node's GPUs inside this process (no collective, no counter
    exchange).  ``num_shards`` <= 0: one shard per device; more shards than devices run round robin."""
    from . import DeviceError, InvalidLength, OutputBufferTooSmall, _Buf

    src, dst = _Buf(input, False), _Buf(output, True)
    if src.device is not None or dst.device is not None:
        raise TypeError("transform_bc7_sharded takes host buffers")
    if src.nbytes % 16 != 0:
        raise InvalidLength(src.nbytes)
    if dst.nbytes < src.nbytes:
        raise OutputBufferTooSmall(src.nbytes, dst.nbytes)
    l = _declare_sharded(_l())
    name = "dxtlt_untransform_bc7_sharded" if inverse else "dxtlt_transform_bc7_sharded"
    rc = getattr(l, name)(src.ptr, dst.ptr, src.nbytes, int(num_shards))
    if rc != _lib.OK:
        raise DeviceError(rc, _lib.last_error())


def shard_pieces(total_blocks: int, first_block: int, num_blocks: int):
    """Placement of one granule-aligned shard (pure host code): three lists of 9 ints (global offset, local offset,
    bytes) -- its slice of the eight main streams and, for the shard that reaches the end, the tail part."""
    from . import DeviceError

    l = _declare_sharded(_l())
    g, lo, n = (C.c_uint64 * 9)(), (C.c_uint64 * 9)(), (C.c_uint64 * 9)()
    rc = l.dxtlt_bc7_shard_pieces(int(total_blocks), int(first_block), int(num_blocks), g, lo, n)
    if rc != _lib.OK:
        raise DeviceError(rc, _lib.last_error())
    return list(g), list(lo), list(n)
