"""BC7 mode-split transform, version 0 -- a format defined by this build (docs/BC7_FORMAT.md); the reference has no BC7
transform.  Thin Python layer over include/dxtlt_bc7.h, same buffer conventions as the BC1-3 functions."""
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

        need = workspace_bytes(src.nbytes)
        with torch.cuda.device(src.device):
            if workspace is None:
                workspace = torch.empty(max(need, 16), dtype=torch.uint8, device=input.device)
            ws = _Buf(workspace, True)
            stream = torch.cuda.current_stream().cuda_stream
            rc = getattr(l, name + "_device")(src.ptr, dst.ptr, src.nbytes, ws.ptr, ws.nbytes, stream)
    if rc != _lib.OK:
        raise DeviceError(rc, _lib.last_error())


def transform_bc7(input, output, workspace=None) -> None:
    _run(False, input, output, workspace)


def untransform_bc7(input, output, workspace=None) -> None:
    _run(True, input, output, workspace)


def _declare_sharded(l):
    if not getattr(l, "_bc7_sharded_declared", False):
        vp, sz, i32, u64p = C.c_void_p, C.c_size_t, C.c_int32, C.POINTER(C.c_uint64)
        for n in ("dxtlt_transform_bc7_sharded", "dxtlt_untransform_bc7_sharded"):
            getattr(l, n).argtypes, getattr(l, n).restype = [vp, vp, sz, i32], i32
        l.dxtlt_bc7_shard_pieces.argtypes = [u64p, i32, i32, u64p, u64p, C.c_uint64, u64p, u64p, u64p]
        l.dxtlt_bc7_shard_pieces.restype = i32
        l._bc7_sharded_declared = True
    return l


def transform_bc7_sharded(input, output, num_shards: int = 0, inverse: bool = False) -> None:
    """Host buffers, block range sharded over the node's GPUs inside this process (no collective).  ``num_shards`` <= 0:
    one shard per device; more shards than devices run round robin."""
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


def shard_pieces(counts, shard: int, first_blocks, num_blocks, total_blocks: int):
    """Placement table of one shard (pure host code): three lists of 19 ints (global offset, local offset, bytes) for
    the pieces `first`, head_0..8, tail_0..8.  ``counts``: per shard, nine per-mode block counts."""
    from . import DeviceError

    l = _declare_sharded(_l())
    s = len(counts)
    flat = (C.c_uint64 * (9 * s))(*[int(c) for row in counts for c in row])
    fb = (C.c_uint64 * s)(*[int(v) for v in first_blocks])
    nb = (C.c_uint64 * s)(*[int(v) for v in num_blocks])
    g, lo, n = (C.c_uint64 * 19)(), (C.c_uint64 * 19)(), (C.c_uint64 * 19)()
    rc = l.dxtlt_bc7_shard_pieces(flat, s, int(shard), fb, nb, int(total_blocks), g, lo, n)
    if rc != _lib.OK:
        raise DeviceError(rc, _lib.last_error())
    return list(g), list(lo), list(n)
