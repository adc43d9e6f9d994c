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
