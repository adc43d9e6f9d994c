"""dxt-lossless-transform on MI355X (gfx950): Python view of the BCn block-transform hot path.

The names follow the reference's core crates (paths under /root/reference/src/core/):

  YCoCgVariant                     dxt-lossless-transform-common/src/color_565/decorrelate.rs:72-84
  Bc1TransformSettings             dxt-lossless-transform-bc1/src/transform/settings.rs:16-43
  Bc2TransformSettings             dxt-lossless-transform-bc2/src/transform/settings.rs:16
  Bc3TransformSettings             dxt-lossless-transform-bc3/src/transform/settings.rs:16-48
  transform_bcN_with_settings      .../safe/transform_with_settings.rs:88  (validated slice API)
  untransform_bcN_with_settings    .../safe/transform_with_settings.rs:192
  BcNValidationError               .../safe/transform_with_settings.rs:18-31

Everything here is a thin layer over the C ABI of libdxtlt_gfx950.so (include/dxtlt_gfx950.h).  Buffers may
be host buffers (bytes, bytearray, numpy uint8: staged H2D/D2H by the library) or CUDA/HIP torch uint8
tensors (device pointers, enqueued on torch's current stream, no copies).  There is no CPU implementation
in this package.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import enum
from typing import Iterator

import numpy as np

from . import _build, _lib
from ._lib import LibraryMissingError  # noqa: F401

__all__ = [
    "YCoCgVariant", "Bc1TransformSettings", "Bc2TransformSettings", "Bc3TransformSettings",
    "Bc1UntransformSettings", "Bc2UntransformSettings", "Bc3UntransformSettings",
    "ValidationError", "InvalidLength", "OutputBufferTooSmall", "DeviceError",
    "transform_bc1_with_settings", "untransform_bc1_with_settings",
    "transform_bc2_with_settings", "untransform_bc2_with_settings",
    "transform_bc3_with_settings", "untransform_bc3_with_settings",
    "transform_range", "transform_sharded", "fill_splitmix64", "build", "load", "set_tuning",
    "stream_table", "plan_shards", "BLOCK_BYTES",
]

BLOCK_BYTES = {"bc1": 8, "bc2": 16, "bc3": 16}
_FMT_ID = {"bc1": 1, "bc2": 2, "bc3": 3}


class YCoCgVariant(enum.IntEnum):
    """Core numbering (decorrelate.rs:72-84).  The stable API crates renumber it; see include/dltbc1.h."""
    NONE = 0
    Variant1 = 1
    Variant2 = 2
    Variant3 = 3


@dataclasses.dataclass(frozen=True)
class Bc1TransformSettings:
    decorrelation_mode: YCoCgVariant = YCoCgVariant.Variant1
    split_colour_endpoints: bool = True

    @staticmethod
    def all_combinations() -> Iterator["Bc1TransformSettings"]:
        """bc1 settings.rs:68 -- every variant x {true, false}."""
        for v in YCoCgVariant:
            for s in (True, False):
                yield Bc1TransformSettings(v, s)


@dataclasses.dataclass(frozen=True)
class Bc2TransformSettings:
    decorrelation_mode: YCoCgVariant = YCoCgVariant.Variant1
    split_colour_endpoints: bool = True

    @staticmethod
    def all_combinations() -> Iterator["Bc2TransformSettings"]:
        for v in YCoCgVariant:
            for s in (True, False):
                yield Bc2TransformSettings(v, s)


@dataclasses.dataclass(frozen=True)
class Bc3TransformSettings:
    decorrelation_mode: YCoCgVariant = YCoCgVariant.Variant1
    split_alpha_endpoints: bool = True
    split_colour_endpoints: bool = True

    @staticmethod
    def all_combinations() -> Iterator["Bc3TransformSettings"]:
        """bc3 settings.rs:74 -- 16 combinations."""
        for v in YCoCgVariant:
            for sa in (True, False):
                for sc in (True, False):
                    yield Bc3TransformSettings(v, sa, sc)


# bc1 settings.rs:33: the untransform settings are the same type
Bc1UntransformSettings = Bc1TransformSettings
Bc2UntransformSettings = Bc2TransformSettings
Bc3UntransformSettings = Bc3TransformSettings


class ValidationError(ValueError):
    """BcNValidationError"""


class InvalidLength(ValidationError):
    def __init__(self, length: int):
        super().__init__(f"invalid input length {length}: not a multiple of the block size")
        self.length = length


class OutputBufferTooSmall(ValidationError):
    def __init__(self, needed: int, actual: int):
        super().__init__(f"output buffer too small: needed {needed}, actual {actual}")
        self.needed, self.actual = needed, actual


class DeviceError(RuntimeError):
    """The HIP side failed (no device, allocation, copy or launch).  Never swallowed, never retried on CPU."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libdxtlt_gfx950 error {code}: {message}")
        self.code = code


def build(force: bool = False) -> str:
    return _build.build(force=force)


def load():
    return _lib.load()


def set_tuning(tile_threads: int = 0, force_path: int = 0) -> None:
    """Tests/tuning: workgroup size of the aligned tiles (0 = per-format default) and kernel path (0 = automatic, 2 = halo /
    shifted tiles always, 0x20 = their generic LDS accesses).  Bits outside tuning_mask() are ignored."""
    load().dxtlt_set_tuning(int(tile_threads), int(force_path))


def tuning_mask() -> int:
    """The force_path bits this build honours: 0x22 for the shipped library (the experiments side build knows more)."""
    return int(load().dxtlt_tuning_mask())


def set_auto_estimator_threads(threads: int) -> None:
    """Opt-in (process-wide): run the size estimator of the auto transforms on `threads` host threads, once per distinct
    section (include/dxtlt_gfx950.h).  1 = the reference's sequence of calls.  The estimator must be thread-safe."""
    load().dxtlt_set_auto_estimator_threads(int(threads))


def get_auto_estimator_threads() -> int:
    return int(load().dxtlt_get_auto_estimator_threads())


# ------------------------------------------------------------------------------------------------------
# buffer plumbing
# ------------------------------------------------------------------------------------------------------
def _is_torch_tensor(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


class _Buf:
    __slots__ = ("ptr", "nbytes", "device", "keep")

    def __init__(self, x, writable: bool):
        if _is_torch_tensor(x):
            import torch

            if x.dtype != torch.uint8 or not x.is_contiguous():
                raise TypeError("torch buffers must be contiguous uint8 tensors")
            self.ptr, self.nbytes, self.keep = x.data_ptr(), x.numel(), x
            self.device = x.device.index if x.is_cuda else None
            if self.device is None and x.numel() and not x.is_cpu:
                raise TypeError("unsupported torch device")
        else:
            if isinstance(x, (bytes, bytearray, memoryview)):
                if writable and isinstance(x, bytes):
                    raise TypeError("output buffer must be writable")
                a = np.frombuffer(x, dtype=np.uint8)
            else:
                a = np.asarray(x)
            if a.dtype != np.uint8 or a.ndim != 1 or not a.flags.c_contiguous:
                raise TypeError("host buffers must be 1-D contiguous uint8")
            if writable and not a.flags.writeable:
                raise TypeError("output buffer must be writable")
            self.ptr, self.nbytes, self.device, self.keep = a.ctypes.data, a.size, None, a


def _check(rc: int) -> None:
    if rc != _lib.OK:
        raise DeviceError(rc, _lib.last_error())


def _settings_tuple(fmt: str, settings):
    mode = int(settings.decorrelation_mode)
    sa = bool(getattr(settings, "split_alpha_endpoints", False)) if fmt == "bc3" else False
    return mode, sa, bool(settings.split_colour_endpoints)


def _call(fmt: str, inverse: bool, input, output, settings) -> None:
    src, dst = _Buf(input, False), _Buf(output, True)
    block = BLOCK_BYTES[fmt]
    # safe-wrapper validation order (bc1 safe/transform_with_settings.rs:93-105): length, then size
    if src.nbytes % block != 0:
        raise InvalidLength(src.nbytes)
    if dst.nbytes < src.nbytes:
        raise OutputBufferTooSmall(src.nbytes, dst.nbytes)
    if (src.device is None) != (dst.device is None):
        raise TypeError("input and output must both be host buffers or both be device tensors")
    l = load()
    mode, sa, sc = _settings_tuple(fmt, settings)
    d = "untransform" if inverse else "transform"
    if src.device is None:
        f = getattr(l, f"dxtlt_{d}_{fmt}_with_settings")
        rc = f(src.ptr, dst.ptr, src.nbytes, mode, sa, sc) if fmt == "bc3" else f(src.ptr, dst.ptr, src.nbytes, mode, sc)
    else:
        import torch

        if src.device != dst.device:
            raise TypeError("input and output tensors live on different devices")
        with torch.cuda.device(src.device):
            stream = torch.cuda.current_stream().cuda_stream
            f = getattr(l, f"dxtlt_{d}_{fmt}_with_settings_device")
            if fmt == "bc3":
                rc = f(src.ptr, dst.ptr, src.nbytes, mode, sa, sc, stream)
            else:
                rc = f(src.ptr, dst.ptr, src.nbytes, mode, sc, stream)
    _check(rc)


def transform_bc1_with_settings(input, output, settings: Bc1TransformSettings = Bc1TransformSettings()) -> None:
    _call("bc1", False, input, output, settings)


def untransform_bc1_with_settings(input, output, settings: Bc1TransformSettings = Bc1TransformSettings()) -> None:
    _call("bc1", True, input, output, settings)


def transform_bc2_with_settings(input, output, settings: Bc2TransformSettings = Bc2TransformSettings()) -> None:
    _call("bc2", False, input, output, settings)


def untransform_bc2_with_settings(input, output, settings: Bc2TransformSettings = Bc2TransformSettings()) -> None:
    _call("bc2", True, input, output, settings)


def transform_bc3_with_settings(input, output, settings: Bc3TransformSettings = Bc3TransformSettings()) -> None:
    _call("bc3", False, input, output, settings)


def untransform_bc3_with_settings(input, output, settings: Bc3TransformSettings = Bc3TransformSettings()) -> None:
    _call("bc3", True, input, output, settings)


def transform_range(fmt: str, inverse: bool, src, dst, total_blocks: int, first_block: int, num_blocks: int,
                    settings) -> None:
    """dxtlt_transform_range_device on torch CUDA tensors.  The AoS-side tensor starts at block `first_block`;
    the SoA-side tensor is the whole transformed buffer."""
    import torch

    s, d = _Buf(src, False), _Buf(dst, True)
    if s.device is None or d.device is None:
        raise TypeError("transform_range takes device tensors")
    block = BLOCK_BYTES[fmt]
    aos, soa = (d, s) if inverse else (s, d)
    if aos.nbytes < num_blocks * block or soa.nbytes < total_blocks * block:
        raise OutputBufferTooSmall(max(num_blocks, total_blocks) * block, min(aos.nbytes, soa.nbytes))
    mode, sa, sc = _settings_tuple(fmt, settings)
    with torch.cuda.device(s.device):
        stream = torch.cuda.current_stream().cuda_stream
        _check(load().dxtlt_transform_range_device(_FMT_ID[fmt], inverse, s.ptr, d.ptr, total_blocks, first_block,
                                                   num_blocks, mode, sa, sc, stream))


def transform_sharded(fmt: str, inverse: bool, input, output, settings, num_devices: int = 0) -> None:
    """Host buffers, block range sharded over the node's GPUs inside this process (no collective)."""
    src, dst = _Buf(input, False), _Buf(output, True)
    if src.device is not None or dst.device is not None:
        raise TypeError("transform_sharded takes host buffers")
    if src.nbytes % BLOCK_BYTES[fmt] != 0:
        raise InvalidLength(src.nbytes)
    if dst.nbytes < src.nbytes:
        raise OutputBufferTooSmall(src.nbytes, dst.nbytes)
    mode, sa, sc = _settings_tuple(fmt, settings)
    _check(load().dxtlt_transform_sharded(_FMT_ID[fmt], inverse, src.ptr, dst.ptr, src.nbytes, mode, sa, sc,
                                          int(num_devices)))


def sharded_last_stats() -> list[dict]:
    """What this thread's last transform_sharded call did, shard by shard (dxtlt_sharded_last_stats): device, the number
    of CPUs the shard's worker thread was bound to (0 = not bound), block range, seconds."""
    import ctypes as C

    class _Stat(C.Structure):
        _fields_ = [("device", C.c_int32), ("cpus_bound", C.c_int32), ("first_block", C.c_uint64), ("blocks", C.c_uint64),
                    ("seconds", C.c_double)]

    l = load()
    l.dxtlt_sharded_last_stats.argtypes, l.dxtlt_sharded_last_stats.restype = [C.POINTER(_Stat), C.c_int32], C.c_int32
    buf = (_Stat * 64)()
    n = min(64, int(l.dxtlt_sharded_last_stats(buf, 64)))
    return [{"device": s.device, "cpus_bound": s.cpus_bound, "first_block": s.first_block, "blocks": s.blocks,
             "seconds": s.seconds} for s in buf[:n]]


def device_local_cpulist(device: int) -> str:
    """CPUs local to a HIP device's PCI function ("" when the kernel does not say): dxtlt_device_local_cpulist."""
    import ctypes as C

    l = load()
    l.dxtlt_device_local_cpulist.argtypes, l.dxtlt_device_local_cpulist.restype = [C.c_int32, C.c_char_p, C.c_size_t], C.c_int32
    out = C.create_string_buffer(4096)
    l.dxtlt_device_local_cpulist(int(device), out, 4096)
    return out.value.decode()


def fill_splitmix64(tensor, seed: int, first_qword: int = 0) -> None:
    """Fill a CUDA uint8 tensor with the synthetic block stream (same bytes as oracle_fill_splitmix64)."""
    import torch

    b = _Buf(tensor, True)
    if b.device is None:
        raise TypeError("fill_splitmix64 takes a device tensor")
    with torch.cuda.device(b.device):
        stream = torch.cuda.current_stream().cuda_stream
        _check(load().dxtlt_fill_splitmix64_device(b.ptr, b.nbytes, seed & (2**64 - 1), first_qword, stream))


# ------------------------------------------------------------------------------------------------------
# host-side layout logic shared by the multi-GPU paths (pure Python, mirrors csrc/bcn_launch.h)
# ------------------------------------------------------------------------------------------------------
def stream_table(fmt: str, settings) -> list[tuple[int, int]]:
    """[(offset_multiplier, width)] per stream: stream s starts at byte offset_multiplier * N and holds
    `width` bytes per block (csrc/bcn_launch.h make_streams)."""
    _, sa, sc = _settings_tuple(fmt, settings)
    widths: list[int] = []
    if fmt == "bc3":
        widths += [1, 1] if sa else [2]
        widths += [6]
    if fmt == "bc2":
        widths += [8]
    widths += [2, 2] if sc else [4]
    widths += [4]
    out, off = [], 0
    for w in widths:
        out.append((off, w))
        off += w
    assert off == BLOCK_BYTES[fmt]
    return out


def plan_shards(total_blocks: int, shards: int, align_blocks: int = 2048) -> list[tuple[int, int]]:
    """Contiguous (first_block, num_blocks) per shard; equal shares rounded down to `align_blocks`, the last
    shard takes the remainder (same rule as dxtlt_api.cpp plan_shards)."""
    share = total_blocks // shards
    share -= share % align_blocks
    plan, at = [], 0
    for i in range(shards):
        n = total_blocks - at if i == shards - 1 else share
        plan.append((at, n))
        at += n
    return plan
