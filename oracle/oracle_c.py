"""ctypes binding of the C oracle (oracle/dxtlt_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of bench.py.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdxtlt_oracle.so")

NONE, VAR1, VAR2, VAR3 = 0, 1, 2, 3
OK, INVALID_LENGTH, OUTPUT_TOO_SMALL = 0, 1, 2


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (``make -C oracle``) if the .so is missing or stale.  DXTLT_ORACLE_SO names another
    build of the same sources instead (tools/asan_host_check.sh: the sanitizer build)."""
    if os.environ.get("DXTLT_ORACLE_SO"):
        return os.environ["DXTLT_ORACLE_SO"]
    srcs = [os.path.join(_HERE, f) for f in ("dxtlt_oracle.c", "dxtlt_oracle_bc7.c", "dxtlt_oracle_avx2.c",
                                             "dxtlt_oracle_norm.c", "dxtlt_oracle.h")]
    stale = (
        force
        or not os.path.exists(_SO)
        or any(os.path.exists(f) and os.path.getmtime(_SO) < os.path.getmtime(f) for f in srcs)
    )
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libdxtlt_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        l = C.CDLL(build())
        u8p, sz, i = C.c_void_p, C.c_size_t, C.c_int
        l.oracle_decorrelate_565.argtypes = [C.c_uint16, i]
        l.oracle_decorrelate_565.restype = C.c_uint16
        l.oracle_recorrelate_565.argtypes = [C.c_uint16, i]
        l.oracle_recorrelate_565.restype = C.c_uint16
        for n in ("bc1", "bc2"):
            for d in ("transform", "untransform"):
                f = getattr(l, f"oracle_{d}_{n}")
                f.argtypes = [u8p, u8p, sz, i, i]
                f.restype = None
                f = getattr(l, f"oracle_{d}_{n}_safe")
                f.argtypes = [u8p, sz, u8p, sz, i, i]
                f.restype = i
        for d in ("transform", "untransform"):
            f = getattr(l, f"oracle_{d}_bc3")
            f.argtypes = [u8p, u8p, sz, i, i, i]
            f.restype = None
            f = getattr(l, f"oracle_{d}_bc3_safe")
            f.argtypes = [u8p, sz, u8p, sz, i, i, i]
            f.restype = i
        for n in ("bc1", "bc2", "bc3"):
            f = getattr(l, f"oracle_generate_{n}_test_data")
            f.argtypes = [sz, u8p]
            f.restype = None
        l.oracle_split_565_color_endpoints.argtypes = [u8p, u8p, sz]
        l.oracle_split_565_color_endpoints.restype = None
        l.oracle_fill_splitmix64.argtypes = [u8p, sz, C.c_uint64, C.c_uint64]
        l.oracle_fill_splitmix64.restype = None
        l.oracle_sum_u64.argtypes = [u8p, sz]
        l.oracle_sum_u64.restype = C.c_uint64
        l.oracle_run_mt.argtypes = [i, i, u8p, u8p, sz, i, i, i, i]
        l.oracle_run_mt.restype = None
        for n in ("oracle_transform_bc7", "oracle_untransform_bc7"):
            getattr(l, n).argtypes = [u8p, u8p, sz]
            getattr(l, n).restype = None
        l.oracle_bc7_force_modes.argtypes = [u8p, sz]
        l.oracle_bc7_force_modes.restype = None
        for n in ("oracle_bc7_record_of_block", "oracle_bc7_block_of_record"):
            getattr(l, n).argtypes, getattr(l, n).restype = [u8p, u8p], None
        l.oracle_bc7_granule.argtypes, l.oracle_bc7_granule.restype = [], C.c_uint
        l.oracle_simd_available.argtypes, l.oracle_simd_available.restype = [], i
        l.oracle_simd_level.argtypes, l.oracle_simd_level.restype = [], i
        l.oracle_simd_set_cap.argtypes, l.oracle_simd_set_cap.restype = [i], i
        l.oracle_bc1_default_simd_mt.argtypes = [i, u8p, u8p, sz, i]
        l.oracle_bc1_default_simd_mt.restype = None
        l.oracle_bc23_simd_mt.argtypes = [i, i, u8p, u8p, sz, i]
        l.oracle_bc23_simd_mt.restype = None
        l.oracle_decode_bc1_block.argtypes, l.oracle_decode_bc1_block.restype = [u8p, u8p], None
        l.oracle_normalize_bc1_blocks.argtypes, l.oracle_normalize_bc1_blocks.restype = [u8p, u8p, sz, i], None
        l.oracle_normalize_bc1_split_blocks_in_place.argtypes = [u8p, u8p, sz, i]
        l.oracle_normalize_bc1_split_blocks_in_place.restype = None
        l.oracle_normalize_bc1_blocks_all_modes.argtypes = [u8p, u8p, u8p, u8p, sz]
        l.oracle_normalize_bc1_blocks_all_modes.restype = i
        l.oracle_transform_bc1_with_normalize_blocks.argtypes = [u8p, u8p, sz, i, i, i]
        l.oracle_transform_bc1_with_normalize_blocks.restype = i
        l.oracle_decode_bc2_block.argtypes, l.oracle_decode_bc2_block.restype = [u8p, u8p], None
        l.oracle_decode_bc3_block.argtypes, l.oracle_decode_bc3_block.restype = [u8p, u8p], None
        l.oracle_decode_blocks.argtypes, l.oracle_decode_blocks.restype = [C.c_int, u8p, u8p, C.c_size_t], None
        l.oracle_count_pixel_differences.argtypes = [C.c_int, u8p, u8p, C.c_size_t]
        l.oracle_count_pixel_differences.restype = C.c_uint64
        l.oracle_normalize_bc2_blocks.argtypes, l.oracle_normalize_bc2_blocks.restype = [u8p, u8p, sz, i], None
        l.oracle_normalize_bc2_split_blocks_in_place.argtypes = [u8p, u8p, u8p, sz, i]
        l.oracle_normalize_bc2_split_blocks_in_place.restype = None
        l.oracle_normalize_bc2_blocks_all_modes.argtypes = [u8p, u8p, u8p, u8p, sz]
        l.oracle_normalize_bc2_blocks_all_modes.restype = None
        l.oracle_normalize_bc3_blocks.argtypes, l.oracle_normalize_bc3_blocks.restype = [u8p, u8p, sz, i, i], None
        l.oracle_normalize_bc3_split_blocks_in_place.argtypes = [u8p, u8p, u8p, u8p, sz, i, i]
        l.oracle_normalize_bc3_split_blocks_in_place.restype = None
        l.oracle_normalize_bc3_blocks_all_modes.argtypes = [u8p, C.POINTER(u8p), sz]
        l.oracle_normalize_bc3_blocks_all_modes.restype = None
        _lib = l
    return _lib


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data


def _as_u8(x) -> np.ndarray:
    a = np.frombuffer(x, dtype=np.uint8) if isinstance(x, (bytes, bytearray, memoryview)) else np.asarray(x)
    assert a.dtype == np.uint8 and a.ndim == 1
    return a


BLOCK = {"bc1": 8, "bc2": 16, "bc3": 16}


def transform(fmt: str, data, variant: int = VAR1, split_colour: bool = True, split_alpha: bool = True,
              inverse: bool = False) -> np.ndarray:
    """oracle_{un,}transform_bcN on a whole buffer (no validation: len must be a block multiple)."""
    a = np.ascontiguousarray(_as_u8(data))
    assert a.size % BLOCK[fmt] == 0
    out = np.empty_like(a)
    d = "untransform" if inverse else "transform"
    f = getattr(lib(), f"oracle_{d}_{fmt}")
    if fmt == "bc3":
        f(_ptr(a), _ptr(out), a.size, int(variant), int(split_alpha), int(split_colour))
    else:
        f(_ptr(a), _ptr(out), a.size, int(variant), int(split_colour))
    return out


def transform_safe(fmt: str, data, out_len: int, variant: int = VAR1, split_colour: bool = True,
                   split_alpha: bool = True, inverse: bool = False):
    """Safe-wrapper semantics: returns (code, out)."""
    a = np.ascontiguousarray(_as_u8(data))
    out = np.zeros(max(out_len, 1), dtype=np.uint8)
    d = "untransform" if inverse else "transform"
    f = getattr(lib(), f"oracle_{d}_{fmt}_safe")
    if fmt == "bc3":
        rc = f(_ptr(a), a.size, _ptr(out), out_len, int(variant), int(split_alpha), int(split_colour))
    else:
        rc = f(_ptr(a), a.size, _ptr(out), out_len, int(variant), int(split_colour))
    return rc, out[:out_len]


def generate_test_data(fmt: str, num_blocks: int) -> np.ndarray:
    out = np.empty(num_blocks * BLOCK[fmt], dtype=np.uint8)
    getattr(lib(), f"oracle_generate_{fmt}_test_data")(num_blocks, _ptr(out))
    return out


def split_565_color_endpoints(data) -> np.ndarray:
    a = np.ascontiguousarray(_as_u8(data))
    out = np.empty_like(a)
    lib().oracle_split_565_color_endpoints(_ptr(a), _ptr(out), a.size)
    return out


def fill_splitmix64(len_bytes: int, seed: int, first_qword: int = 0) -> np.ndarray:
    out = np.empty(len_bytes, dtype=np.uint8)
    lib().oracle_fill_splitmix64(_ptr(out), len_bytes, seed & (2**64 - 1), first_qword)
    return out


def sum_u64(data) -> int:
    a = np.ascontiguousarray(_as_u8(data))
    return int(lib().oracle_sum_u64(_ptr(a), a.size))


def run_mt(fmt: str, src: np.ndarray, dst: np.ndarray, variant: int, split_colour: bool, split_alpha: bool,
           inverse: bool, threads: int) -> None:
    kind = {"bc1": 1, "bc2": 2, "bc3": 3}[fmt]
    lib().oracle_run_mt(kind, int(inverse), _ptr(src), _ptr(dst), src.size, int(variant), int(split_alpha),
                        int(split_colour), int(threads))


def decorrelate(v: int, variant: int) -> int:
    return int(lib().oracle_decorrelate_565(v, variant))


def recorrelate(v: int, variant: int) -> int:
    return int(lib().oracle_recorrelate_565(v, variant))


def simd_available() -> bool:
    return bool(lib().oracle_simd_available())


def simd_level() -> int:
    """0 = scalar, 2 = AVX2, 5 = AVX-512BW: what run_bc1_default_simd uses on this CPU (under the current cap)."""
    return int(lib().oracle_simd_level())


def simd_set_cap(cap: int) -> int:
    """Cap the vector level (0, 2, 5); returns the level now in effect."""
    return int(lib().oracle_simd_set_cap(int(cap)))


SIMD_NAMES = {0: "scalar", 2: "AVX2", 5: "AVX-512BW"}


def run_bc1_default_simd(src: np.ndarray, dst: np.ndarray, inverse: bool, threads: int) -> None:
    """AVX-512BW / AVX2 port (scalar when the CPU has neither) of BC1 {Variant1, split}; cpu_baseline leg and its test only."""
    lib().oracle_bc1_default_simd_mt(int(inverse), _ptr(src), _ptr(dst), src.size, int(threads))


def run_bc23_simd(kind: int, src: np.ndarray, dst: np.ndarray, inverse: bool, threads: int) -> None:
    """AVX2 port (scalar when the CPU has no AVX2) of BC2 {Variant1, split colours} (kind 2) or BC3 standard {None, no
    splits} (kind 3); cpu_baseline leg and its test only."""
    assert kind in (2, 3) and src.size % 16 == 0 and dst.size == src.size
    lib().oracle_bc23_simd_mt(kind, int(inverse), _ptr(src), _ptr(dst), src.size, int(threads))


def transform_bc7(data, inverse: bool = False) -> np.ndarray:
    """BC7 granule-sorted field split v2 (docs/BC7_FORMAT.md) -- this build's own format, parity unpinned."""
    a = np.ascontiguousarray(_as_u8(data))
    assert a.size % 16 == 0
    out = np.empty_like(a)
    (lib().oracle_untransform_bc7 if inverse else lib().oracle_transform_bc7)(_ptr(a), _ptr(out), a.size)
    return out


def bc7_record(block, inverse: bool = False) -> np.ndarray:
    """One 16-byte block -> its 16-byte record (or back)."""
    a = np.ascontiguousarray(_as_u8(block))
    assert a.size == 16
    out = np.empty_like(a)
    (lib().oracle_bc7_block_of_record if inverse else lib().oracle_bc7_record_of_block)(_ptr(a), _ptr(out))
    return out


def bc7_granule() -> int:
    return int(lib().oracle_bc7_granule())


def bc7_force_modes(data: np.ndarray) -> np.ndarray:
    """In place: give every block a valid mode marker, mode = (byte 15 & 7)."""
    assert data.dtype == np.uint8 and data.size % 16 == 0 and data.flags.c_contiguous
    lib().oracle_bc7_force_modes(_ptr(data), data.size)
    return data


# ---- BC1 block normalisation (reference experimental module; dxtlt_oracle_norm.c) --------------------------
NORMALIZE_NONE, NORMALIZE_COLOR0_ONLY, NORMALIZE_REPLICATE_COLOR = 0, 1, 2


def decode_bc1_block(block) -> np.ndarray:
    """16 RGBA8888 pixels, shape (16, 4), row-major (util/bc1_decode.rs:42)."""
    a = np.ascontiguousarray(_as_u8(block))
    assert a.size == 8
    out = np.empty(64, dtype=np.uint8)
    lib().oracle_decode_bc1_block(_ptr(a), _ptr(out))
    return out.reshape(16, 4)


def normalize_bc1_blocks(data, mode: int) -> np.ndarray:
    a = np.ascontiguousarray(_as_u8(data))
    assert a.size % 8 == 0
    out = np.empty_like(a)
    lib().oracle_normalize_bc1_blocks(_ptr(a), _ptr(out), a.size, int(mode))
    return out


def normalize_bc1_split_blocks(colors, indices, mode: int):
    """Returns normalised copies of the two arrays (the C function works in place on the copies)."""
    c = np.array(_as_u8(colors), dtype=np.uint8, copy=True)
    x = np.array(_as_u8(indices), dtype=np.uint8, copy=True)
    assert c.size == x.size and c.size % 4 == 0
    lib().oracle_normalize_bc1_split_blocks_in_place(_ptr(c), _ptr(x), c.size // 4, int(mode))
    return c, x


def normalize_bc1_blocks_all_modes(data):
    a = np.ascontiguousarray(_as_u8(data))
    assert a.size % 8 == 0
    outs = [np.empty_like(a) for _ in range(3)]
    any_n = lib().oracle_normalize_bc1_blocks_all_modes(_ptr(a), _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]), a.size)
    return outs, bool(any_n)


def transform_bc1_with_normalize_blocks(data, mode: int, variant: int = VAR1, split_colour: bool = True) -> np.ndarray:
    a = np.ascontiguousarray(_as_u8(data))
    assert a.size % 8 == 0
    out = np.empty_like(a)
    rc = lib().oracle_transform_bc1_with_normalize_blocks(_ptr(a), _ptr(out), a.size, int(mode), int(variant),
                                                          int(split_colour))
    assert rc == 0
    return out


# ---- BC2 / BC3 block normalisation (reference experimental modules; dxtlt_oracle_norm.c) ---------------------
ALPHA_NONE, ALPHA_UNIFORM_ZERO_INDICES, ALPHA_OPAQUE_FILL_ALL, ALPHA_OPAQUE_ZERO_ALPHA_MAX_INDICES = 0, 1, 2, 3


def decode_block(fmt: str, block) -> np.ndarray:
    """16 RGBA8888 pixels, shape (16, 4) (bc1/bc2/bc3 util decoders of the reference)."""
    a = np.ascontiguousarray(_as_u8(block))
    assert a.size == BLOCK[fmt]
    out = np.empty(64, dtype=np.uint8)
    getattr(lib(), f"oracle_decode_{fmt}_block")(_ptr(a), _ptr(out))
    return out.reshape(16, 4)


def normalize_bc2_blocks(data, color_mode: int) -> np.ndarray:
    a = np.ascontiguousarray(_as_u8(data))
    assert a.size % 16 == 0
    out = np.empty_like(a)
    lib().oracle_normalize_bc2_blocks(_ptr(a), _ptr(out), a.size, int(color_mode))
    return out


def normalize_bc3_blocks(data, alpha_mode: int, color_mode: int) -> np.ndarray:
    a = np.ascontiguousarray(_as_u8(data))
    assert a.size % 16 == 0
    out = np.empty_like(a)
    lib().oracle_normalize_bc3_blocks(_ptr(a), _ptr(out), a.size, int(alpha_mode), int(color_mode))
    return out


def normalize_bc2_split_blocks(alpha, colors, indices, color_mode: int):
    al = np.ascontiguousarray(_as_u8(alpha))
    c = np.array(_as_u8(colors), dtype=np.uint8, copy=True)
    x = np.array(_as_u8(indices), dtype=np.uint8, copy=True)
    lib().oracle_normalize_bc2_split_blocks_in_place(_ptr(al), _ptr(c), _ptr(x), c.size // 4, int(color_mode))
    return c, x


def normalize_bc3_split_blocks(aep, aidx, cep, cidx, alpha_mode: int, color_mode: int):
    arrs = [np.array(_as_u8(v), dtype=np.uint8, copy=True) for v in (aep, aidx, cep, cidx)]
    lib().oracle_normalize_bc3_split_blocks_in_place(*[_ptr(v) for v in arrs], arrs[0].size // 2, int(alpha_mode),
                                                     int(color_mode))
    return arrs


def normalize_bc2_blocks_all_modes(data):
    a = np.ascontiguousarray(_as_u8(data))
    outs = [np.empty_like(a) for _ in range(3)]
    lib().oracle_normalize_bc2_blocks_all_modes(_ptr(a), *[_ptr(o) for o in outs], a.size)
    return outs


def normalize_bc3_blocks_all_modes(data):
    a = np.ascontiguousarray(_as_u8(data))
    outs = [np.empty_like(a) for _ in range(12)]
    ptrs = (C.c_void_p * 12)(*[o.ctypes.data for o in outs])
    lib().oracle_normalize_bc3_blocks_all_modes(_ptr(a), ptrs, a.size)
    return outs


# ---- array decoders (dxtlt_oracle_norm.c) -------------------------------------------------------------------
def decode_blocks(fmt: str, data) -> np.ndarray:
    """One Decoded4x4Block (64 bytes: sixteen r, g, b, a, row-major) per block."""
    a = np.ascontiguousarray(_as_u8(data))
    assert a.size % BLOCK[fmt] == 0
    out = np.empty(a.size // BLOCK[fmt] * 64, dtype=np.uint8)
    lib().oracle_decode_blocks({"bc1": 1, "bc2": 2, "bc3": 3}[fmt], _ptr(a), _ptr(out), a.size // BLOCK[fmt])
    return out


def count_pixel_differences(fmt: str, a, b) -> int:
    x, y = np.ascontiguousarray(_as_u8(a)), np.ascontiguousarray(_as_u8(b))
    assert x.size == y.size and x.size % BLOCK[fmt] == 0
    return int(lib().oracle_count_pixel_differences({"bc1": 1, "bc2": 2, "bc3": 3}[fmt], _ptr(x), _ptr(y), x.size // BLOCK[fmt]))
