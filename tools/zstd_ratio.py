"""Compressed sizes with the image's system libzstd (libzstd.so.1, 1.4.8 here) through ctypes -- measurement only.

BASELINE.json configs[4] quotes its compression ratios with `zstd -3` / `-19`; the image has no Python zstd binding, so
round 1 reported zlib-6 alone.  The reference pins zstd 1.5.7 (src/Cargo.lock, zstd-sys 2.0.16) and compresses magicless
without checksum (extensions/compressors/dxt-lossless-transform-zstd/src/lib.rs:146-200); ZSTD_compress of 1.4.8 writes
the 4-byte magic and differs in absolute ratio by version -- the comparison plain vs transformed is what is reported.
Used by bench.py's archive leg and tests; never by the product path."""
import ctypes
import ctypes.util

_lib = None


def _load():
    global _lib
    if _lib is None:
        for name in ("libzstd.so.1", ctypes.util.find_library("zstd")):
            if not name:
                continue
            try:
                lib = ctypes.CDLL(name)
            except OSError:
                continue
            lib.ZSTD_compressBound.restype = ctypes.c_size_t
            lib.ZSTD_compressBound.argtypes = [ctypes.c_size_t]
            lib.ZSTD_compress.restype = ctypes.c_size_t
            lib.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
            lib.ZSTD_decompress.restype = ctypes.c_size_t
            lib.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
            lib.ZSTD_isError.restype = ctypes.c_uint
            lib.ZSTD_isError.argtypes = [ctypes.c_size_t]
            lib.ZSTD_versionString.restype = ctypes.c_char_p
            _lib = lib
            break
        else:
            _lib = False
    return _lib or None


def available():
    return _load() is not None


def version():
    lib = _load()
    return lib.ZSTD_versionString().decode() if lib else None


def compress(data, level):
    """bytes-like -> compressed bytes (one ZSTD_compress call), or None when libzstd is not there"""
    lib = _load()
    if lib is None:
        return None
    raw = bytes(data)
    cap = lib.ZSTD_compressBound(len(raw))
    out = ctypes.create_string_buffer(cap)
    n = lib.ZSTD_compress(out, cap, raw, len(raw), int(level))
    if lib.ZSTD_isError(n):
        raise RuntimeError(f"ZSTD_compress failed at level {level}")
    return out.raw[:n]


def decompress(blob, size):
    lib = _load()
    out = ctypes.create_string_buffer(max(size, 1))
    n = lib.ZSTD_decompress(out, size, bytes(blob), len(blob))
    if lib.ZSTD_isError(n) or n != size:
        raise RuntimeError("ZSTD_decompress failed")
    return out.raw[:size]


def compressed_size(data, level):
    c = compress(data, level)
    return None if c is None else len(c)
