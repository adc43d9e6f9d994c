"""File-format layer: TransformHeader bit layout, DDS parsing (CPU only) and the DDS handler
round trip (GPU).  Mirrors the reference's tests: embed/mod.rs:204-256, embed/formats/bc1.rs tests, dds/parse_dds.rs
tests, dds/likely_dds.rs tests, handler/file_format_handler.rs tests."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

import cabi
from helpers import GOLDEN, all_settings


class DdsInfo(C.Structure):
    _fields_ = [("Format", C.c_uint8), ("DataOffset", C.c_uint8), ("DataLength", C.c_uint32)]


NOT_A_DDS, UNKNOWN, BC1, BC2, BC3, BC6H, BC7, RGBA8888, BGRA8888, BGR888, BC4, BC5 = range(12)


@pytest.fixture(scope="module")
def lib(pkg):
    l = C.CDLL(pkg._lib.lib_path())
    vp, sz = C.c_void_p, C.c_size_t
    l.dxtlt_transform_header_pack.argtypes = [C.c_int32, C.c_uint8, C.c_bool, C.c_bool]
    l.dxtlt_transform_header_pack.restype = C.c_uint32
    l.dxtlt_transform_header_unpack.argtypes = [C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_uint8),
                                                C.POINTER(C.c_bool), C.POINTER(C.c_bool)]
    l.dxtlt_transform_header_unpack.restype = C.c_int32
    l.is_dds.argtypes, l.is_dds.restype = [vp, sz], C.c_bool
    l.parse_dds.argtypes, l.parse_dds.restype = [vp, sz], DdsInfo
    l.dxtlt_dds_transform.argtypes = [vp, sz, vp, sz, C.c_uint8, C.c_bool, C.c_bool]
    l.dxtlt_dds_transform.restype = C.c_int32
    l.dxtlt_dds_transform_auto.argtypes = [vp, sz, vp, sz, C.POINTER(cabi.DltSizeEstimator), C.c_bool]
    l.dxtlt_dds_transform_auto.restype = C.c_int32
    l.dxtlt_dds_untransform.argtypes = [vp, sz, vp, sz]
    l.dxtlt_dds_untransform.restype = C.c_int32
    return l


def dds_file(fmt: str) -> np.ndarray:
    h = np.fromfile(os.path.join(GOLDEN, f"r2-256-{fmt}.header.bin"), dtype=np.uint8)
    p = np.fromfile(os.path.join(GOLDEN, f"r2-256-{fmt}.payload.bin"), dtype=np.uint8)
    return np.concatenate([h, p])


def legacy_header(fourcc: bytes, width: int, height: int, mips: int = 0, pf_flags: int = 0x4) -> bytearray:
    h = bytearray(128)
    h[0:4] = b"DDS "
    struct.pack_into("<I", h, 4, 124)
    flags = 0x1 | 0x2 | 0x4 | 0x1000 | (0x20000 if mips else 0)
    struct.pack_into("<III", h, 8, flags, height, width)
    struct.pack_into("<I", h, 0x1C, mips)
    struct.pack_into("<II", h, 0x4C, 32, pf_flags)
    h[0x54:0x58] = fourcc
    return h


# ---- TransformHeader ---------------------------------------------------------------------------------------
def test_header_bit_layout(lib):
    # format in bits 0-3, data in bits 4-31 (embed/mod.rs:105-120); BC1 data = version:2 | split:1 | variant:2 with
    # Variant1=0, Variant2=1, Variant3=2, None=3 (embed/formats/bc1.rs:34-60)
    assert lib.dxtlt_transform_header_pack(0, 1, False, True) == 0x40           # BC1, Variant1 + split
    assert lib.dxtlt_transform_header_pack(0, 0, False, False) == (3 << 3) << 4  # BC1, None, no split
    assert lib.dxtlt_transform_header_pack(1, 2, False, True) == 1 | ((1 << 2 | 1 << 3) << 4)  # BC2, Variant2 + split
    assert lib.dxtlt_transform_header_pack(0, 3, False, False) == (2 << 3) << 4  # BC1, Variant3
    for fmt, has_alpha in ((0, False), (1, False), (2, True)):
        for v in range(4):
            for sa in ((False, True) if has_alpha else (False,)):
                for sc in (False, True):
                    h = lib.dxtlt_transform_header_pack(fmt, v, sa, sc)
                    assert h & 0xF == fmt and (h >> 4) & 3 == 0  # format code, header version 0
                    f, m, a, c = C.c_int32(), C.c_uint8(), C.c_bool(), C.c_bool()
                    assert lib.dxtlt_transform_header_unpack(h, C.byref(f), C.byref(m), C.byref(a), C.byref(c)) == 0
                    assert (f.value, m.value, a.value, c.value) == (fmt, v, sa, sc)


def test_header_rejects_bad_version_and_format(lib):
    f, m, a, c = C.c_int32(), C.c_uint8(), C.c_bool(), C.c_bool()
    bad_version = 0 | (3 << 4)  # bc1.rs test_invalid_header_version: only version 0 is valid
    assert lib.dxtlt_transform_header_unpack(bad_version, C.byref(f), C.byref(m), C.byref(a), C.byref(c)) == 5
    assert lib.dxtlt_transform_header_unpack(0x3, C.byref(f), C.byref(m), C.byref(a), C.byref(c)) == 4  # BC7
    assert lib.dxtlt_transform_header_unpack(0xF, C.byref(f), C.byref(m), C.byref(a), C.byref(c)) == 4


def _header_api(lib):
    lib.dxtlt_transform_header_new.argtypes = [C.c_int32, C.c_uint32]
    lib.dxtlt_transform_header_new.restype = C.c_uint32
    lib.dxtlt_transform_header_format.argtypes = [C.c_uint32]
    lib.dxtlt_transform_header_format.restype = C.c_int32
    lib.dxtlt_transform_header_format_data.argtypes = [C.c_uint32]
    lib.dxtlt_transform_header_format_data.restype = C.c_uint32
    lib.dxtlt_transform_header_write.argtypes = [C.c_uint32, C.c_void_p]
    lib.dxtlt_transform_header_write.restype = None
    lib.dxtlt_transform_header_read.argtypes = [C.c_void_p]
    lib.dxtlt_transform_header_read.restype = C.c_uint32
    lib.dxtlt_transform_header_pack_reserved_format.argtypes = [C.c_int32, C.c_bool]
    lib.dxtlt_transform_header_pack_reserved_format.restype = C.c_uint32
    lib.dxtlt_transform_header_unpack_reserved_format.argtypes = [C.c_uint32, C.POINTER(C.c_int32), C.POINTER(C.c_bool)]
    lib.dxtlt_transform_header_unpack_reserved_format.restype = C.c_int32
    return lib


def test_transform_format_codes_and_header_bitfield(lib):
    """embed/mod.rs tests test_transform_format_conversion, test_transform_header_bitfield, test_header_read_write,
    test_little_endian_byte_order, replayed on the C surface"""
    l = _header_api(lib)
    # Bc1 Bc2 Bc3 Bc7 Bc6H Rgba8888 Bgra8888 Bgr888 Bc4 Bc5 = 0 .. 9; 0x0F is no format
    for code in range(10):
        assert l.dxtlt_transform_header_format(l.dxtlt_transform_header_new(code, 0)) == code
    for code in range(10, 16):
        assert l.dxtlt_transform_header_format(code) == -1
    h = l.dxtlt_transform_header_new(0, 0x0ABCDEF0)
    assert l.dxtlt_transform_header_format(h) == 0 and l.dxtlt_transform_header_format_data(h) == 0x0ABCDEF0
    h2 = l.dxtlt_transform_header_new(2, 0xFFFFFFFF)   # data masked to 28 bits
    assert l.dxtlt_transform_header_format(h2) == 2 and l.dxtlt_transform_header_format_data(h2) == 0x0FFFFFFF
    buf = np.zeros(4, dtype=np.uint8)
    h3 = l.dxtlt_transform_header_new(3, 0x1234567)   # BC7
    assert h3 == 0x12345673
    l.dxtlt_transform_header_write(h3, buf.ctypes.data)
    assert bytes(buf) == bytes([0x73, 0x56, 0x34, 0x12])
    assert l.dxtlt_transform_header_read(buf.ctypes.data) == h3


@pytest.mark.parametrize("code", [8, 9, 5, 6, 7])   # Bc4, Bc5 (split_endpoints); Rgba8888, Bgra8888, Bgr888 (decorrelation)
def test_reserved_format_headers(lib, code):
    """embed/formats/{bc4,bc5,rgba8888,bgra8888,bgr888}.rs: version:2 | flag:1 | reserved:25 -- their tests
    test_*_pack_unpack_roundtrip, test_roundtrip_all_possible_transform_details, test_header_version_and_reserved_fields,
    test_invalid_header_version, test_invalid_reserved_bits, test_format_association"""
    l = _header_api(lib)
    f, flag = C.c_int32(), C.c_bool()
    seen = set()
    for want in (False, True):
        h = l.dxtlt_transform_header_pack_reserved_format(code, want)
        seen.add(h)
        assert l.dxtlt_transform_header_format(h) == code
        data = l.dxtlt_transform_header_format_data(h)
        assert data & 3 == 0 and data >> 3 == 0 and (data >> 2) & 1 == int(want)
        assert l.dxtlt_transform_header_unpack_reserved_format(h, C.byref(f), C.byref(flag)) == 0
        assert (f.value, flag.value) == (code, want)
    assert len(seen) == 2
    bad_version = l.dxtlt_transform_header_new(code, 3)         # only version 0 is valid
    assert l.dxtlt_transform_header_unpack_reserved_format(bad_version, C.byref(f), C.byref(flag)) == 5
    bad_reserved = l.dxtlt_transform_header_new(code, 1 << 3)   # reserved = 1
    assert l.dxtlt_transform_header_unpack_reserved_format(bad_reserved, C.byref(f), C.byref(flag)) == 5
    assert l.dxtlt_transform_header_unpack_reserved_format(l.dxtlt_transform_header_new(code, 1 << 27), C.byref(f), C.byref(flag)) == 5
    # the transform formats proper are not this family's, and no DDS call accepts a reserved format
    assert l.dxtlt_transform_header_unpack_reserved_format(l.dxtlt_transform_header_new(0, 0), C.byref(f), C.byref(flag)) == 4
    m, a, c = C.c_uint8(), C.c_bool(), C.c_bool()
    h = l.dxtlt_transform_header_pack_reserved_format(code, True)
    assert lib.dxtlt_transform_header_unpack(h, C.byref(f), C.byref(m), C.byref(a), C.byref(c)) == 4


# ---- DDS parsing -------------------------------------------------------------------------------------------
def test_is_dds(lib):
    # likely_dds.rs tests: magic + at least 128 bytes
    ok = np.frombuffer(b"DDS " + bytes(124), dtype=np.uint8)
    assert lib.is_dds(ok.ctypes.data, ok.size)
    assert not lib.is_dds(ok.ctypes.data, 127)
    bad = np.frombuffer(b"DDX " + bytes(124), dtype=np.uint8)
    assert not lib.is_dds(bad.ctypes.data, bad.size)
    assert not lib.is_dds(None, 128) and not lib.is_dds(ok.ctypes.data, 0)


def test_parse_real_textures(lib):
    for fmt, code, length in (("bc1", BC1, 32768), ("bc2", BC2, 65536), ("bc3", BC3, 65536)):
        d = dds_file(fmt)
        info = lib.parse_dds(d.ctypes.data, d.size)
        assert (info.Format, info.DataOffset, info.DataLength) == (code, 128, length)
    info = lib.parse_dds(None, 0)
    assert (info.Format, info.DataOffset, info.DataLength) == (NOT_A_DDS, 0, 0)


@pytest.mark.parametrize("fourcc,code", [(b"DXT1", BC1), (b"DXT2", BC2), (b"DXT3", BC2), (b"DXT4", BC3),
                                         (b"DXT5", BC3), (b"ATI1", BC4), (b"BC4U", BC4), (b"ATI2", BC5),
                                         (b"BC5S", BC5), (b"ABCD", UNKNOWN)])
def test_parse_legacy_fourcc(lib, fourcc, code):
    # parse_dds.rs parse_dds_handles_legacy_formats
    h = np.frombuffer(bytes(legacy_header(fourcc, 4, 4)) + bytes(16), dtype=np.uint8)
    info = lib.parse_dds(h.ctypes.data, h.size)
    assert info.Format == code and info.DataOffset == 128


@pytest.mark.parametrize("dxgi,code", [(70, BC1), (71, BC1), (72, BC1), (74, BC2), (77, BC3), (80, BC4), (83, BC5),
                                       (95, BC6H), (98, BC7), (28, RGBA8888), (87, BGRA8888), (2, UNKNOWN)])
def test_parse_dx10(lib, dxgi, code):
    h = legacy_header(b"DX10", 8, 8)
    ext = bytearray(20)
    struct.pack_into("<I", ext, 0, dxgi)
    d = np.frombuffer(bytes(h) + bytes(ext) + bytes(64), dtype=np.uint8)
    info = lib.parse_dds(d.ctypes.data, d.size)
    assert info.Format == code and info.DataOffset == 148
    short = np.frombuffer(bytes(h) + bytes(10), dtype=np.uint8)  # DX10 header cut short -> not parseable
    assert lib.parse_dds(short.ctypes.data, short.size).Format == NOT_A_DDS


@pytest.mark.parametrize("w,h,mips,fourcc,want", [
    (4, 4, 0, b"DXT1", 8), (256, 256, 0, b"DXT1", 32768), (256, 256, 9, b"DXT1", 43704),
    (5, 7, 0, b"DXT5", 2 * 2 * 16), (1, 1, 1, b"DXT3", 16), (16, 4, 3, b"DXT1", (4 * 1 + 2 * 1 + 1 * 1) * 8),
])
def test_block_data_length(lib, w, h, mips, fourcc, want):
    # calculate_data_length_for_block_compression: ceil(w/4)*ceil(h/4)*block per level, halving down to 1x1
    d = np.frombuffer(bytes(legacy_header(fourcc, w, h, mips)) + bytes(16), dtype=np.uint8)
    assert lib.parse_dds(d.ctypes.data, d.size).DataLength == want


def model_block_length(w, h, mips, block):
    """calculate_data_length_for_block_compression, level by level: u32 products wrap, the running total saturates
    (parse_dds.rs:316-323); the 1 x 1 tail in closed form so that the model finishes for any mip count"""
    total = 0
    for i in range(mips):
        if w == 1 and h == 1:
            return min(total + block * (mips - i), 0xFFFFFFFF)
        level = (((w + 3) // 4) * ((h + 3) // 4) * block) & 0xFFFFFFFF
        total = min(total + level, 0xFFFFFFFF)
        w, h = max(w // 2, 1), max(h // 2, 1)
    return total


def test_hostile_mip_counts_cost_nothing_and_saturate(lib):
    import time
    for w, h, mips, fourcc, block in [(4, 4, 0xFFFFFFFF, b"DXT1", 8), (256, 256, 0xFFFFFFFF, b"DXT5", 16),
                                      (1, 1, 0x80000000, b"DXT3", 16), (65536, 65536, 0x7FFFFFFF, b"DXT1", 8),
                                      (3, 1000, 123456789, b"DXT5", 16), (16384, 16384, 40, b"DXT1", 8)]:
        d = np.frombuffer(bytes(legacy_header(fourcc, w, h, mips)) + bytes(16), dtype=np.uint8)
        t = time.perf_counter()
        got = lib.parse_dds(d.ctypes.data, d.size).DataLength
        assert time.perf_counter() - t < 0.05                      # the file's mip count is not a loop bound
        assert got == model_block_length(w, h, mips, block), (w, h, mips)
    # the same for uncompressed pixels: 4 bytes per pixel, 1 x 1 forever
    hd = legacy_header(b"\0\0\0\0", 2, 2, 0xFFFFFFFF, 0x41)
    struct.pack_into("<IIIII", hd, 0x58, 32, 0xFF, 0xFF00, 0xFF0000, 0xFF000000)
    d = np.frombuffer(bytes(hd) + bytes(64), dtype=np.uint8)
    i = lib.parse_dds(d.ctypes.data, d.size)
    assert (i.Format, i.DataLength) == (RGBA8888, 0xFFFFFFFF)


def test_random_headers_never_read_past_the_header_and_match_the_model(lib):
    """2 000 random 128 / 148-byte headers placed at the very end of a buffer (a read past the header would run into
    the guard page of a separate mapping at best -- here the point is the length model, any size, any mip count)."""
    rng = np.random.default_rng(0xDD5)
    fourccs = [(b"DXT1", BC1, 8), (b"DXT3", BC2, 16), (b"DXT5", BC3, 16), (b"ATI1", BC4, 8), (b"BC5U", BC5, 16)]
    for case in range(2000):
        cc, code, block = fourccs[int(rng.integers(0, len(fourccs)))]
        w = int(rng.choice([0, 1, 2, 3, 4, 5, 255, 256, 257, 4096, 65535, 65536, 0xFFFFFFFF, int(rng.integers(0, 1 << 20))]))
        h = int(rng.choice([0, 1, 2, 3, 4, 7, 256, 1000, 65536, 0xFFFFFFFF, int(rng.integers(0, 1 << 20))]))
        mips = int(rng.choice([0, 1, 2, 9, 13, 33, 1000, 0xFFFFFFFF, int(rng.integers(0, 1 << 32))]))
        hd = legacy_header(cc, w, h, mips)
        if case % 3 == 0:
            struct.pack_into("<I", hd, 8, 0x1007)                  # DDSD_MIPMAPCOUNT clear: one level whatever the count says
            mips_eff = 1
        else:
            mips_eff = max(mips, 1)
        d = np.frombuffer(bytes(hd), dtype=np.uint8)
        i = lib.parse_dds(d.ctypes.data, d.size)
        assert (i.Format, i.DataOffset) == (code, 128), (case, cc)
        assert i.DataLength == model_block_length(w, h, mips_eff, block), (case, w, h, mips)


def test_uncompressed_formats(lib):
    def rgb(bits, masks, flags):
        hd = legacy_header(b"\0\0\0\0", 4, 2, 0, flags)
        struct.pack_into("<IIIII", hd, 0x58, bits, *masks)
        return np.frombuffer(bytes(hd) + bytes(64), dtype=np.uint8)
    d = rgb(32, (0xFF, 0xFF00, 0xFF0000, 0xFF000000), 0x41)
    i = lib.parse_dds(d.ctypes.data, d.size)
    assert (i.Format, i.DataLength) == (RGBA8888, 32)
    d = rgb(32, (0xFF0000, 0xFF00, 0xFF, 0xFF000000), 0x41)
    assert lib.parse_dds(d.ctypes.data, d.size).Format == BGRA8888
    d = rgb(24, (0xFF0000, 0xFF00, 0xFF, 0), 0x40)
    i = lib.parse_dds(d.ctypes.data, d.size)
    assert (i.Format, i.DataLength) == (BGR888, 24)
    d = rgb(16, (0xF800, 0x7E0, 0x1F, 0), 0x40)  # RGB565: unknown format, length from the bit count
    i = lib.parse_dds(d.ctypes.data, d.size)
    assert (i.Format, i.DataLength) == (UNKNOWN, 16)


def test_handler_argument_checks(lib):
    d = dds_file("bc1")
    out = np.zeros(d.size, dtype=np.uint8)
    assert lib.dxtlt_dds_transform(d.ctypes.data, d.size, out.ctypes.data, d.size - 1, 1, False, True) == 1
    notdds = np.zeros(200, dtype=np.uint8)
    assert lib.dxtlt_dds_transform(notdds.ctypes.data, 200, out.ctypes.data, out.size, 1, False, True) == 2
    assert lib.dxtlt_dds_transform(d.ctypes.data, d.size - 100, out.ctypes.data, out.size, 1, False, True) == 3  # truncated
    bc7 = np.frombuffer(bytes(legacy_header(b"DX10", 4, 4)) + struct.pack("<I", 98) + bytes(16) + bytes(16), dtype=np.uint8)
    assert lib.dxtlt_dds_transform(bc7.ctypes.data, bc7.size, out.ctypes.data, out.size, 1, False, True) == 4
    assert lib.dxtlt_dds_untransform(d.ctypes.data, 3, out.ctypes.data, out.size) == 3
    assert lib.dxtlt_dds_untransform(d.ctypes.data, d.size, out.ctypes.data, 10) == 1
    assert lib.dxtlt_dds_transform(None, 10, out.ctypes.data, out.size, 1, False, True) == 9


# ---- DDS handler end to end ----------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["bc1", "bc2", "bc3"])
def test_dds_roundtrip_all_settings(lib, oracle, fmt):
    d = np.concatenate([dds_file(fmt), np.arange(37, dtype=np.uint8)])  # trailing bytes are copied verbatim
    payload = d[128:-37]
    for v, sa, sc in all_settings(fmt):
        t = np.zeros_like(d)
        assert lib.dxtlt_dds_transform(d.ctypes.data, d.size, t.ctypes.data, t.size, v, bool(sa), bool(sc)) == 0
        # magic replaced by the TransformHeader, the rest of the header and the tail untouched, payload transformed
        want_header = lib.dxtlt_transform_header_pack({"bc1": 0, "bc2": 1, "bc3": 2}[fmt], v, bool(sa), bool(sc))
        assert struct.unpack("<I", t[:4].tobytes())[0] == want_header
        assert np.array_equal(t[4:128], d[4:128]) and np.array_equal(t[-37:], d[-37:])
        assert np.array_equal(t[128:-37], oracle.transform(fmt, payload, v, sc, sa))
        assert not lib.is_dds(t.ctypes.data, t.size)
        r = np.zeros_like(d)
        assert lib.dxtlt_dds_untransform(t.ctypes.data, t.size, r.ctypes.data, r.size) == 0
        assert np.array_equal(r, d)


def reference_integration_test_dds() -> np.ndarray:
    """The synthetic BC1 DDS the reference's own integration test builds (api/dxt-lossless-transform-file-formats-api/
    tests/integration_test.rs:10-57): a 128-byte legacy header (flags CAPS | HEIGHT | WIDTH | PIXELFORMAT | LINEARSIZE,
    4 x 4 pixels, FourCC DXT1 -- it leaves the pixel-format size field zero) and ONE block: red, green, zero indices."""
    d = bytearray(0x80 + 8)
    d[0:4] = b"DDS "
    struct.pack_into("<I", d, 4, 124)
    struct.pack_into("<I", d, 8, 0x1 | 0x2 | 0x4 | 0x1000 | 0x80000)
    struct.pack_into("<II", d, 0x0C, 4, 4)
    struct.pack_into("<I", d, 0x50, 0x4)
    d[0x54:0x58] = b"DXT1"
    d[0x80:0x88] = bytes([0x00, 0xF8, 0xE0, 0x07, 0, 0, 0, 0])
    return np.frombuffer(bytes(d), dtype=np.uint8).copy()


def test_reference_integration_fixture_is_recognised(lib):
    """integration_test.rs:94-105 (test_handler_detection): the handler takes the synthetic DDS, refuses 128 zero bytes."""
    d = reference_integration_test_dds()
    assert lib.is_dds(d.ctypes.data, d.size)
    info = lib.parse_dds(d.ctypes.data, d.size)
    assert (info.Format, info.DataOffset, info.DataLength) == (BC1, 128, 8)
    zeros = np.zeros(128, dtype=np.uint8)
    assert not lib.is_dds(zeros.ctypes.data, zeros.size)
    out = np.zeros(128, dtype=np.uint8)
    assert lib.dxtlt_dds_transform(zeros.ctypes.data, zeros.size, out.ctypes.data, out.size, 1, False, True) != 0


@pytest.mark.gpu
def test_reference_integration_fixture_round_trip(lib, oracle):
    """integration_test.rs:59-92 (test_dds_bc1_transform_roundtrip): transform with the default BC1 settings
    (TransformBundle::default_all), the magic must change; untransform restores it -- and here the whole file."""
    d = reference_integration_test_dds()
    t = np.zeros_like(d)
    assert lib.dxtlt_dds_transform(d.ctypes.data, d.size, t.ctypes.data, t.size, 1, False, True) == 0
    assert t[:4].tobytes() != d[:4].tobytes()
    assert np.array_equal(t[4:128], d[4:128])
    assert np.array_equal(t[128:], oracle.transform("bc1", d[128:], 1, True))
    r = np.zeros_like(d)
    assert lib.dxtlt_dds_untransform(t.ctypes.data, t.size, r.ctypes.data, r.size) == 0
    assert r[:4].tobytes() == b"DDS " and np.array_equal(r, d)


@pytest.mark.gpu
def test_dds_auto_transform(lib, oracle):
    from oracle import oracle_auto

    for fmt in ("bc1", "bc2", "bc3"):
        d = dds_file(fmt)
        est, py_est = cabi.make_estimator("zlib")
        choice, want, _ = oracle_auto.transform_auto(fmt, d[128:], lambda b: py_est(bytes(b)), False)
        t = np.zeros_like(d)
        assert lib.dxtlt_dds_transform_auto(d.ctypes.data, d.size, t.ctypes.data, t.size, C.byref(est), False) == 0
        assert np.array_equal(t[128:], want)
        f, m, a, c = C.c_int32(), C.c_uint8(), C.c_bool(), C.c_bool()
        hdr = struct.unpack("<I", t[:4].tobytes())[0]
        assert lib.dxtlt_transform_header_unpack(hdr, C.byref(f), C.byref(m), C.byref(a), C.byref(c)) == 0
        assert (m.value, int(a.value), int(c.value)) == choice
        r = np.zeros_like(d)
        assert lib.dxtlt_dds_untransform(t.ctypes.data, t.size, r.ctypes.data, r.size) == 0
        assert np.array_equal(r, d)


BC7_PRIVATE_HEADER = 3 | ((0xD175 << 12 | 2) << 4)   # TransformFormat::Bc7, vendor tag 0xD175, format version 2


def test_bc7_switch_is_off_by_default_and_needs_no_device(lib):
    """Upstream's dispatch refuses BC7; so do the handler functions unless the caller opts in to this build's format."""
    lib.dxtlt_file_formats_enable_bc7.argtypes, lib.dxtlt_file_formats_enable_bc7.restype = [C.c_bool], None
    d = dds_file("bc7")
    out = np.zeros_like(d)
    assert lib.dxtlt_dds_transform(d.ctypes.data, d.size, out.ctypes.data, out.size, 1, False, True) == 4
    hdr = np.concatenate([np.frombuffer(struct.pack("<I", 3), dtype=np.uint8), d[4:]])   # TransformFormat::Bc7, no data bits
    assert lib.dxtlt_dds_untransform(hdr.ctypes.data, hdr.size, out.ctypes.data, out.size) == 4
    lib.dxtlt_file_formats_enable_bc7(True)
    try:
        # this build's BC7 files carry a vendor tag and a format version in the data bits; all-zero data bits are
        # upstream's to assign and are refused, as is any other tag or version
        version1 = 3 | ((0xD175 << 12 | 1) << 4)   # files of the previous format version: no green decorrelation, refused
        for word in (3, 3 | (1 << 6), version1, BC7_PRIVATE_HEADER ^ (1 << 4), BC7_PRIVATE_HEADER ^ (1 << 20)):
            bad = np.concatenate([np.frombuffer(struct.pack("<I", word), dtype=np.uint8), d[4:]])
            assert lib.dxtlt_dds_untransform(bad.ctypes.data, bad.size, out.ctypes.data, out.size) == 5, hex(word)
    finally:
        lib.dxtlt_file_formats_enable_bc7(False)


@pytest.mark.gpu
def test_bc7_dds_roundtrip_when_enabled(lib, oracle):
    lib.dxtlt_file_formats_enable_bc7.argtypes, lib.dxtlt_file_formats_enable_bc7.restype = [C.c_bool], None
    d = np.concatenate([dds_file("bc7"), np.arange(21, dtype=np.uint8)])
    info = lib.parse_dds(d.ctypes.data, d.size)
    off, length = info.DataOffset, info.DataLength
    assert info.Format == BC7 and off + length == d.size - 21
    lib.dxtlt_file_formats_enable_bc7(True)
    try:
        t = np.zeros_like(d)
        assert lib.dxtlt_dds_transform(d.ctypes.data, d.size, t.ctypes.data, t.size, 1, False, True) == 0
        assert struct.unpack("<I", t[:4].tobytes())[0] == BC7_PRIVATE_HEADER
        assert np.array_equal(t[4:off], d[4:off]) and np.array_equal(t[-21:], d[-21:])
        assert np.array_equal(t[off:off + length], oracle.transform_bc7(d[off:off + length]))
        r = np.zeros_like(d)
        assert lib.dxtlt_dds_untransform(t.ctypes.data, t.size, r.ctypes.data, r.size) == 0
        assert np.array_equal(r, d)
    finally:
        lib.dxtlt_file_formats_enable_bc7(False)


# ---- many DDS files per call (additive): dxtlt_dds_transform_batch ---------------------------------------------------
class DdsBatchItem(C.Structure):
    _fields_ = [("input", C.c_void_p), ("input_len", C.c_size_t), ("output", C.c_void_p), ("output_len", C.c_size_t),
                ("decorrelation_mode", C.c_uint8), ("split_alpha_endpoints", C.c_bool), ("split_colour_endpoints", C.c_bool),
                ("status", C.c_int32)]


def bind_batch(lib):
    lib.dxtlt_dds_transform_batch.argtypes = [C.POINTER(DdsBatchItem), C.c_size_t, C.c_bool]
    lib.dxtlt_dds_transform_batch.restype = C.c_size_t
    return lib.dxtlt_dds_transform_batch


def make_items(files, outs, settings):
    items = (DdsBatchItem * len(files))()
    for it, f, o, (m, a, c) in zip(items, files, outs, settings):
        it.input, it.input_len = f.ctypes.data if f is not None else None, f.size if f is not None else 0
        it.output, it.output_len = o.ctypes.data, o.size
        it.decorrelation_mode, it.split_alpha_endpoints, it.split_colour_endpoints = m, a, c
        it.status = -1
    return items


def test_dds_batch_rejects_item_by_item_without_a_device(lib):
    """Items that fail validation get the single call's status and never reach the device."""
    batch = bind_batch(lib)
    good = dds_file("bc1")
    not_dds = np.zeros(200, dtype=np.uint8)
    short_payload = good[:1000].copy()                              # header says 32 KiB of blocks, 872 bytes follow
    unknown = np.frombuffer(bytes(legacy_header(b"ATI2", 8, 8)) + bytes(64), dtype=np.uint8)      # BC5: no transform for it
    files = [None, not_dds, short_payload, unknown, good]
    outs = [np.zeros(max(64, 0 if f is None else f.size), dtype=np.uint8) for f in files]
    outs[4] = np.zeros(10, dtype=np.uint8)                          # output too small
    items = make_items(files, outs, [(1, False, True)] * len(files))
    assert batch(items, len(files), False) == len(files)
    assert [it.status for it in items] == [9, 2, 3, 4, 1]           # NULL, invalid header, too short, unknown format, output too small
    assert batch(None, 3, False) == 3
    assert batch(items, 0, False) == 0


@pytest.mark.gpu
def test_dds_batch_equals_one_call_per_file(lib, oracle):
    batch = bind_batch(lib)
    rng = np.random.default_rng(0xDD5B)
    files, settings = [], []
    for k in range(40):
        fmt = ("bc1", "bc2", "bc3")[k % 3]
        if k % 5 == 0:
            f = dds_file(fmt)                                       # the reference's textures
        else:                                                       # synthetic: w x h blocks, some with mip chains and trailing bytes
            w, h, mips = int(rng.integers(1, 300)), int(rng.integers(1, 300)), int(rng.integers(0, 6))
            hd = legacy_header({"bc1": b"DXT1", "bc2": b"DXT3", "bc3": b"DXT5"}[fmt], w, h, mips)
            probe = np.frombuffer(bytes(hd), dtype=np.uint8)
            length = lib.parse_dds(probe.ctypes.data, probe.size).DataLength
            f = np.concatenate([probe, rng.integers(0, 256, length + int(rng.integers(0, 9)), dtype=np.uint8)])
        files.append(np.ascontiguousarray(f))
        settings.append((int(rng.integers(0, 4)), bool(rng.integers(0, 2)), bool(rng.integers(0, 2))))
    files.insert(7, np.zeros(300, dtype=np.uint8))                  # one bad apple in the middle: the rest must not care
    settings.insert(7, (1, False, True))
    outs = [np.zeros_like(f) for f in files]
    items = make_items(files, outs, settings)
    assert batch(items, len(files), False) == 1
    for i, (f, o, (m, a, c)) in enumerate(zip(files, outs, settings)):
        if i == 7:
            assert items[i].status == 2                             # DXTLT_FF_INVALID_INPUT_HEADER
            continue
        assert items[i].status == 0
        want = np.zeros_like(f)
        assert lib.dxtlt_dds_transform(f.ctypes.data, f.size, want.ctypes.data, want.size, m, a, c) == 0
        assert np.array_equal(o, want), i
    # and back: settings from every file's own TransformHeader
    backs = [np.zeros_like(f) for f in files]
    inv = make_items(outs, backs, [(0, False, False)] * len(files))
    failed = batch(inv, len(files), True)
    # the all-zero "file" reads as TransformFormat::Bc1 with an empty payload: whatever the single call says, the batch says
    single = np.zeros_like(outs[7])
    assert inv[7].status == lib.dxtlt_dds_untransform(outs[7].ctypes.data, outs[7].size, single.ctypes.data, single.size)
    assert failed == (1 if inv[7].status else 0) and np.array_equal(backs[7], single)
    for i, (f, b) in enumerate(zip(files, backs)):
        if i != 7:
            assert inv[i].status == 0 and np.array_equal(b, f), i


@pytest.mark.gpu
def test_dds_batch_handles_bc7_files_when_enabled(lib, oracle):
    batch = bind_batch(lib)
    lib.dxtlt_file_formats_enable_bc7.argtypes, lib.dxtlt_file_formats_enable_bc7.restype = [C.c_bool], None
    files = [dds_file("bc7"), dds_file("bc1"), dds_file("bc7")]
    outs = [np.zeros_like(f) for f in files]
    items = make_items(files, outs, [(1, False, True)] * 3)
    assert batch(items, 3, False) == 2 and [it.status for it in items] == [4, 0, 4]      # refused as upstream refuses them
    lib.dxtlt_file_formats_enable_bc7(True)
    try:
        items = make_items(files, outs, [(1, False, True)] * 3)
        assert batch(items, 3, False) == 0
        assert struct.unpack("<I", outs[0][:4].tobytes())[0] == BC7_PRIVATE_HEADER
        backs = [np.zeros_like(f) for f in files]
        assert batch(make_items(outs, backs, [(0, False, False)] * 3), 3, True) == 0
        assert all(np.array_equal(b, f) for b, f in zip(backs, files))
    finally:
        lib.dxtlt_file_formats_enable_bc7(False)
