"""Second, independently written restatement of the BCn transform in numpy.

TEST INFRASTRUCTURE ONLY (see oracle/dxtlt_oracle.h).  Where the C oracle walks one block at a
time like the reference's scalar loops, this file states the *format*: it views the block array as
a structured table and writes each output stream as one slice.  The two are written separately on
purpose; tests/test_oracle.py requires them to agree byte-for-byte, and
tests/golden/make_golden.py uses this file to emit the committed fixtures.

Format statement (N = number of blocks, little-endian fields):
  BC1  block {c0:u16, c1:u16, idx:u32}
       out = [colours 4N][idx 4N]; colours = N x (c0',c1') pairs, or all c0' then all c1' when split
       (reference: dxt-lossless-transform-bc1/src/transform/transform_with_settings.rs:39-71)
  BC2  block {alpha:u64, c0, c1, idx}
       out = [alpha 8N][colours 4N][idx 4N]            (bc2 .../transform_with_settings.rs:30-73)
  BC3  block {a0:u8, a1:u8, aidx:6B, c0, c1, idx}
       out = [alpha endpoints 2N][aidx 6N][colours 4N][idx 4N]; alpha endpoints = (a0,a1) pairs, or
       all a0 then all a1 when split                   (bc3 .../transform_with_settings.rs:32-142)
  c' = YCoCg-R(c) in 5-bit modular arithmetic (common/src/color_565/decorrelate.rs:101-344).
"""
from __future__ import annotations

import numpy as np

NONE, VAR1, VAR2, VAR3 = 0, 1, 2, 3

BC1 = np.dtype([("c0", "<u2"), ("c1", "<u2"), ("idx", "<u4")])
BC2 = np.dtype([("alpha", "<u8"), ("c0", "<u2"), ("c1", "<u2"), ("idx", "<u4")])
BC3 = np.dtype([("a0", "u1"), ("a1", "u1"), ("aidx", "u1", (6,)), ("c0", "<u2"), ("c1", "<u2"), ("idx", "<u4")])
assert BC1.itemsize == 8 and BC2.itemsize == 16 and BC3.itemsize == 16


def decorrelate(c: np.ndarray, variant: int) -> np.ndarray:
    """Vectorised Color565::decorrelate_ycocg_r on a u16 array."""
    if variant == NONE:
        return c.astype(np.uint16)
    v = c.astype(np.int32)
    r, g, gl, b = (v >> 11) & 31, (v >> 6) & 31, (v >> 5) & 1, v & 31
    co = (r - b) % 32
    t = (b + co // 2) % 32
    cg = (g - t) % 32
    y = (t + cg // 2) % 32
    if variant == VAR1:
        o = (y << 11) | (co << 6) | (gl << 5) | cg
    elif variant == VAR2:
        o = (gl << 15) | (y << 10) | (co << 5) | cg
    elif variant == VAR3:
        o = (y << 11) | (co << 6) | (cg << 1) | gl
    else:
        raise ValueError(variant)
    return o.astype(np.uint16)


def recorrelate(c: np.ndarray, variant: int) -> np.ndarray:
    if variant == NONE:
        return c.astype(np.uint16)
    v = c.astype(np.int32)
    if variant == VAR1:
        y, co, gl, cg = (v >> 11) & 31, (v >> 6) & 31, (v >> 5) & 1, v & 31
    elif variant == VAR2:
        gl, y, co, cg = (v >> 15) & 1, (v >> 10) & 31, (v >> 5) & 31, v & 31
    elif variant == VAR3:
        y, co, cg, gl = (v >> 11) & 31, (v >> 6) & 31, (v >> 1) & 31, v & 1
    else:
        raise ValueError(variant)
    t = (y - cg // 2) % 32
    g = (cg + t) % 32
    b = (t - co // 2) % 32
    r = (b + co) % 32
    return ((r << 11) | (g << 6) | (gl << 5) | b).astype(np.uint16)


def _u8(x) -> np.ndarray:
    a = np.frombuffer(x, dtype=np.uint8) if isinstance(x, (bytes, bytearray, memoryview)) else np.asarray(x)
    assert a.dtype == np.uint8 and a.ndim == 1
    return np.ascontiguousarray(a)


def _colour_section(c0: np.ndarray, c1: np.ndarray, split: bool) -> np.ndarray:
    if split:
        return np.concatenate([c0.astype("<u2").view(np.uint8), c1.astype("<u2").view(np.uint8)])
    pairs = np.empty((c0.size, 2), dtype="<u2")
    pairs[:, 0], pairs[:, 1] = c0, c1
    return pairs.reshape(-1).view(np.uint8)


def _colour_unsection(sec: np.ndarray, n: int, split: bool):
    w = sec.view("<u2")
    if split:
        return w[:n].copy(), w[n:].copy()
    p = w.reshape(n, 2)
    return p[:, 0].copy(), p[:, 1].copy()


def transform_bc1(data, variant=VAR1, split_colour=True) -> np.ndarray:
    blk = _u8(data).view(BC1)
    c0, c1 = decorrelate(blk["c0"], variant), decorrelate(blk["c1"], variant)
    return np.concatenate([_colour_section(c0, c1, split_colour), blk["idx"].astype("<u4").view(np.uint8)])


def untransform_bc1(data, variant=VAR1, split_colour=True) -> np.ndarray:
    a = _u8(data)
    n = a.size // 8
    out = np.empty(n, dtype=BC1)
    c0, c1 = _colour_unsection(a[: 4 * n], n, split_colour)
    out["c0"], out["c1"] = recorrelate(c0, variant), recorrelate(c1, variant)
    out["idx"] = a[4 * n:].view("<u4")
    return out.view(np.uint8)


def transform_bc2(data, variant=VAR1, split_colour=True) -> np.ndarray:
    blk = _u8(data).view(BC2)
    c0, c1 = decorrelate(blk["c0"], variant), decorrelate(blk["c1"], variant)
    return np.concatenate([
        blk["alpha"].astype("<u8").view(np.uint8),
        _colour_section(c0, c1, split_colour),
        blk["idx"].astype("<u4").view(np.uint8),
    ])


def untransform_bc2(data, variant=VAR1, split_colour=True) -> np.ndarray:
    a = _u8(data)
    n = a.size // 16
    out = np.empty(n, dtype=BC2)
    out["alpha"] = a[: 8 * n].view("<u8")
    c0, c1 = _colour_unsection(a[8 * n: 12 * n], n, split_colour)
    out["c0"], out["c1"] = recorrelate(c0, variant), recorrelate(c1, variant)
    out["idx"] = a[12 * n:].view("<u4")
    return out.view(np.uint8)


def transform_bc3(data, variant=VAR1, split_alpha=True, split_colour=True) -> np.ndarray:
    blk = _u8(data).view(BC3)
    if split_alpha:
        alpha = np.concatenate([blk["a0"], blk["a1"]])
    else:
        alpha = np.stack([blk["a0"], blk["a1"]], axis=1).reshape(-1)
    c0, c1 = decorrelate(blk["c0"], variant), decorrelate(blk["c1"], variant)
    return np.concatenate([
        np.ascontiguousarray(alpha),
        np.ascontiguousarray(blk["aidx"]).reshape(-1),
        _colour_section(c0, c1, split_colour),
        blk["idx"].astype("<u4").view(np.uint8),
    ])


def untransform_bc3(data, variant=VAR1, split_alpha=True, split_colour=True) -> np.ndarray:
    a = _u8(data)
    n = a.size // 16
    out = np.empty(n, dtype=BC3)
    if split_alpha:
        out["a0"], out["a1"] = a[:n], a[n: 2 * n]
    else:
        p = a[: 2 * n].reshape(n, 2)
        out["a0"], out["a1"] = p[:, 0], p[:, 1]
    out["aidx"] = a[2 * n: 8 * n].reshape(n, 6)
    c0, c1 = _colour_unsection(a[8 * n: 12 * n], n, split_colour)
    out["c0"], out["c1"] = recorrelate(c0, variant), recorrelate(c1, variant)
    out["idx"] = a[12 * n:].view("<u4")
    return out.view(np.uint8)


def transform(fmt: str, data, variant=VAR1, split_colour=True, split_alpha=True, inverse=False) -> np.ndarray:
    if fmt == "bc1":
        return (untransform_bc1 if inverse else transform_bc1)(data, variant, split_colour)
    if fmt == "bc2":
        return (untransform_bc2 if inverse else transform_bc2)(data, variant, split_colour)
    if fmt == "bc3":
        return (untransform_bc3 if inverse else transform_bc3)(data, variant, split_alpha, split_colour)
    raise ValueError(fmt)


def splitmix64(seed: int, first_qword: int, count: int) -> np.ndarray:
    """qword i = mix(seed + (first_qword + i + 1) * GOLDEN) -- same stream as oracle_fill_splitmix64."""
    m = np.uint64(0xFFFFFFFFFFFFFFFF)
    with np.errstate(over="ignore"):
        idx = np.arange(first_qword + 1, first_qword + 1 + count, dtype=np.uint64)
        z = (np.uint64(seed & int(m)) + idx * np.uint64(0x9E3779B97F4A7C15)) & m
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & m
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & m
        return z ^ (z >> np.uint64(31))


# ------------------------------------------------------------------------------------------------------------
# BC7 granule-sorted field split, version 2 (docs/BC7_FORMAT.md) -- a format defined by this build; parity unpinned.
# Second, independently written statement: one block at a time as a Python integer, fields by name (slow: small cases).
# ------------------------------------------------------------------------------------------------------------
BC7_GRANULE = 1024
# fields after the mode marker, LSB first: (name, count, width); hexpat:286-654
BC7_FIELDS = {
    0: [("part", 1, 4), ("R", 6, 4), ("G", 6, 4), ("B", 6, 4), ("P", 6, 1), ("idx", 1, 45)],
    1: [("part", 1, 6), ("R", 4, 6), ("G", 4, 6), ("B", 4, 6), ("P", 2, 1), ("idx", 1, 46)],
    2: [("part", 1, 6), ("R", 6, 5), ("G", 6, 5), ("B", 6, 5), ("idx", 1, 29)],
    3: [("part", 1, 6), ("R", 4, 7), ("G", 4, 7), ("B", 4, 7), ("P", 4, 1), ("idx", 1, 30)],
    4: [("rot", 1, 2), ("sel", 1, 1), ("R", 2, 5), ("G", 2, 5), ("B", 2, 5), ("A", 2, 6), ("idx", 1, 31), ("idx2", 1, 47)],
    5: [("rot", 1, 2), ("R", 2, 7), ("G", 2, 7), ("B", 2, 7), ("A", 2, 8), ("idx", 1, 31), ("idx2", 1, 31)],
    6: [("R", 2, 7), ("G", 2, 7), ("B", 2, 7), ("A", 2, 7), ("P", 2, 1), ("idx", 1, 63)],
    7: [("part", 1, 6), ("R", 4, 5), ("G", 4, 5), ("B", 4, 5), ("A", 4, 5), ("P", 4, 1), ("idx", 1, 30)],
}
_BC7_HEADER, _BC7_ENDPOINTS = ("part", "rot", "sel"), ("R", "G", "B", "A")


def bc7_modes(first: np.ndarray) -> np.ndarray:
    """class = trailing zero count of byte 0, 8 when byte 0 is zero (the reserved encoding)"""
    f = first.astype(np.int64)
    low = f & -f  # lowest set bit, 0 for 0
    m = np.full(f.shape, 8, dtype=np.int64)
    nz = low != 0
    m[nz] = np.log2(low[nz]).astype(np.int64)
    return m


def _bc7_parse(b: int, m: int):
    pos, out = m + 1, []
    for name, count, width in BC7_FIELDS[m]:
        for _ in range(count):
            out.append((name, (b >> pos) & ((1 << width) - 1), width))
            pos += width
    assert pos == 128
    return out


def _bc7_pack(pieces, start: int, value: int) -> int:
    at = start
    for v, w in pieces:
        value |= v << at
        at += w
    assert at == 128
    return value


def _bc7_green_out(fields, sign: int):
    """Colour decorrelation of version 2: every red and blue endpoint field becomes its difference to the green field of
    the same endpoint, modulo the field width (sign = -1), or gets it back (sign = +1).  Green and alpha stay."""
    green = [v for n, v, w in fields if n == "G"]
    out, seen = [], {"R": 0, "B": 0}
    for n, v, w in fields:
        if n in seen:
            v = (v + sign * green[seen[n]]) & ((1 << w) - 1)
            seen[n] += 1
        out.append((n, v, w))
    return out


def bc7_record_of_block(b: int, m: int) -> int:
    """marker | header | p-bits | index bits | low parts of the endpoints | high nibbles of the endpoints, the endpoints
    being R - G, G, B - G (modulo the field width) and A"""
    if m == 8:
        return b
    f = _bc7_green_out(_bc7_parse(b, m), -1)
    hdr = [(v, w) for n, v, w in f if n in _BC7_HEADER]
    pb = [(v, w) for n, v, w in f if n == "P"]
    idx = [(v, w) for n, v, w in f if n.startswith("idx")]
    ep = [(v, w) for n, v, w in f if n in _BC7_ENDPOINTS]
    lows = [(v & ((1 << (w - 4)) - 1), w - 4) for v, w in ep]
    highs = [(v >> (w - 4), 4) for v, w in ep]
    return _bc7_pack(hdr + pb + idx + lows + highs, m + 1, 1 << m)


def bc7_block_of_record(r: int, m: int) -> int:
    if m == 8:
        return r
    layout = BC7_FIELDS[m]
    widths = {"hdr": [], "pb": [], "idx": [], "ep": []}
    for name, count, width in layout:
        key = "hdr" if name in _BC7_HEADER else "pb" if name == "P" else "idx" if name.startswith("idx") else "ep"
        widths[key] += [width] * count
    pos = m + 1

    def take(w):
        nonlocal pos
        v = (r >> pos) & ((1 << w) - 1)
        pos += w
        return v

    hdr = [take(w) for w in widths["hdr"]]
    pb = [take(w) for w in widths["pb"]]
    idx = [take(w) for w in widths["idx"]]
    lows = [take(w - 4) for w in widths["ep"]]
    highs = [take(4) for _ in widths["ep"]]
    assert pos == 128
    ep = [(h << (w - 4)) | l for h, l, w in zip(highs, lows, widths["ep"])]
    # green back into red and blue (endpoint fields are in block order: all reds, all greens, all blues, alphas)
    names = [name for name, count, width in layout if name in _BC7_ENDPOINTS for _ in range(count)]
    ep = [v for _, v, _ in _bc7_green_out(list(zip(names, ep, widths["ep"])), +1)]
    # back into block order: header, endpoints, p-bits, indices
    b, at = 1 << m, m + 1
    for v, w in list(zip(hdr, widths["hdr"])) + list(zip(ep, widths["ep"])) + list(zip(pb, widths["pb"])) + list(zip(idx, widths["idx"])):
        b |= v << at
        at += w
    assert at == 128
    return b


_BC7_STREAMS = [(0, 8, 1), (8, 2, 9), (10, 1, 11), (11, 1, 12), (12, 1, 13), (13, 1, 14), (14, 1, 15)]  # (offset, width, record byte)


def _bc7_sorted_order(modes: np.ndarray) -> np.ndarray:
    """sorted position -> block index, granule by granule, stable by class"""
    order = []
    for g0 in range(0, modes.size, BC7_GRANULE):
        m = modes[g0:g0 + BC7_GRANULE]
        order.append(g0 + np.argsort(m, kind="stable"))
    return np.concatenate(order) if order else np.zeros(0, dtype=np.int64)


def _bc7_part(blk: np.ndarray) -> np.ndarray:
    n = blk.shape[0]
    modes = bc7_modes(blk[:, 0])
    recs = np.empty((n, 16), dtype=np.uint8)
    for i in range(n):
        r = bc7_record_of_block(int.from_bytes(blk[i].tobytes(), "little"), int(modes[i]))
        recs[i] = np.frombuffer(r.to_bytes(16, "little"), dtype=np.uint8)
    srt = recs[_bc7_sorted_order(modes)]
    parts = [srt[:, rb:rb + w].reshape(-1) for _, w, rb in _BC7_STREAMS]
    parts.append(recs[:, 0].copy())
    return np.concatenate(parts) if n else np.zeros(0, dtype=np.uint8)


def _bc7_unpart(a: np.ndarray) -> np.ndarray:
    n = a.size // 16
    first = a[15 * n:16 * n]
    modes = bc7_modes(first)
    order = _bc7_sorted_order(modes)
    recs = np.empty((n, 16), dtype=np.uint8)
    recs[:, 0] = first
    for off, w, rb in _BC7_STREAMS:
        recs[order, rb:rb + w] = a[off * n:(off + w) * n].reshape(n, w)
    out = np.empty((n, 16), dtype=np.uint8)
    for i in range(n):
        b = bc7_block_of_record(int.from_bytes(recs[i].tobytes(), "little"), int(modes[i]))
        out[i] = np.frombuffer(b.to_bytes(16, "little"), dtype=np.uint8)
    return out.reshape(-1)


def transform_bc7(data) -> np.ndarray:
    blk = _u8(data).reshape(-1, 16)
    main_n = blk.shape[0] - blk.shape[0] % BC7_GRANULE
    return np.ascontiguousarray(np.concatenate([_bc7_part(blk[:main_n]), _bc7_part(blk[main_n:])]))


def untransform_bc7(data) -> np.ndarray:
    a = _u8(data)
    n = a.size // 16
    main_n = n - n % BC7_GRANULE
    return np.ascontiguousarray(np.concatenate([_bc7_unpart(a[:16 * main_n]), _bc7_unpart(a[16 * main_n:])]))


# ---- BC1 block normalisation (reference experimental module), second statement: all 16 pixels, vectorised -------------
def decode_bc1_pixels(data) -> np.ndarray:
    """(N, 16) uint32 pixels r | g << 8 | b << 16 | a << 24 (util/bc1_decode.rs:42-100)."""
    b = _u8(data).reshape(-1, 8).astype(np.uint32)
    c0 = b[:, 0] | (b[:, 1] << 8)
    c1 = b[:, 2] | (b[:, 3] << 8)
    idx = b[:, 4] | (b[:, 5] << 8) | (b[:, 6] << 16) | (b[:, 7] << 24)

    def chan(c):
        r, g, bl = (c >> 11) & 31, (c >> 5) & 63, c & 31
        return (r << 3) | (r >> 2), (g << 2) | (g >> 4), (bl << 3) | (bl >> 2)

    r0, g0, b0 = chan(c0)
    r1, g1, b1 = chan(c1)
    four = c0 > c1
    opaque = np.uint32(0xFF000000)

    def pack(r, g, bl):
        return r | (g << 8) | (bl << 16) | opaque

    pal = np.empty((b.shape[0], 4), dtype=np.uint32)
    pal[:, 0] = pack(r0, g0, b0)
    pal[:, 1] = pack(r1, g1, b1)
    pal[:, 2] = np.where(four, pack((2 * r0 + r1) // 3, (2 * g0 + g1) // 3, (2 * b0 + b1) // 3),
                         pack((r0 + r1) // 2, (g0 + g1) // 2, (b0 + b1) // 2))
    pal[:, 3] = np.where(four, pack((r0 + 2 * r1) // 3, (g0 + 2 * g1) // 3, (b0 + 2 * b1) // 3), np.uint32(0))
    sel = (idx[:, None] >> (2 * np.arange(16, dtype=np.uint32))[None, :]) & 3
    return np.take_along_axis(pal, sel.astype(np.int64), axis=1)


def normalize_bc1_blocks(data, mode: int) -> np.ndarray:
    """normalize.rs:38-188, 214-258.  mode: 0 None, 1 Color0Only, 2 ReplicateColor."""
    src = _u8(data)
    out = src.copy()
    if mode == 0 or src.size == 0:
        return out
    px = decode_bc1_pixels(src)
    same = (px == px[:, :1]).all(axis=1)
    first = px[:, 0]
    transparent = same & ((first >> 24) == 0)
    r, g, b = first & 255, (first >> 8) & 255, (first >> 16) & 255
    c565 = ((r & 0xF8) << 8) | ((g & 0xFC) << 3) | (b >> 3)
    rr, gg, bb = (c565 >> 11) & 31, (c565 >> 5) & 63, c565 & 31
    back = ((rr << 3) | (rr >> 2)) | (((gg << 2) | (gg >> 4)) << 8) | (((bb << 3) | (bb >> 2)) << 16)
    solid = same & ~transparent & (back == (first & 0xFFFFFF))
    o = out.reshape(-1, 8)
    o[transparent] = 0xFF
    lo, hi = (c565 & 255).astype(np.uint8), (c565 >> 8).astype(np.uint8)
    o[solid, 0], o[solid, 1] = lo[solid], hi[solid]
    o[solid, 2] = lo[solid] if mode == 2 else 0
    o[solid, 3] = hi[solid] if mode == 2 else 0
    o[solid, 4:] = 0
    return out
