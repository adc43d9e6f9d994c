#!/usr/bin/env python3
"""BC7 format lab: compressed size of candidate stream layouts on the reference's BC7 test texture and on a synthetic
photo-like texture (tools/bc7_synth.py) -- the table of docs/BC7_FORMAT.md section 4.  CPU only, test tooling.

    python tools/bc7_synth.py 1024 /tmp/bc7_synth_1024.bin && python tools/bc7_format_lab.py [/tmp/bc7_synth_1024.bin]

Every candidate is: per-block record = marker + an ordering of the block's bit fields; records cut into byte slots of
fixed widths; slot streams over the blocks sorted by mode inside granules of T blocks (or over global per-mode streams).
"""
import lzma
import os
import subprocess
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle_c, oracle_np  # noqa: E402

F = oracle_np.BC7_FIELDS
HDR = ("part", "rot", "sel")
ZSTD = "/opt/conda/bin/zstd"


def comp(data) -> dict:
    data = bytes(data)
    r = {"zlib6": len(zlib.compress(data, 6)), "lzma": len(lzma.compress(data, preset=6))}
    if os.path.exists(ZSTD):
        for lvl in (3, 19):
            r[f"zstd{lvl}"] = len(subprocess.run([ZSTD, f"-{lvl}", "-c", "--no-progress"], input=data, capture_output=True).stdout)
    return r


def load(path):
    p = np.fromfile(path, dtype=np.uint8)
    blocks = [int.from_bytes(bytes(p[i:i + 16]), "little") for i in range(0, len(p), 16)]
    modes = [int(m) for m in oracle_np.bc7_modes(p.reshape(-1, 16)[:, 0])]
    parsed = []
    for b, m in zip(blocks, modes):
        f = {}
        if m < 8:
            for name, v, w in oracle_np._bc7_parse(b, m):
                f.setdefault(name, []).append((v, w))
        parsed.append(f)
    return p, blocks, modes, parsed


def split(m, f):
    names = [n for n, _, _ in F[m]]
    hdr = [x for n in names if n in HDR for x in f[n]]
    idx = [x for n in names if n.startswith("idx") for x in f[n]]
    ep = [x for c in "RGBA" if c in f for x in f[c]]
    return hdr, idx, ep, list(f.get("P", []))


def hi(x, k):
    return (x[0] >> (x[1] - k), k)


def lo(x, k):
    return (x[0] & ((1 << k) - 1), k)


def order_natural(m, f):
    return [x for n, _, _ in F[m] for x in f[n]]


def order_hdr_p_idx_ep(m, f):
    hdr, idx, ep, pb = split(m, f)
    return hdr + pb + idx + ep


def order_hilo(m, f):                       # version 1
    hdr, idx, ep, pb = split(m, f)
    return hdr + pb + idx + [lo(x, x[1] - 4) for x in ep if x[1] > 4] + [hi(x, 4) for x in ep]


def green_out(f, how="sub"):
    """colour decorrelation candidates on the endpoint fields of one block: 'sub' (version 2) = red and blue as differences
    to the green of the same endpoint, modulo the field width; 'ycocg' = reversible YCoCg-R modulo the field width;
    'delta' = the second endpoint of every subset as a difference to the first"""
    ch = {c: [list(x) for x in f[c]] for c in "RGBA" if c in f}
    n, w = len(ch["R"]), ch["R"][0][1]
    M = (1 << w) - 1
    if how == "delta":
        for c in ch:
            wc = ch[c][0][1]
            for k in range(0, n, 2):
                ch[c][k + 1][0] = (ch[c][k + 1][0] - ch[c][k][0]) & ((1 << wc) - 1)
    for i in range(n):
        r, g, b = ch["R"][i][0], ch["G"][i][0], ch["B"][i][0]
        if how == "sub":
            ch["R"][i][0], ch["B"][i][0] = (r - g) & M, (b - g) & M
        elif how == "ycocg":
            co = (r - b) & M
            t = (b + (co >> 1)) & M
            cg = (g - t) & M
            ch["R"][i][0], ch["G"][i][0], ch["B"][i][0] = co, (t + (cg >> 1)) & M, cg
    return [tuple(x) for c in "RGBA" if c in ch for x in ch[c]]


def order_v2(m, f):                         # version 2: version 1's record over green-decorrelated endpoints
    hdr, idx, ep, pb = split(m, f)
    ep = green_out(f, "sub")
    return hdr + pb + idx + [lo(x, x[1] - 4) for x in ep if x[1] > 4] + [hi(x, 4) for x in ep]


def order_hilo_with(how):
    def order(m, f):
        hdr, idx, ep, pb = split(m, f)
        ep = green_out(f, how)
        return hdr + pb + idx + [lo(x, x[1] - 4) for x in ep if x[1] > 4] + [hi(x, 4) for x in ep]
    return order


def order_v2_high_nibbles_only(m, f):       # green taken out of the high nibbles only (cheaper on the device)
    hdr, idx, ep, pb = split(m, f)
    ch = {c: list(f[c]) for c in "RGBA" if c in f}
    highs = {c: [hi(x, 4)[0] for x in ch[c]] for c in ch}
    highs["R"] = [(a - g) & 15 for a, g in zip(highs["R"], highs["G"])]
    highs["B"] = [(a - g) & 15 for a, g in zip(highs["B"], highs["G"])]
    return hdr + pb + idx + [lo(x, x[1] - 4) for x in ep if x[1] > 4] + [(v, 4) for c in "RGBA" if c in highs for v in highs[c]]


def order_hilo_by_channel(m, f):
    hdr, idx, ep, pb = split(m, f)
    ch = [c for c in "RGBA" if c in f]
    highs = [hi(f[c][i], 4) for i in range(len(f["R"])) for c in ch]
    return hdr + pb + idx + [lo(x, x[1] - 4) for x in ep if x[1] > 4] + highs


def order_hilo_wide_only(m, f):             # no split for modes with endpoints of 5 bits or fewer
    hdr, idx, ep, pb = split(m, f)
    if f["R"][0][1] < 6:
        return hdr + pb + idx + ep
    return order_hilo(m, f)


def assemble(corpus, order_fn, slots, T=1024, placement="granule"):
    p, blocks, modes, parsed = corpus
    n = len(blocks)
    recs = []
    for b, m, f in zip(blocks, modes, parsed):
        if m == 8:
            recs.append((b & 0xFF, b >> 8))
            continue
        v, at = 0, 0
        for val, w in order_fn(m, f):
            v |= val << at
            at += w
        assert at == 127 - m
        full = (1 << m) | (v << (m + 1))
        recs.append((full & 0xFF, full >> 8))
    out = bytearray()
    if placement == "granule":
        order = []
        for s in range(0, n, T):
            idx = list(range(s, min(n, s + T)))
            idx.sort(key=lambda i: modes[i])
            order += idx
        at = 0
        for w in slots:
            for i in order:
                out += ((recs[i][1] >> at) & ((1 << (8 * w)) - 1)).to_bytes(w, "little")
            at += 8 * w
        out += bytes(r[0] for r in recs)
    else:   # global per-mode streams
        out += bytes(r[0] for r in recs)
        for m in range(9):
            sel = [r[1] for r, mm in zip(recs, modes) if mm == m]
            at = 0
            for w in slots:
                for r in sel:
                    out += ((r >> at) & ((1 << (8 * w)) - 1)).to_bytes(w, "little")
                at += 8 * w
    assert len(out) == len(p)
    return out


def version0(corpus):
    """round 1's format: byte 0 stream, then per mode a head and a tail byte stream (global placement)"""
    head = [9, 9, 11, 11, 5, 7, 7, 11, 15]
    p, blocks, modes, _ = corpus
    out = bytearray(b & 0xFF for b in blocks)
    for m in range(9):
        hs, ts = bytearray(), bytearray()
        for b, mm in zip(blocks, modes):
            if mm == m:
                by = b.to_bytes(16, "little")
                hs += by[1:1 + head[m]]
                ts += by[1 + head[m]:]
        out += hs + ts
    return out


def main():
    corpora = {"reference r2-256-bc7 (4096 blocks)": os.path.join(ROOT, "tests", "golden", "r2-256-bc7.payload.bin")}
    for extra in sys.argv[1:]:
        corpora[os.path.basename(extra)] = extra
    v1_slots = (8, 2, 1, 1, 1, 1, 1)
    for name, path in corpora.items():
        c = load(path)
        base = comp(c[0])
        print(f"== {name}: untransformed {base}")

        def row(label, data):
            r = comp(data)
            print(f"  {label:64s} " + "  ".join(f"{k} {100 * (v / base[k] - 1):+5.1f}%" for k, v in r.items()), flush=True)

        row("version 0 (bytes by mode, global head / tail streams)", version0(c))
        row("version 1 = hi/lo records, slots 8+2+1x5, granule 1024", assemble(c, order_hilo, v1_slots, 1024))
        v2 = oracle_c.transform_bc7(c[0])
        assert bytes(v2) == bytes(assemble(c, order_v2, v1_slots, 1024)), "lab statement of version 2 == the oracle"
        row("VERSION 2 = version 1 + red and blue minus green (oracle)", v2)
        row("  same records, global per-mode streams", assemble(c, order_v2, v1_slots, placement="permode"))
        for T in (256, 2048, 4096):
            row(f"  same records, granule {T}", assemble(c, order_v2, v1_slots, T))
        row("  green out of the high nibbles only", assemble(c, order_v2_high_nibbles_only, v1_slots))
        row("  YCoCg-R modulo the field width instead", assemble(c, order_hilo_with("ycocg"), v1_slots))
        row("  second endpoint of a subset minus the first instead", assemble(c, order_hilo_with("delta"), v1_slots))
        row("  version 1 records: no hi/lo split for endpoints of <= 5 bits", assemble(c, order_hilo_wide_only, v1_slots))
        row("  version 1 records: high nibbles grouped by channel", assemble(c, order_hilo_by_channel, v1_slots))
        for slots in ((8, 4, 2, 1), (8, 4, 1, 1, 1), (4, 4, 4, 2, 1), (4, 4, 2, 2, 1, 1, 1), (15,)):
            row(f"  version 1 records, slots {slots}", assemble(c, order_hilo, slots))
        row("fields in block order, slots 8+2+1x5", assemble(c, order_natural, v1_slots))
        row("header, p-bits, indices, endpoints whole, slots 8+2+1x5", assemble(c, order_hdr_p_idx_ep, v1_slots))


if __name__ == "__main__":
    main()
