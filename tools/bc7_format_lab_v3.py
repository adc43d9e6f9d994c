"""Format-lab trial of round 3: BC7 records that keep the fields' pitch (every output word = shifted channel words merged under masks --
about one device instruction per two fields instead of four) against version 2, on the reference's BC7 texture and on synthetic ones
(tools/bc7_synth.py).  Result: profiles/r03_bc7_experiments.txt section 4 -- the cheap records give back most of version 2's gain.
    python tools/bc7_synth.py 1024 /tmp/bc7_synth_1024.bin && python tools/bc7_format_lab_v3.py /tmp/bc7_synth_1024.bin"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bc7_format_lab as L  # noqa: E402
import numpy as np

def decor(f, how):
    ch = {c: [x[0] for x in f[c]] for c in "RGBA" if c in f}
    w = f["R"][0][1]; M = (1 << w) - 1
    if how == "sub":
        ch["R"] = [(r - g) & M for r, g in zip(ch["R"], ch["G"])]
        ch["B"] = [(b - g) & M for b, g in zip(ch["B"], ch["G"])]
    elif how == "xor":
        ch["R"] = [(r ^ g) & M for r, g in zip(ch["R"], ch["G"])]
        ch["B"] = [(b ^ g) & M for b, g in zip(ch["B"], ch["G"])]
    return ch, w

def bits(v, lo, n):
    return ((v >> lo) & ((1 << n) - 1), n)

# slot recipes: list of words; each word = list of (channel, lo, n) LSB first within the slot
def recipe_a(w):
    k = w - 4
    if k == 0:
        return None
    H1 = [("G", 4, k), ("R", k, 4)]
    H2 = [("G", w - 2 * k, k), ("B", k, 4)]
    Lw = [("G", 0, w - 2 * k), ("R", 0, k), ("B", 0, k)]
    return {"L": [Lw], "H": [H2, H1]}

def recipe_b(w):
    # G top4 kept together: H1 = R top4 | G top k ; H2 = B top4 | G next... variant: highs words hold top (w-?)..
    k = w - 4
    if k == 0:
        return None
    # H1 = G top4 + R top k ; H2 = B top4 + R next k ; L = R low (w-2k), G low k, B low k
    H1 = [("R", 4, k), ("G", k, 4)]
    H2 = [("R", w - 2 * k, k), ("B", k, 4)]
    Lw = [("R", 0, w - 2 * k), ("G", 0, k), ("B", 0, k)]
    return {"L": [Lw], "H": [H2, H1]}

def make_order(recipe_fn, how="sub", alpha="whole_hi", hi_order="word", tail_first=True):
    def order(m, f):
        hdr, idx, ep, pb = L.split(m, f)
        ch, w = decor(f, how)
        n = len(ch["R"])
        rec = recipe_fn(w)
        lows, highs = [], []
        if rec is None:
            highs = [(v, w) for c in "RGB" for v in ch[c]]
        else:
            for word in rec["L"]:
                for i in range(n):
                    lows += [bits(ch[c][i], lo, nb) for c, lo, nb in word]
            for word in rec["H"]:
                for i in range(n):
                    highs += [bits(ch[c][i], lo, nb) for c, lo, nb in word]
        if "A" in ch:
            wa = f["A"][0][1]
            if alpha == "whole_hi":
                highs = highs + [(v, wa) for v in ch["A"]]
            elif alpha == "whole_mid":
                lows = lows + [(v, wa) for v in ch["A"]]
            elif alpha == "whole_first_hi":
                highs = [(v, wa) for v in ch["A"]] + highs
            elif alpha == "split":
                lows = lows + [bits(v, 0, wa - 4) for v in ch["A"] if wa > 4]
                highs = highs + [bits(v, wa - 4, 4) for v in ch["A"]]
        return hdr + pb + idx + lows + highs
    return order

def order_v2_xor(m, f):
    hdr, idx, ep, pb = L.split(m, f)
    ch, w = decor(f, "xor")
    ep = [(v, f[c][0][1]) for c in "RGBA" if c in ch for v in ch[c]]
    return hdr + pb + idx + [L.lo(x, x[1] - 4) for x in ep if x[1] > 4] + [L.hi(x, 4) for x in ep]

def main():
    corpora = {"ref": os.path.join(ROOT, "tests", "golden", "r2-256-bc7.payload.bin")}
    for p in sys.argv[1:]:
        corpora[os.path.basename(p)] = p
    slots = (8, 2, 1, 1, 1, 1, 1)
    for name, path in corpora.items():
        c = L.load(path)
        base = L.comp(c[0])
        print("==", name, base)
        def row(label, data):
            r = L.comp(data)
            print(f"  {label:60s} " + "  ".join(f"{k} {100 * (v / base[k] - 1):+5.1f}%" for k, v in r.items()), flush=True)
        row("v2", L.assemble(c, L.order_v2, slots))
        row("v2 with xor", L.assemble(c, order_v2_xor, slots))
        for rn, rf in (("A", recipe_a), ("B", recipe_b)):
            for how in ("sub", "xor"):
                for alpha in ("whole_hi", "whole_mid", "whole_first_hi", "split"):
                    row(f"v3 recipe {rn} {how} alpha={alpha}", L.assemble(c, make_order(rf, how, alpha), slots))
main()
