#!/usr/bin/env python3
"""Prints DESIGN.md section 5's table from profiles/<tag>_bench_bc1.json, <tag>_kernels.json and <tag>_pmc.json, so that the
document quotes what the files hold:   python tools/design_table.py r04_z"""
import json, os, sys
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04_z"
line = json.loads([l for l in open(os.path.join(P, f"{tag}_bench_bc1.json")) if l.startswith('{"metric"')][0])
kern = json.load(open(os.path.join(P, f"{tag}_kernels.json")))["kernels"]
pmc = json.load(open(os.path.join(P, f"{tag}_pmc.json")))["kernels"]

def prof(prefix, threads):
    """median ms (frac) of the profiler row of a kernel at a grid"""
    for r in kern:
        if r["kernel"].startswith(prefix) and r["grid_threads"] == threads:
            return f"{r['median_ns'] / 1e6:.3f} ({r['frac_of_8TBps_median']:.3f})"
    return "—"

def ratio(prefix, threads):
    for r in pmc:
        if r["kernel"].startswith(prefix) and r["grid_threads"] == threads:
            return f"{r['ratio']:.5f}".rstrip("0").rstrip(".") if abs(r["ratio"] - 1) > 5e-4 else f"{r['ratio']:.5f}"
    return "—"

rows = [
    ("**configs[1] BC1 8 GiB (headline)**", None, "fwd_tiled<1,", "inv_tiled<1,", 536870912),
    ("configs[2] BC3 8 GiB", "bc3", "fwd_tiled<3,", "inv_tiled<3,", 536870912),
    ("BC2 8 GiB", "bc2", "fwd_tiled<2,", "inv_tiled<2,", 536870912),
    ("configs[3] BC7 4 GiB, modes uniform", "bc7_uniform", "bc7::bc7_forward", "bc7::bc7_inverse", 67108864),
    ("configs[3] BC7 4 GiB, modes skewed", "bc7_skewed", None, None, 0),
    ("configs[4] share: 32 × 256 MiB BC1 / BC3", "archive", None, None, 0),
    ("**corpus, BC1**: 2130 textures, one batch call", "corpus", "batch_kernel<1, 1, false, true, false, 256>", "batch_kernel<1, 1, false, true, true, 256>", 569927680),
    ("corpus, BC3: 1065 textures", "corpus_bc3", "batch_kernel<3, 1, true, true, false, 256>", "batch_kernel<3, 1, true, true, true, 256>", 549961728),
]
print("| config | `value` GiB/s | forward ms / frac | inverse ms / frac | rocprofv3 median ms (frac) | PMC traffic ÷ algorithmic |")
print("|---|---|---|---|---|---|")
for name, leg, kf, ki, threads in rows:
    v = line if leg is None else line["legs"][leg]
    c = v["config"] if leg is None else v
    r = v["roofline"]
    p = f"{prof(kf, threads)} / {prof(ki, threads)}" if kf else ("(same kernels)" if leg == "bc7_skewed" else "per texture: the 16 777 216-thread rows")
    t = f"{ratio(kf, threads)} / {ratio(ki, threads)}" if kf else "—"
    print(f"| {name} | {v['value']:.0f} | {c['fwd_ms']:.3f} / {r['frac']:.3f} | {c['inv_ms']:.3f} / {r['inverse_kernel']['frac']:.3f} | {p} | {t} |")
