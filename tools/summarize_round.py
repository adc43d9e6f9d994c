#!/usr/bin/env python3
"""Condenses gpurun_out/<round>_final (tools/r03_measure.sh, ROUND=r04 for round 4) into profiles/<tag>_*:

    python tools/summarize_round.py r03_z [r03_final]

  <tag>_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary of the driver's bench command, as rocprofv3 wrote it
  <tag>_kernels.json         per (kernel, grid): launches, average / min / max duration, the algorithmic bytes of one launch
                             (2 x the bytes its grid covers) and the fraction of the 8 TB/s HBM peak they give -- one row per
                             leg of the bench line, so every `roofline.frac` in it can be checked against the profiler
  <tag>_pmc.json             HBM bytes per launch of the same kernels from the FETCH_SIZE / WRITE_SIZE passes (KiB counters;
                             FETCH_SIZE doubled: on gfx950 it tallies 128-byte requests at 64 bytes, MI355X_MICROARCH.md "HBM")
  <tag>_bench_*.json         the bench lines
and refreshes profiles/pmc_traffic.json (the figure bench.py quotes as roofline.traffic, with its source)."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", sys.argv[2] if len(sys.argv) > 2 else "r03_final")
out = os.path.join(ROOT, "profiles")
PEAK = 8e12


def one(pattern):
    hits = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    if not hits:
        raise SystemExit(f"missing {pattern}")
    return hits[-1]


def corpus_bytes():
    """block bytes of the corpus legs of the profiled bench line (the batch kernel's grid holds edge tiles: its bytes are the leg's)"""
    try:
        with open(os.path.join(src, "prof_kt.json")) as f:
            legs = json.loads([l for l in f if l.startswith('{"metric"')][0]).get("legs", {})
        return {1: legs.get("corpus", {}).get("bytes"), 3: legs.get("corpus_bc3", {}).get("bytes")}
    except (OSError, ValueError, IndexError):
        return {}


CORPUS = None


def launch_bytes(name, grid_threads, wg):
    """algorithmic bytes of one launch: read + write of the blocks its grid covers; None for kernels that are not a leg"""
    global CORPUS
    wgs = grid_threads // wg
    m = re.search(r"batch_kernel<(\d),", name)
    if m:
        if CORPUS is None:
            CORPUS = corpus_bytes()
        b = CORPUS.get(int(m.group(1)))
        # the corpus legs' launches: one 4 KiB tile per workgroup but for a buffer's last one
        return 2 * b if b and abs(wgs * 4096 - b) < b // 100 else None
    if re.search(r"(fwd|inv)_tiled<", name):
        return 2 * wgs * wg * 16            # one 16-byte vector per lane
    if re.search(r"bc7_(forward|inverse)<", name):
        return 2 * wgs * 1024 * 16          # one granule of 1024 blocks per workgroup
    return None


def short(name):
    return name.split("(")[0].replace("void ", "").replace("dxtlt::", "")


rows = collections.defaultdict(list)
with open(one("prof_kt/**/*_kernel_trace.csv")) as f:
    for r in csv.DictReader(f):
        g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        wg = int(r["Workgroup_Size_X"])
        rows[(short(r["Kernel_Name"]), g, wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
kernels = []
for (name, g, wg), d in sorted(rows.items()):
    b = launch_bytes(name, g, wg)
    if b is None or b < (1 << 28) or len(d) < 5:        # the bench's small check launches are not legs
        continue
    avg = sum(d) / len(d)
    med = sorted(d)[len(d) // 2]
    # the average covers EVERY launch of the kernel at this grid, the untimed warm-up ones too (for the legs: 100 ms of them, the
    # first of which run on a chip that is still ramping its clocks); the median is what the timed steps see
    kernels.append({"kernel": name, "grid_threads": g, "workgroup": wg, "launches": len(d), "average_ns": round(avg),
                    "median_ns": med, "min_ns": min(d), "max_ns": max(d), "algorithmic_bytes_per_launch": b,
                    "frac_of_8TBps": round(b / (avg * 1e-9) / PEAK, 4), "frac_of_8TBps_median": round(b / (med * 1e-9) / PEAK, 4)})
json.dump({"source": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 --host-array-gib 0 --no-cpu-baseline",
           "kernels": kernels}, open(os.path.join(out, f"{tag}_kernels.json"), "w"), indent=1)
shutil.copy(one("prof_kt/**/*_kernel_stats.csv"), os.path.join(out, f"{tag}_kernel_stats.csv"))

pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for d, counter in (("prof_fetch", "FETCH_SIZE"), ("prof_write", "WRITE_SIZE")):
    with open(one(f"{d}/**/*_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            g, wg = int(r["Grid_Size"]), int(r["Workgroup_Size"])
            name = short(r["Kernel_Name"])
            if (launch_bytes(name, g, wg) or 0) >= (1 << 28):
                pmc[(name, g, wg)][counter].append(float(r["Counter_Value"]))
traffic = []
for (name, g, wg), c in sorted(pmc.items()):
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    rd = 2 * 1024 * sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
    wr = 1024 * sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
    alg = launch_bytes(name, g, wg)
    traffic.append({"kernel": name, "grid_threads": g, "read_bytes": round(rd), "written_bytes": round(wr),
                    "hbm_bytes_per_launch": round(rd + wr), "algorithmic_bytes_per_launch": alg,
                    "ratio": round((rd + wr) / alg, 6), "launches_sampled": len(c["FETCH_SIZE"])})
json.dump({"source": "separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over python3 bench.py --steps 3 --warmup 1 "
                     "--leg-steps 2 --no-cpu-baseline --host-array-gib 0; FETCH_SIZE x 2 x 1024, WRITE_SIZE x 1024",
           "kernels": traffic}, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1)

for f in glob.glob(os.path.join(src, "bench_*.json")):
    if os.path.getsize(f):
        shutil.copy(f, os.path.join(out, f"{tag}_{os.path.basename(f)}"))
shutil.copy(os.path.join(src, "prof_kt.json"), os.path.join(out, f"{tag}_bench_under_rocprof.json"))

def find(pattern, alg):
    hits = [t for t in traffic if pattern in t["kernel"] and t["algorithmic_bytes_per_launch"] == alg]
    return hits[0]["hbm_bytes_per_launch"] if hits else None


# per-launch HBM bytes of the legs' kernels (bench.py attaches them to `legs.*.roofline.traffic`, labelled as a committed pass)
CB = CORPUS or corpus_bytes()
leg_traffic = {
    "bc3": {"bytes": 8 << 30, "fwd": find("fwd_tiled<3, 1, true, true", 2 * (8 << 30)), "inv": find("inv_tiled<3, 1, true, true", 2 * (8 << 30))},
    "bc2": {"bytes": 8 << 30, "fwd": find("fwd_tiled<2, 1, false, true", 2 * (8 << 30)), "inv": find("inv_tiled<2, 1, false, true", 2 * (8 << 30))},
    "bc7": {"bytes": 4 << 30, "fwd": find("bc7_forward<", 2 * (4 << 30)), "inv": find("bc7_inverse<", 2 * (4 << 30))},
    "corpus": {"bytes": CB.get(1), "fwd": find("batch_kernel<1, 1, false, true, false, 256>", 2 * (CB.get(1) or 0)),
               "inv": find("batch_kernel<1, 1, false, true, true, 256>", 2 * (CB.get(1) or 0))},
    "corpus_bc3": {"bytes": CB.get(3), "fwd": find("batch_kernel<3, 1, true, true, false, 256>", 2 * (CB.get(3) or 0)),
                   "inv": find("batch_kernel<3, 1, true, true, true, 256>", 2 * (CB.get(3) or 0))},
    "archive_texture": {"bytes": 256 << 20,
                        "bc1_fwd": find("fwd_tiled<1, 1, false, true", 2 * (256 << 20)), "bc1_inv": find("inv_tiled<1, 1, false, true", 2 * (256 << 20)),
                        "bc3_fwd": find("fwd_tiled<3, 1, true, true", 2 * (256 << 20)), "bc3_inv": find("inv_tiled<3, 1, true, true", 2 * (256 << 20))},
}
head = {}
for t in traffic:
    if "fwd_tiled<1, 1, false, true" in t["kernel"] and t["algorithmic_bytes_per_launch"] == 2 * (8 << 30):
        head["fwd"] = t
    if "inv_tiled<1, 1, false, true" in t["kernel"] and t["algorithmic_bytes_per_launch"] == 2 * (8 << 30):
        head["inv"] = t
if len(head) == 2:
    json.dump({"format": "bc1", "workload_bytes": 8 << 30, "source": f"profiles/{tag}_pmc.json",
               "fwd_hbm_bytes_per_launch": head["fwd"]["hbm_bytes_per_launch"],
               "inv_hbm_bytes_per_launch": head["inv"]["hbm_bytes_per_launch"], "legs": leg_traffic},
              open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
for k in kernels:
    print(f"{k['kernel']:60s} grid {k['grid_threads']:>11d} x{k['launches']:<3d} avg {k['average_ns'] / 1e6:7.3f} ms  frac {k['frac_of_8TBps']:.4f}  median {k['median_ns'] / 1e6:7.3f} ms  frac {k['frac_of_8TBps_median']:.4f}")
for t in traffic:
    print(f"{t['kernel']:60s} grid {t['grid_threads']:>11d} HBM / algorithmic = {t['ratio']:.5f}")
