#!/usr/bin/env python3
"""Turns rocprofv3 CSV output (gpurun_out/prof_*) into the small summaries committed under profiles/.

    python tools/summarize_profiles.py <tag> [<dir under gpurun_out>]      # e.g. r02_z r02_final/bc1

Inputs (produced on the GPU box by tools/r02_measure.sh, see profiles/README.md for the exact commands), under
gpurun_out/ or the given sub-directory:
  gpurun_out/prof_kt     --kernel-trace --stats            of `python3 bench.py --steps 20 --warmup 3`
  gpurun_out/prof_fetch  --pmc FETCH_SIZE --kernel-trace   of `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline`
  gpurun_out/prof_write  --pmc WRITE_SIZE --kernel-trace   (same command)
  gpurun_out/prof_req    --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read (128-B requests tallied at 64 B), so
the read side is doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.  Both corrections are cross-checked
here against the raw request counters (RDREQ x 128 B, WRREQ_64B x 64 B).
"""
from __future__ import annotations

import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "profiles")
SRC = os.path.join(ROOT, "gpurun_out")


def one(pattern: str) -> str:
    hits = sorted(glob.glob(os.path.join(SRC, pattern)))
    if not hits:
        raise SystemExit(f"missing {pattern}")
    return hits[-1]


def short(name: str):
    if "fwd_tiled" in name or "bc7_forward" in name:
        return "fwd_tiled"      # the forward kernel of the format benched (bc7: bc7_forward)
    if "inv_tiled" in name or "bc7_inverse" in name:
        return "inv_tiled"
    return None


def counters(dirname: str):
    """per-launch averages over the FULL-SIZE dispatches of each kernel (bench.py's checks also launch the kernels on
    small samples: those rows have a smaller grid and are left out)"""
    rows = []
    with open(one(f"{dirname}/*/*_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if k:
                rows.append((k, r["Counter_Name"], int(r["Grid_Size"]), float(r["Counter_Value"])))
    biggest = collections.defaultdict(int)
    for k, _, g, _ in rows:
        biggest[k] = max(biggest[k], g)
    agg = collections.defaultdict(list)
    for k, c, g, v in rows:
        if g == biggest[k]:
            agg[(k, c)].append(v)
    return {k: sum(v) / len(v) for k, v in agg.items()}


def full_size_durations():
    """average duration (ns) of the full-size dispatches of the forward / inverse kernel, from the kernel trace"""
    rows = []
    with open(one("prof_kt/*/*_kernel_trace.csv")) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if k:
                rows.append((k, int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
    biggest = collections.defaultdict(int)
    for k, g, _ in rows:
        biggest[k] = max(biggest[k], g)
    out = {}
    for k in biggest:
        d = [t for kk, g, t in rows if kk == k and g == biggest[k]]
        out[k] = {"launches": len(d), "average_ns": sum(d) / len(d), "min_ns": min(d), "max_ns": max(d)}
    return out


def main() -> None:
    global SRC
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    if len(sys.argv) > 2:
        SRC = os.path.join(SRC, sys.argv[2])
    stats_rows = []
    with open(one("prof_kt/*/*_kernel_stats.csv")) as f:
        for r in csv.DictReader(f):
            if short(r["Name"]) or "fill_splitmix64" in r["Name"]:
                stats_rows.append(r)
    with open(os.path.join(OUT, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(stats_rows[0].keys()))
        w.writeheader()
        w.writerows(stats_rows)

    bench_line = None
    with open(os.path.join(SRC, "prof_kt.log")) as f:
        for line in f:
            if line.startswith('{"metric"'):
                bench_line = json.loads(line)
    with open(os.path.join(OUT, f"{tag}_bench_under_rocprof.json"), "w") as f:
        json.dump(bench_line, f, indent=1)
        f.write("\n")

    c = {}
    for d in ("prof_fetch", "prof_write", "prof_req"):
        c.update(counters(d))
    nbytes = bench_line["config"]["bytes_per_gpu"]
    rec = {"format": bench_line["config"]["format"], "workload_bytes": nbytes, "source": f"profiles/{tag}_pmc.json"}
    detail = {}
    for k in ("fwd_tiled", "inv_tiled"):
        fetch_kib, write_kib = c[(k, "FETCH_SIZE")], c[(k, "WRITE_SIZE")]
        read_b = 2 * fetch_kib * 1024   # gfx950: FETCH_SIZE counts 128-B requests at 64 B
        write_b = write_kib * 1024
        detail[k] = {
            "FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib,
            "TCC_EA0_RDREQ_sum": c.get((k, "TCC_EA0_RDREQ_sum")), "TCC_EA0_RDREQ_32B_sum": c.get((k, "TCC_EA0_RDREQ_32B_sum")),
            "TCC_EA0_WRREQ_sum": c.get((k, "TCC_EA0_WRREQ_sum")), "TCC_EA0_WRREQ_64B_sum": c.get((k, "TCC_EA0_WRREQ_64B_sum")),
            "hbm_read_bytes": read_b, "hbm_write_bytes": write_b, "hbm_bytes": read_b + write_b,
            "read_bytes_from_RDREQ_x128": (c.get((k, "TCC_EA0_RDREQ_sum")) or 0) * 128,
            "write_bytes_from_WRREQ_x64": (c.get((k, "TCC_EA0_WRREQ_sum")) or 0) * 64,
            "algorithmic_bytes": 2 * nbytes, "traffic_over_algorithmic": (read_b + write_b) / (2 * nbytes),
        }
    dur = full_size_durations()
    for k in ("fwd_tiled", "inv_tiled"):
        detail[k]["rocprofv3_full_size_launches"] = dur.get(k)
        if dur.get(k):
            detail[k]["fraction_of_8TBps_on_algorithmic_bytes"] = round(2 * nbytes / (dur[k]["average_ns"] * 1e-9) / 8e12, 4)
    rec["fwd_hbm_bytes_per_launch"] = int(detail["fwd_tiled"]["hbm_bytes"])
    rec["inv_hbm_bytes_per_launch"] = int(detail["inv_tiled"]["hbm_bytes"])
    with open(os.path.join(OUT, f"{tag}_pmc.json"), "w") as f:
        json.dump(detail, f, indent=1)
        f.write("\n")
    if rec["format"] == "bc1":   # what bench.py quotes (labelled) as roofline.traffic for the headline workload
        with open(os.path.join(OUT, "pmc_traffic.json"), "w") as f:
            json.dump(rec, f, indent=1)
            f.write("\n")
    print(json.dumps(rec))
    for r in stats_rows:
        print(r["Name"][:60], r["Calls"], r["AverageNs"])


if __name__ == "__main__":
    main()
