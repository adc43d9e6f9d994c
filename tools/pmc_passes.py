#!/usr/bin/env python3
"""Runs one rocprofv3 --pmc pass per counter set over a workload script and prints per-kernel averages.

    python3 tools/pmc_passes.py OUT_DIR SETS_FILE -- python3 tools/odd_pmc.py 1

SETS_FILE: one counter set per line -- at most 8 SQ and 4 TCC slots per set (FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2:
MI355X_MICROARCH.md "rocprofv3 PMC slots"; an over-subscribed set does not fail, it hangs until the pass timeout) (space separated counter names; '#' comments).  Every pass is its own run with
--kernel-trace only (never combined with a trace domain gpurun refuses).  A pass whose counters this rocprofv3 does not
know fails on its own and is reported; a pass that times out stops the whole script (no further GPU work after a hang).
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys


def main():
    out_dir, sets_file = sys.argv[1], sys.argv[2]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    sets = [l.split("#")[0].split() for l in open(sets_file)]
    sets = [s for s in sets if s]
    os.makedirs(out_dir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    summary = collections.defaultdict(dict)
    for i, counters in enumerate(sets):
        d = os.path.join(out_dir, f"pass{i}")
        full = ["rocprofv3", "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", d, "--"] + cmd
        try:
            r = subprocess.run(full, env=env, capture_output=True, text=True, timeout=int(os.environ.get("PMC_PASS_TIMEOUT", "240")))
        except subprocess.TimeoutExpired:
            print(f"pass {i} {counters}: TIMEOUT, stopping", flush=True)
            break
        if r.returncode != 0:
            print(f"pass {i} {counters}: rc {r.returncode}\n{r.stderr[-600:]}", flush=True)
            continue
        agg = collections.defaultdict(list)
        for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                agg[(row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in sorted(agg.items()):
            short = k.split("(")[0]
            summary[short][c] = sum(v) / len(v)
            summary[short].setdefault("_dispatches", len(v))
        print(f"pass {i} {counters}: ok", flush=True)
    with open(os.path.join(out_dir, "summary.json"), "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    for k, d in sorted(summary.items()):
        print(k)
        for c, v in sorted(d.items()):
            print(f"    {c:32s} {v:,.1f}")


if __name__ == "__main__":
    main()
