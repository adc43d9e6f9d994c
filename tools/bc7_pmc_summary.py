#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv rows per kernel name prefix (BC7 tuning aid)."""
import csv
import glob
import sys
from collections import defaultdict

for d in sys.argv[1:]:
    acc = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(set)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("dxtlt::bc7::", "")
            if "bc7" not in k:
                continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[k].add(r["Dispatch_Id"])
    print("==", d)
    for k in sorted(acc):
        n = len(calls[k])
        print(k, "dispatches", n, {c: round(v / n) for c, v in sorted(acc[k].items())})
