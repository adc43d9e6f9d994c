import csv, glob, collections, sys
for d in sys.argv[1:]:
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{d}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "tiled" in k:
                agg[(k.split("<")[0].split("::")[-1], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print(d, k, f"{sum(v)/len(v):.1f}", len(v))
