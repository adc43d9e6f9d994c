"""Headline + legs fractions of one library (DXTLT_LIB_PATH), compact: for same-box A/B of library builds.
Runs bench.py --no-cpu-baseline --host-array-gib 0 as a child and prints fwd / inv fractions per leg."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--host-array-gib", "0", *sys.argv[1:]],
                   capture_output=True, text=True)
if r.returncode != 0:
    print(r.stderr[-2000:]); sys.exit(1)
d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][0])
row = {"headline": (d["roofline"]["frac"], d["roofline"]["inverse_kernel"]["frac"])}
for k, v in d.get("legs", {}).items():
    row[k] = (v["roofline"]["frac"], v["roofline"]["inverse_kernel"]["frac"])
print(os.environ.get("DXTLT_LIB_PATH", "default"), json.dumps(row))
