"""The host path against the link's own ceiling, one run: bench.sharded_host_array (pageable and pinned host arrays, with and
without a hipHostRegister of the caller's arrays around the call) beside bench.pcie_pair_ceiling.  PROBE_GIB=4"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import dxt_lossless_transform_amd as pkg

pkg.load()
dev = torch.device("cuda:0")
gib = float(os.environ.get("PROBE_GIB", "4"))
out = {"ceiling": bench.pcie_pair_ceiling(torch, dev)}
for pinned in ("0", "1"):
    os.environ["DXTLT_BENCH_PINNED_HOST"] = pinned
    r = bench.sharded_host_array(pkg, torch, "bc1", pkg.Bc1TransformSettings(), int(gib * 2**30), 0x0BC10002, dev, 1)
    out["pinned" if pinned == "1" else "pageable"] = {k: r[k] for k in ("fwd_GiBps", "inv_GiBps", "host_memory", "setup_s")}
    out["pinned" if pinned == "1" else "pageable"]["frac"] = r["per_device"][0].get("roofline", {}).get("frac")
print(json.dumps(out))
