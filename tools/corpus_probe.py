"""The corpus-shaped legs of bench.py alone (2130 mip-chained BC1 textures, 8693 MiB; the BC3 twin), one batch call per
direction: fraction of the HBM peak forward / inverse.  PROBE_FMTS=bc1,bc3  PROBE_STEPS=10  PROBE_SCALE=1.0  PROBE_ALIGN=256 (byte boundary every texture starts on)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import dxt_lossless_transform_amd as pkg

pkg.load()
dev = torch.device("cuda:0")
out = {}
for fmt in os.environ.get("PROBE_FMTS", "bc1,bc3").split(","):
    leg = bench.run_corpus_leg(pkg, torch, dev, fmt, int(os.environ.get("PROBE_STEPS", "10")), 2,
                               float(os.environ.get("PROBE_SCALE", "1.0")), cpu=False, align=int(os.environ.get("PROBE_ALIGN", "256")))
    out[fmt] = {"fwd": leg["roofline"]["frac"], "inv": leg["roofline"]["inverse_kernel"]["frac"], "fwd_ms": leg["fwd_ms"],
                "inv_ms": leg["inv_ms"], "exact": leg["bit_exact_roundtrip"] and leg["oracle_textures_exact"], "textures": leg["textures"]}
print(json.dumps(out))
