#!/usr/bin/env python3
"""BASELINE.json configs[3]: BC7 granule-sorted field split, version 2 (this build's own format, docs/BC7_FORMAT.md) on a 4 GiB synthetic
mode-mixed buffer, one MI355X.  Prints fwd / inv time and the fraction of the HBM peak on ALGORITHMIC bytes (2*len)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dxt_lossless_transform_amd as pkg  # noqa: E402
from dxt_lossless_transform_amd import bc7  # noqa: E402

gib = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
dist = sys.argv[2] if len(sys.argv) > 2 else "uniform"
steps = int(os.environ.get("STEPS", "20"))
dev = torch.device("cuda:0")
n = int(gib * (1 << 30)) // 16
x = torch.empty(n * 16, dtype=torch.uint8, device=dev)
pkg.fill_splitmix64(x, 0x0BC70004)
b = x.view(-1, 16)
r = b[:, 15].to(torch.int32)
if dist == "uniform":
    m = r & 7
elif dist == "mode6":
    m = torch.full_like(r, 6)
elif dist == "reserved":   # byte 0 == 0 everywhere: class 8, records == blocks -- the sort and the streams without any bit-field work
    m = torch.full_like(r, 8)
elif dist == "mode0":
    m = torch.full_like(r, 0)
elif dist == "runs":   # texture-like: long runs of one mode (64 blocks), modes skewed
    rr = r.view(-1, 64)[:, :1].expand(-1, 64).reshape(-1)
    m = torch.where(rr < 140, 6, torch.where(rr < 200, 1, torch.where(rr < 230, 3, rr & 7))).to(torch.int32)
else:  # texture-like skew: mode 6 > 1 > 3 > others
    m = torch.where(r < 140, 6, torch.where(r < 200, 1, torch.where(r < 230, 3, r & 7))).to(torch.int32)
low = ((2 << m) - 1).to(torch.uint8)
b[:, 0] = (b[:, 0] & ~low) | ((1 << m) & 0xFF).to(torch.uint8)
del r, low
y, z = torch.empty_like(x), torch.empty_like(x)
# warm up for WARM_MS of wall time, not for a number of launches: after an idle phase the chip needs ~40 ms of load to reach its
# steady clocks, and these kernels -- close to the vector-issue bound -- run at 0.5-0.7 of peak until then (tools/clock_ramp_probe.py,
# profiles/r03_clock_ramp.txt).  Rounds 1-2 and the first half of round 3 warmed up with two launches: their BC7 figures are ramp figures.
import time
warm_ms = float(os.environ.get("WARM_MS", "150"))
t0 = time.perf_counter()
while (time.perf_counter() - t0) * 1e3 < warm_ms:
    for _ in range(8):          # back to back: a wait after every launch would let the clocks sag again
        bc7.transform_bc7(x, y)
        bc7.untransform_bc7(y, z)
    torch.cuda.synchronize()
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
torch.cuda.synchronize()
for k in range(steps):
    ev[k][0].record()
    bc7.transform_bc7(x, y)
    ev[k][1].record()
    bc7.untransform_bc7(y, z)
    ev[k][2].record()
torch.cuda.synchronize()
fwd = sum(e[0].elapsed_time(e[1]) for e in ev) / steps
inv = sum(e[1].elapsed_time(e[2]) for e in ev) / steps
nbytes = x.numel()
print(json.dumps({
    "workload": f"BC7 granule-sorted field split v2, {gib:g} GiB, modes {dist}", "roundtrip_exact": bool(torch.equal(z, x)),
    "fwd_ms": round(fwd, 3), "inv_ms": round(inv, 3),
    "fwd_GiBps": round(nbytes / fwd / 1e-3 / 2**30, 1), "inv_GiBps": round(nbytes / inv / 1e-3 / 2**30, 1),
    "fwd_frac_of_8TBps_on_2len": round(2 * nbytes / (fwd * 1e-3) / 8e12, 4),
    "inv_frac_of_8TBps_on_2len": round(2 * nbytes / (inv * 1e-3) / 8e12, 4),
    "mode_counts": torch.bincount(m, minlength=9).tolist(),
}))
