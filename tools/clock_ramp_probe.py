"""Per-launch kernel times after an idle second: how long does the chip take to reach its steady rate?  (BC7 4 GiB and BC1 8 GiB,
forward / inverse alternating, 24 launches each, two rounds with a one-second sleep in between.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dxt_lossless_transform_amd as pkg
from dxt_lossless_transform_amd import bc7
import bench
dev = torch.device("cuda:0")
x = torch.empty(8 << 30, dtype=torch.uint8, device=dev); y = torch.empty_like(x); z = torch.empty_like(x)
for name in ("bc7", "bc1"):
    if name == "bc7":
        a, b, c = x[:4 << 30], y[:4 << 30], z[:4 << 30]
        pkg.fill_splitmix64(a, 0x0BC70004); bench.bc7_force_modes_device(torch, a, "uniform")
        f = lambda: bc7.transform_bc7(a, b); g = lambda: bc7.untransform_bc7(b, c)
        nbytes = 4 << 30
    else:
        pkg.fill_splitmix64(x, 0x0BC10002)
        f = lambda: pkg.transform_bc1_with_settings(x, y); g = lambda: pkg.untransform_bc1_with_settings(y, z)
        nbytes = 8 << 30
    torch.cuda.synchronize()
    for rnd in range(2):
        time.sleep(1.0)
        n = 24
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * n + 1)]
        for i in range(n):
            ev[2 * i].record(); f(); ev[2 * i + 1].record(); g()
        ev[2 * n].record(); torch.cuda.synchronize()
        fr = lambda ms: 2 * nbytes / (ms * 1e-3) / 8e12
        print(name, "round", rnd, "fwd frac:", " ".join(f"{fr(ev[2*i].elapsed_time(ev[2*i+1])):.3f}" for i in range(n)))
        print(name, "round", rnd, "inv frac:", " ".join(f"{fr(ev[2*i+1].elapsed_time(ev[2*i+2])):.3f}" for i in range(n)), flush=True)
