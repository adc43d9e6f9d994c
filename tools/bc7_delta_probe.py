"""One hipMalloc (physically whatever the driver gives, but FIXED for the process), the three BC7 buffers carved out of it: x at 0, y at
4 GiB + 64 MiB + delta, z behind y at a fixed distance.  Does the forward / inverse level move with delta -- i.e. with the bits of y's
address that enter the channel fold (tools/channel_model.py) -- while nothing physical changes?  4 GiB uniform mix, steady state."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dxt_lossless_transform_amd as pkg
from dxt_lossless_transform_amd import bc7
import bench
dev = torch.device("cuda:0")
n = 4 << 30
big = torch.empty(3 * n + (2 << 30), dtype=torch.uint8, device=dev)
print("base address:", hex(big.data_ptr()), flush=True)
x = big[0:n]
pkg.fill_splitmix64(x, 0x0BC70004); bench.bc7_force_modes_device(torch, x, "uniform")
def measure(y, z):
    f = lambda: bc7.transform_bc7(x, y); g = lambda: bc7.untransform_bc7(y, z)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.12:
        for _ in range(8): f(); g()
        torch.cuda.synchronize()
    steps = 16
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * steps + 1)]
    for i in range(steps):
        ev[2 * i].record(); f(); ev[2 * i + 1].record(); g()
    ev[2 * steps].record(); torch.cuda.synchronize()
    fw = sum(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(steps)) / steps
    iv = sum(ev[2 * i + 1].elapsed_time(ev[2 * i + 2]) for i in range(steps)) / steps
    return 2 * n / (fw * 1e-3) / 8e12, 2 * n / (iv * 1e-3) / 8e12
deltas = [0] + [256 << k for k in range(0, 18)] + [3 << 20, 5 << 20, 7 << 20, 12 << 20, 20 << 20, 36 << 20, 0]
for d in deltas:
    y = big[n + (64 << 20) + d: 2 * n + (64 << 20) + d]
    z = big[2 * n + (512 << 20): 3 * n + (512 << 20)]
    fw, iv = measure(y, z)
    print(f"delta {d:>10d} ({d / 2**20:9.4f} MiB)  fwd {fw:.4f}  inv {iv:.4f}", flush=True)
assert torch.equal(x, z)
