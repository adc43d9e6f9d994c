"""BC7 forward / inverse (uniform mix, steady state) over block counts around 4 GiB, three fresh allocations each: is the
placement lottery of profiles/r04_bc7_placement.txt a property of the power-of-two stream offsets (8n, 10n, 11n ... with
n = 2^28) meeting physically contiguous backing, or of any size?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dxt_lossless_transform_amd as pkg
from dxt_lossless_transform_amd import bc7
import bench
dev = torch.device("cuda:0")
def measure(x, y, z, n):
    f = lambda: bc7.transform_bc7(x, y); g = lambda: bc7.untransform_bc7(y, z)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        for _ in range(8): f(); g()
        torch.cuda.synchronize()
    steps = 20
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * steps + 1)]
    for i in range(steps):
        ev[2 * i].record(); f(); ev[2 * i + 1].record(); g()
    ev[2 * steps].record(); torch.cuda.synchronize()
    fw = sum(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(steps)) / steps
    iv = sum(ev[2 * i + 1].elapsed_time(ev[2 * i + 2]) for i in range(steps)) / steps
    return 2 * n / (fw * 1e-3) / 8e12, 2 * n / (iv * 1e-3) / 8e12
G = 1024 * 16   # bytes per granule
for granules_off in (0, -37, 53, -4099, 8191, 0, -(1 << 14), 1 << 14):
    n = (4 << 30) + granules_off * G
    row = []
    for rep in range(3):
        sp = torch.empty((2 + 4 * rep) << 20, dtype=torch.uint8, device=dev)
        x = torch.empty(n, dtype=torch.uint8, device=dev); y = torch.empty_like(x); z = torch.empty_like(x)
        pkg.fill_splitmix64(x, 0x0BC70004); bench.bc7_force_modes_device(torch, x, "uniform")
        fw, iv = measure(x, y, z, n)
        row.append(f"{fw:.4f}/{iv:.4f}")
        del x, y, z, sp
        torch.cuda.empty_cache()
    print(f"4 GiB {granules_off:+7d} granules: " + "  ".join(row), flush=True)
