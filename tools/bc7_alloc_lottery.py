"""BC7 forward / inverse (4 GiB, uniform mix, steady state) over fresh allocations inside ONE process: three 4 GiB buffers are
allocated behind a spacer of varying size, measured, freed.  Does the rate depend on where the allocator puts them?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dxt_lossless_transform_amd as pkg
from dxt_lossless_transform_amd import bc7
import bench
dev = torch.device("cuda:0")
n = 4 << 30
def measure(x, y, z):
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
for spacer_mib in (0, 2, 6, 34, 130, 514, 1026, 2050, 0, 4098, 8194, 2):
    sp = torch.empty(max(1, spacer_mib) << 20, dtype=torch.uint8, device=dev) if spacer_mib else None
    x = torch.empty(n, dtype=torch.uint8, device=dev); y = torch.empty_like(x); z = torch.empty_like(x)
    pkg.fill_splitmix64(x, 0x0BC70004); bench.bc7_force_modes_device(torch, x, "uniform")
    fw, iv = measure(x, y, z)
    print(f"spacer {spacer_mib:5d} MiB  x {x.data_ptr():#x} y-x {(y.data_ptr() - x.data_ptr()) / 2**20:9.1f} MiB z-y {(z.data_ptr() - y.data_ptr()) / 2**20:9.1f} MiB  fwd {fw:.4f} inv {iv:.4f}", flush=True)
    del x, y, z, sp
    torch.cuda.empty_cache()
