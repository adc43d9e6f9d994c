"""Does the relative placement of the block buffer and the transformed buffer matter for the BC7 kernels?  One allocation, the three
4 GiB buffers carved out of it `pad` bytes further apart than their size; steady state (150 ms warm-up), uniform mix, fwd / inv fraction."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dxt_lossless_transform_amd as pkg
from dxt_lossless_transform_amd import bc7
import bench
dev = torch.device("cuda:0")
n = 4 << 30
big = torch.empty(3 * n + (1 << 30), dtype=torch.uint8, device=dev)
print("base address mod 2^32:", hex(big.data_ptr() & 0xFFFFFFFF), flush=True)
for pad in (0, 4096, 65536 + 4096, (1 << 20) + 4096, (16 << 20) + 8192, (128 << 20) + 12288, 0):
    x = big[0:n]; y = big[n + pad: 2 * n + pad]; z = big[2 * n + 2 * pad: 3 * n + 2 * pad]
    pkg.fill_splitmix64(x, 0x0BC70004); bench.bc7_force_modes_device(torch, x, "uniform")
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
    assert torch.equal(x, z)
    print(f"pad {pad:>10d}  fwd {2 * n / (fw * 1e-3) / 8e12:.4f}  inv {2 * n / (iv * 1e-3) / 8e12:.4f}", flush=True)
