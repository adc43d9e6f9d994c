"""Relative placement of the block buffer, the transformed buffer and the restored buffer: BC1 / BC3 default settings, 8 GiB each, carved out of one
allocation `pad` bytes further apart than their size.  fwd / inv fraction of 8 TB/s, 20 steps after 8 warm-up pairs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dxt_lossless_transform_amd as pkg
dev = torch.device("cuda:0")
n = 8 << 30
big = torch.empty(3 * n + (1 << 30), dtype=torch.uint8, device=dev)
pkg.fill_splitmix64(big[:n], 3)
for fmt in ("bc1", "bc3"):
    f = getattr(pkg, f"transform_{fmt}_with_settings"); g = getattr(pkg, f"untransform_{fmt}_with_settings")
    for pad in (0, 128, 512, 1024, 2048, 4096, 8192, 4096 + 128, 65536 + 2048, (1 << 20) + 4096, (2 << 20) + 1024, 0):
        x = big[0:n]; y = big[n + pad: 2 * n + pad]; z = big[2 * n + 2 * pad: 3 * n + 2 * pad]
        for _ in range(8): f(x, y); g(y, z)
        steps = 20
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * steps + 1)]
        for i in range(steps):
            ev[2 * i].record(); f(x, y); ev[2 * i + 1].record(); g(y, z)
        ev[2 * steps].record(); torch.cuda.synchronize()
        fw = sum(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(steps)) / steps
        iv = sum(ev[2 * i + 1].elapsed_time(ev[2 * i + 2]) for i in range(steps)) / steps
        assert torch.equal(x, z)
        print(f"{fmt} pad {pad:>9d}  fwd {2 * n / (fw * 1e-3) / 8e12:.4f}  inv {2 * n / (iv * 1e-3) / 8e12:.4f}", flush=True)
