"""Single-buffer aligned tiles: does a power-of-two block count (streams powers of two apart) cost anything?  fwd / inv fraction of 8 TB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dxt_lossless_transform_amd as pkg
dev = torch.device("cuda:0")
def timed(fn, steps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps * 1e-3
for fmt, B in (("bc1", 8), ("bc3", 16)):
    base = (4 << 30) // B
    buf = torch.empty((base + (1 << 22)) * B, dtype=torch.uint8, device=dev); pkg.fill_splitmix64(buf, 3)
    y = torch.empty_like(buf); z = torch.empty_like(buf)
    f = getattr(pkg, f"transform_{fmt}_with_settings"); g = getattr(pkg, f"untransform_{fmt}_with_settings")
    for extra in (0, 2048, 2048 * 3, 2048 * 17, 2048 * 129, 2048 * 1025, 0):
        n = (base + extra) * B
        x, yy, zz = buf[:n], y[:n], z[:n]
        tf = timed(lambda: f(x, yy)); ti = timed(lambda: g(yy, zz))
        print(fmt, "blocks 2^k +", extra, round(2 * n / tf / 8e12, 4), round(2 * n / ti / 8e12, 4), flush=True)
