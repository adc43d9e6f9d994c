"""Forward halo tiles of 256 lanes (the product's) against 512 lanes (dxtlt_set_tuning(512, 0)): half the halo share per tile, a fatter
workgroup.  One buffer per case, 4 GiB (+ extra blocks: the stream shifts, hence the halo size), fraction of 8 TB/s on 2 * len.
`23`: the residue of every mip-chained texture's block count modulo 64 -- the corpus legs' halo (BC3 28 blocks, BC1 23)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dxt_lossless_transform_amd as pkg
dev = torch.device("cuda:0")
def timed(fn, steps=30):
    for _ in range(3): fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(steps): fn()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / steps
warm = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
for _ in range(50): warm.add_(1)
torch.cuda.synchronize(); del warm
for fmt, B in (("bc3", 16), ("bc1", 8)):
    base = (4 << 30) // B
    for extra in (1, 23, 63):
        n = base + extra
        x = torch.empty(n * B, dtype=torch.uint8, device=dev); pkg.fill_splitmix64(x, 5)
        y = torch.empty_like(x); z = torch.empty_like(x)
        f = getattr(pkg, f"transform_{fmt}_with_settings"); g = getattr(pkg, f"untransform_{fmt}_with_settings")
        row = []
        for threads in (256, 512, 256, 512):
            pkg.set_tuning(threads if threads == 512 else 0, 0)
            t = timed(lambda: f(x, y))
            pkg.set_tuning(0, 0)
            g(y, z); torch.cuda.synchronize()
            assert torch.equal(x, z), (fmt, extra, threads)
            row.append(2 * n * B / (t * 1e-3) / 8e12)
        print(f"{fmt} 2^k + {extra:2d} blocks: 256 lanes {row[0]:.4f} {row[2]:.4f}   512 lanes {row[1]:.4f} {row[3]:.4f}   (512 - 256: {((row[1] + row[3]) - (row[0] + row[2])) / 2:+.4f})", flush=True)
        del x, y, z
