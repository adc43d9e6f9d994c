"""Does the spacing of a small buffer's streams matter?  BC3 default settings through dxtlt_transform_batch_device, 1 GiB in all,
buffers of 2^16 blocks (1 MiB: every stream base a multiple of 64 KiB) against 2^16 + 2^11 + 128 * j blocks (bases still on
128-byte lines -- aligned tiles -- but no longer powers of two apart).  Fraction of 8 TB/s on 2 * len, forward / inverse."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dxt_lossless_transform_amd as pkg
from dxt_lossless_transform_amd import batch
dev = torch.device("cuda:0")
total = 1 << 30
big = torch.empty(total + (64 << 20), dtype=torch.uint8, device=dev); pkg.fill_splitmix64(big[:total], 5)
outb = torch.empty_like(big)
def timed(fn, reps=10):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3
for fmt, B in (("bc3", 16), ("bc1", 8)):
    st = pkg.Bc3TransformSettings() if fmt == "bc3" else pkg.Bc1TransformSettings()
    for blocks in (1 << 16, (1 << 16) + (1 << 11), (1 << 16) + (1 << 11) + 128, (1 << 16) + 3 * 128, (1 << 16) + 17 * 128, 1 << 18, (1 << 18) + 5 * 128):
        n = blocks * B
        count = total // n
        res = []
        for inverse in (False, True):
            items = [(fmt, inverse, big[i * n:(i + 1) * n], outb[i * n:(i + 1) * n], st) for i in range(count)]
            prep = batch.prepare_batch(items)
            t = timed(lambda: batch.run_prepared_batch(prep))
            res.append(round(2 * n * count / t / 8e12, 3))
        print(fmt, "blocks per buffer", blocks, "=", n >> 10, "KiB x", count, res, flush=True)
