"""Which alignment costs the small-buffer batch: the streams INSIDE a buffer (size 2^k) or the buffers among each other (stride 2^k)?
BC3 / BC1 default settings, buffers of 2^16 blocks placed at stride size + pad."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dxt_lossless_transform_amd as pkg
from dxt_lossless_transform_amd import batch
dev = torch.device("cuda:0")
total = 1 << 30
big = torch.empty(total + (256 << 20), dtype=torch.uint8, device=dev); pkg.fill_splitmix64(big[:total], 5)
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
    n = (1 << 16) * B
    count = total // n
    for pad_in, pad_out in ((0, 0), (4352, 4352), (0, 4352), (4352, 0), (128 * 1024 + 256, 128 * 1024 + 256), (2304, 2304)):
        res = []
        for inverse in (False, True):
            items = [(fmt, inverse, big[i * (n + pad_in):i * (n + pad_in) + n], outb[i * (n + pad_out):i * (n + pad_out) + n], st) for i in range(count)]
            prep = batch.prepare_batch(items)
            t = timed(lambda: batch.run_prepared_batch(prep))
            res.append(round(2 * n * count / t / 8e12, 3))
        print(fmt, "2^16-block buffers, stride pad in/out", pad_in, pad_out, res, flush=True)
