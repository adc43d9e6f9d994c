"""dxtlt_transform_batch_device, kernel time only (HIP events around the C call, prepared item array): BC1 / BC3 default
settings (PROBE_FMTS, PROBE_COUNTS choose), forward and inverse, 1 x 1 GiB, 64 x 16 MiB, 1024 x 1 MiB, 4096 x 256 KiB; fraction of the HBM peak on 2 * len,
beside the single-buffer call on the 1 GiB buffer."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dxt_lossless_transform_amd as pkg
from dxt_lossless_transform_amd import batch
dev = torch.device("cuda:0")
out = {}
for fmt in os.environ.get("PROBE_FMTS", "bc1,bc3").split(","):
    st = pkg.Bc1TransformSettings() if fmt == "bc1" else pkg.Bc3TransformSettings()
    total = 1 << 30
    big = torch.empty(total, dtype=torch.uint8, device=dev); pkg.fill_splitmix64(big, 5)
    outb = torch.empty_like(big)
    def timed(fn, reps=10):
        for _ in range(5):      # the table ring has four slots; each grows on its first use
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e-3
    f = getattr(pkg, f"transform_{fmt}_with_settings"); g = getattr(pkg, f"untransform_{fmt}_with_settings")
    res = {"single_1GiB": (round(2 * total / timed(lambda: f(big, outb, st)) / 8e12, 3), round(2 * total / timed(lambda: g(big, outb, st)) / 8e12, 3))}
    for count in [int(c) for c in os.environ.get("PROBE_COUNTS", "1,64,1024,4096").split(",")]:
        n = total // count
        for inverse in (False, True):
            items = [(fmt, inverse, big[i * n:(i + 1) * n], outb[i * n:(i + 1) * n], st) for i in range(count)]
            prep = batch.prepare_batch(items)
            t = timed(lambda: batch.run_prepared_batch(prep))
            res.setdefault(f"{count} x {n >> 10} KiB", []).append(round(2 * total / t / 8e12, 3))
        if count >= 64:   # one block short: odd block counts, every stream base off its line (a DDS payload with a mip chain)
            m = n - (8 if fmt == "bc1" else 16)
            for inverse in (False, True):
                items = [(fmt, inverse, big[i * n:i * n + m], outb[i * n:i * n + m], st) for i in range(count)]
                prep = batch.prepare_batch(items)
                t = timed(lambda: batch.run_prepared_batch(prep))
                res.setdefault(f"{count} x ({n >> 10} KiB - 1 block)", []).append(round(2 * m * count / t / 8e12, 3))
    out[fmt] = res
print(json.dumps(out))
