"""dxtlt_transform_batch_device on arrays of equal buffers, kernel time only: PROBE_CASES = "fmt:count:blocks:stride_bytes;..."
(stride 0 = blocks * block size rounded up to 256).  Prints the fraction of the HBM peak on 2 * len, forward / inverse."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dxt_lossless_transform_amd as pkg
from dxt_lossless_transform_amd import batch
dev = torch.device("cuda:0")

def timed(fn, reps=10):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3

for case in os.environ["PROBE_CASES"].split(";"):
    fmt, count, blocks, stride = case.split(":")
    count, blocks, stride = int(count), int(blocks), int(stride)
    B = 8 if fmt == "bc1" else 16
    st = pkg.Bc1TransformSettings() if fmt == "bc1" else pkg.Bc3TransformSettings()
    n = blocks * B
    stride = stride or (n + 255) // 256 * 256
    big = torch.empty(count * stride, dtype=torch.uint8, device=dev); pkg.fill_splitmix64(big, 5)
    outb = torch.empty_like(big)
    res = []
    for inverse in (False, True):
        items = [(fmt, inverse, big[i * stride:i * stride + n], outb[i * stride:i * stride + n], st) for i in range(count)]
        prep = batch.prepare_batch(items)
        t = timed(lambda: batch.run_prepared_batch(prep))
        res.append(round(2 * n * count / t / 8e12, 3))
    print(case, res, flush=True)
    del big, outb
