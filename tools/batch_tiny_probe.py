"""dxtlt_transform_batch_device on tens of thousands of TINY device buffers (the case nothing measured): kernel time only,
fraction of 8 TB/s on 2 * len and microseconds per call, forward / inverse.  Same box: DXTLT_LIB_PATH=ab/libdxtlt_r04.so against the tree.
  one_wg      65 536 BC1 buffers of 1..511 blocks: every buffer ONE workgroup (round 4: general lookup, a byte index that saturates after
              255 buffers per 4096 workgroups and a walk of up to ~3800 entries; now the entry is the workgroup number)
  few_tiles   65 536 BC1 buffers of 1..1500 blocks, 1-3 tiles each (round 4: the same walk; now 16-bit index + bisection)
  mip43k      8000 x 5463 blocks (a 256 x 256 texture with its mip chain, 43 KiB), equal buffers: the division lookup in both"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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

rng = np.random.default_rng(0x71)
cases = {"one_wg": [int(x) for x in rng.integers(1, 512, 65536)], "few_tiles": [int(x) for x in rng.integers(1, 1500, 65536)],
         "mip43k": [5463] * 8000}
st = pkg.Bc1TransformSettings()
for name, counts in cases.items():
    offs, at = [], 0
    for n in counts:
        offs.append(at)
        at += (n * 8 + 255) // 256 * 256
    x = torch.empty(at, dtype=torch.uint8, device=dev); pkg.fill_splitmix64(x, 9)
    y = torch.zeros_like(x); z = torch.zeros_like(x)
    total = sum(counts) * 8
    fw = batch.prepare_batch([("bc1", False, x[o:o + n * 8], y[o:o + n * 8], st) for n, o in zip(counts, offs)])
    iv = batch.prepare_batch([("bc1", True, y[o:o + n * 8], z[o:o + n * 8], st) for n, o in zip(counts, offs)])
    tf = timed(lambda: batch.run_prepared_batch(fw)); ti = timed(lambda: batch.run_prepared_batch(iv))
    ok = all(bool(torch.equal(x[o:o + n * 8], z[o:o + n * 8])) for n, o in list(zip(counts, offs))[::997])
    print(f"{name:10s} {len(counts):6d} buffers {total / 2**20:8.1f} MiB  fwd {2 * total / tf / 8e12:.3f} ({tf * 1e6:7.1f} us)  inv {2 * total / ti / 8e12:.3f} ({ti * 1e6:7.1f} us)  round trip {'exact' if ok else 'WRONG'}", flush=True)
    del x, y, z
