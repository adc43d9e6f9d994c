"""Workload for PMC passes: ONE 1 GiB BC3 buffer through the single call and through the batch call (one entry), forward + inverse x3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dxt_lossless_transform_amd as pkg
from dxt_lossless_transform_amd import batch
fmt = sys.argv[1] if len(sys.argv) > 1 else "bc3"
st = pkg.Bc3TransformSettings() if fmt == "bc3" else pkg.Bc1TransformSettings()
x = torch.empty(1 << 30, dtype=torch.uint8, device="cuda:0"); pkg.fill_splitmix64(x, 3)
y = torch.empty_like(x); z = torch.empty_like(x)
f = getattr(pkg, f"transform_{fmt}_with_settings"); g = getattr(pkg, f"untransform_{fmt}_with_settings")
pf = batch.prepare_batch([(fmt, False, x, y, st)]); pi = batch.prepare_batch([(fmt, True, y, z, st)])
for _ in range(3):
    f(x, y, st); g(y, z, st)
    batch.run_prepared_batch(pf); batch.run_prepared_batch(pi)
torch.cuda.synchronize()
assert torch.equal(x, z)
