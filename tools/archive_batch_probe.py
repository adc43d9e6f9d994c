"""configs[4] slice (32 alternating 256 MiB BC1 / BC3 textures): one call per texture against ONE batch call per direction (the BC1 textures and
the BC3 textures are each a regular array: the tiled kernel with blockIdx.y = texture).  Steady state; fraction of 8 TB/s on 2 * bytes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dxt_lossless_transform_amd as pkg
from dxt_lossless_transform_amd import batch
dev = torch.device("cuda:0")
tex, k = 256 << 20, 32
x = torch.empty(k * tex, dtype=torch.uint8, device=dev); pkg.fill_splitmix64(x, 5)
y = torch.empty_like(x); z = torch.empty_like(x)
xs, ys, zs = (list(t.view(k, tex).unbind(0)) for t in (x, y, z))
fm = ["bc1" if i % 2 == 0 else "bc3" for i in range(k)]
st = {"bc1": pkg.Bc1TransformSettings(), "bc3": pkg.Bc3TransformSettings()}
F = {f: getattr(pkg, f"transform_{f}_with_settings") for f in st}
G = {f: getattr(pkg, f"untransform_{f}_with_settings") for f in st}
pf = batch.prepare_batch([(fm[i], False, xs[i], ys[i], st[fm[i]]) for i in range(k)])
pi = batch.prepare_batch([(fm[i], True, ys[i], zs[i], st[fm[i]]) for i in range(k)])
def per_tex_f():
    for i in range(k): F[fm[i]](xs[i], ys[i], st[fm[i]])
def per_tex_g():
    for i in range(k): G[fm[i]](ys[i], zs[i], st[fm[i]])
for name, f, g in (("one call per texture", per_tex_f, per_tex_g), ("one batch call", lambda: batch.run_prepared_batch(pf), lambda: batch.run_prepared_batch(pi))) * 2:
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        f(); g(); torch.cuda.synchronize()
    steps = 10
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * steps + 1)]
    for i in range(steps):
        ev[2 * i].record(); f(); ev[2 * i + 1].record(); g()
    ev[2 * steps].record(); torch.cuda.synchronize()
    fw = sum(ev[2 * i].elapsed_time(ev[2 * i + 1]) for i in range(steps)) / steps
    iv = sum(ev[2 * i + 1].elapsed_time(ev[2 * i + 2]) for i in range(steps)) / steps
    assert torch.equal(x, z)
    print(f"{name:22s} fwd {2 * k * tex / (fw * 1e-3) / 8e12:.4f}  inv {2 * k * tex / (iv * 1e-3) / 8e12:.4f}", flush=True)
