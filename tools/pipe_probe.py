"""EXPERIMENT: the software-pipelined persistent forward kernel (DXTLT_PIPE_GRID=<workgroups>, read once per process) against
the one-tile-per-workgroup kernel on an aligned buffer.  PROBE_FMT=bc1|bc3 PROBE_GIB=4.  Prints forward fraction of the HBM peak."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dxt_lossless_transform_amd as pkg
dev = torch.device("cuda:0")
fmt = os.environ.get("PROBE_FMT", "bc1")
n = int(float(os.environ.get("PROBE_GIB", "4")) * 2**30)
x = torch.empty(n, dtype=torch.uint8, device=dev); pkg.fill_splitmix64(x, 3)
y = torch.empty_like(x); z = torch.empty_like(x)
f = getattr(pkg, f"transform_{fmt}_with_settings"); g = getattr(pkg, f"untransform_{fmt}_with_settings")
def timed(fn, steps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / steps * 1e-3
best = min(timed(lambda: f(x, y)) for _ in range(3))
g(y, z)
ok = torch.equal(x, z)
print(f"{fmt} grid={os.environ.get('DXTLT_PIPE_GRID', 'off'):>6s} fwd {2 * n / best / 8e12:.4f} round_trip_exact={ok}", flush=True)
assert ok
