"""EXPERIMENT (library built with -DDXTLT_WG_TIMING): phase marks of the single-buffer aligned forward kernel (fwd_tiled), lane 0 of every
workgroup: kernel start -> tile begins -> load arrived -> behind the barrier -> store issued.  PROBE_FMT=bc3 (1 GiB buffer)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dxt_lossless_transform_amd as pkg
from dxt_lossless_transform_amd import _lib
fmt = os.environ.get("PROBE_FMT", "bc3")
st = pkg.Bc1TransformSettings() if fmt == "bc1" else pkg.Bc3TransformSettings()
if os.environ.get("PROBE_SETTINGS"):          # variant,split_alpha,split_colour
    v, sa, sc = (int(t) for t in os.environ["PROBE_SETTINGS"].split(","))
    st = pkg.Bc1TransformSettings(pkg.YCoCgVariant(v), bool(sc)) if fmt == "bc1" else pkg.Bc3TransformSettings(pkg.YCoCgVariant(v), bool(sa), bool(sc))
drop = int(os.environ.get("PROBE_DROP_BLOCKS", "0")) * (8 if fmt == "bc1" else 16)      # e.g. -1 -> 2^k + 1 blocks: halo tiles
x = torch.empty((1 << 30) - drop, dtype=torch.uint8, device="cuda:0"); pkg.fill_splitmix64(x, 3)
y = torch.empty_like(x)
f = getattr(pkg, f"transform_{fmt}_with_settings")
for _ in range(20):
    f(x, y, st)
torch.cuda.synchronize()
lib = _lib.load()
wgs = min(1 << 20, (1 << 30) // (2048 if fmt == "bc1" else 4096))
m = np.zeros(8 * wgs, dtype=np.uint32)
lib.dxtlt_debug_read_wg_marks_single.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.dxtlt_debug_read_wg_marks_single(m.ctypes.data, m.size) == 0
m = m.reshape(-1, 8).astype(np.int64)
ph = [((m[:, i] - m[:, 0]) & 0xFFFFFFFF) * 0.01 for i in (1, 2, 3, 4)]
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    f(x, y, st)
b.record(); torch.cuda.synchronize()
print(fmt, os.environ.get("PROBE_SETTINGS", "default"), "drop", os.environ.get("PROBE_DROP_BLOCKS", "0"), "fraction of peak", round(2 * x.numel() / (a.elapsed_time(b) / 10 * 1e-3) / 8e12, 4))
m = m[(m[:, 1] != 0) & (m[:, 4] != 0)]
print(fmt, f"forward tile: start -> tile begins {ph[0].mean():.2f} us -> load arrived {ph[1].mean():.2f} -> behind the barrier {ph[2].mean():.2f} -> store issued {ph[3].mean():.2f}")
