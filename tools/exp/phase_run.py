import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import dxt_lossless_transform_amd as pkg
from dxt_lossless_transform_amd import bc7
dist = sys.argv[1] if len(sys.argv) > 1 else "uniform"
dev = torch.device("cuda:0")
n = (1 << 30) // 16
x = torch.empty(n * 16, dtype=torch.uint8, device=dev); pkg.fill_splitmix64(x, 0x0BC70004)
b = x.view(-1, 16); r = b[:, 15].to(torch.int32)
m = (r & 7) if dist == "uniform" else torch.full_like(r, 6)
low = ((2 << m) - 1).to(torch.uint8); b[:, 0] = (b[:, 0] & ~low) | ((1 << m) & 0xFF).to(torch.uint8)
y = torch.empty_like(x)
for _ in range(3):
    bc7.transform_bc7(x, y)
torch.cuda.synchronize()
lib = C.CDLL(pkg._lib.lib_path())
buf = (C.c_ulonglong * (512 * 16))()
assert lib.dxtlt_dbg_read(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(512, 16).astype(np.int64)
a = a[a[:, 0] != 0][:120]
idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 15]
names = ["entry->before b1 (loads issued)", "barrier 1", "rank+count (loads arrive)", "barrier 2", "bases + scatter raw", "barrier 3", "read sorted", "barrier 4",
         "permute + image writes", "barrier 5", "image reads + stores issued"]
d = np.diff(a[:, idx], axis=1)
tot = (a[:, 15] - a[:, 0])
print(dist, "samples", a.shape[0], "WG lifetime (wave 0) cycles: median", int(np.median(tot)), "mean", int(tot.mean()))
for nme, col in zip(names, d.T):
    print(f"  {nme:36s} median {int(np.median(col)):7d}  mean {int(col.mean()):7d}  ({100 * col.mean() / tot.mean():5.1f} %)")
