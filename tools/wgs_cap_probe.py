"""Kernels that move more than 16 bytes per lane -- the all-modes normalisation kernels (one read, 3 / 3 / 12 copies written)
and the fused auto-candidate kernels -- for a run under `rocprofv3 --kernel-trace --stats` with DXTLT_EXPERIMENT_WGS_PER_CU
set: does a cap on the resident workgroups per CU pay for them as it does for the decoders?  1 GiB of blocks each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import ctypes as C
import numpy as np
import torch
import dxt_lossless_transform_amd as pkg
from dxt_lossless_transform_amd import normalize as n1, normalize23 as n23
import cabi

dev = torch.device("cuda:0")
nbytes = 1 << 30
x = torch.empty(nbytes, dtype=torch.uint8, device=dev); pkg.fill_splitmix64(x, 77)
outs = [torch.empty_like(x) for _ in range(3)]
for _ in range(5):
    n1.normalize_blocks_all_modes(x, outs)
    n23.normalize_blocks_all_modes("bc2", x, outs)
half = x[: nbytes // 4]
outs12 = [torch.empty_like(half) for _ in range(12)]
for _ in range(5):
    n23.normalize_blocks_all_modes("bc3", half, outs12)
torch.cuda.synchronize()
# the auto kernels: host entry point, 256 MiB, length estimator in C
lib = cabi.bind(C.CDLL(pkg._lib.lib_path()))
h = np.frombuffer(x[: 256 << 20].cpu().numpy(), dtype=np.uint8)
y = np.zeros_like(h)
for n in (1, 2, 3):
    for use_all in (False, True):
        est = cabi.zstd_c_estimator(None)[0]
        settings = {1: cabi.CoreSettings2, 2: cabi.CoreSettings2, 3: cabi.CoreSettings3}[n]()
        f = getattr(lib, f"dltbc{n}core_transform_auto")
        for _ in range(3):
            r = f(h.ctypes.data, h.size, y.ctypes.data, y.size, C.byref(est), cabi.AutoSettings(use_all), C.byref(settings))
            assert r.ErrorCode == 0
print("done")
