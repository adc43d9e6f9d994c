#!/usr/bin/env python3
"""Workload for PMC passes on the shifted tiles: BC3 (default settings) forward and inverse on 2^26 (+ extra) blocks.
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -- python3 tools/odd_pmc.py 1"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dxt_lossless_transform_amd as pkg  # noqa: E402

extra = int(sys.argv[1]) if len(sys.argv) > 1 else 1
force = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0      # experiment switches of dxtlt_set_tuning (0x100: identity tile order)
n = (1 << 26) + extra
x = torch.empty(16 * n, dtype=torch.uint8, device="cuda:0")
pkg.fill_splitmix64(x, 3)
y = torch.empty_like(x)
z = torch.empty_like(x)
pkg.set_tuning(0, force)
for _ in range(3):
    pkg.transform_bc3_with_settings(x, y)
    pkg.untransform_bc3_with_settings(y, z)
torch.cuda.synchronize()
assert torch.equal(x, z)
