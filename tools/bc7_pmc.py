#!/usr/bin/env python3
"""Workload for PMC passes on the BC7 kernels: 1 GiB, forward + inverse x3.  argv[1]: uniform | mode6 | skewed"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dxt_lossless_transform_amd as pkg  # noqa: E402
from dxt_lossless_transform_amd import bc7  # noqa: E402

dist = sys.argv[1] if len(sys.argv) > 1 else "uniform"
n = (1 << 30) // 16
x = torch.empty(n * 16, dtype=torch.uint8, device="cuda:0")
pkg.fill_splitmix64(x, 0x0BC70004)
b = x.view(-1, 16)
r = b[:, 15].to(torch.int32)
m = (r & 7) if dist == "uniform" else torch.full_like(r, 6) if dist == "mode6" else torch.where(r < 140, 6, torch.where(r < 200, 1, torch.where(r < 230, 3, r & 7))).to(torch.int32)
low = ((2 << m) - 1).to(torch.uint8)
b[:, 0] = (b[:, 0] & ~low) | (1 << m).to(torch.uint8)
y, z = torch.empty_like(x), torch.empty_like(x)
for _ in range(3):
    bc7.transform_bc7(x, y)
    bc7.untransform_bc7(y, z)
torch.cuda.synchronize()
assert torch.equal(x, z)
