"""BC1 without the colour split runs its forward HALO tiles 0.026 under the other seven settings (0.786-0.788 against 0.811-0.815 at
2^k + 1 blocks, profiles/r03_settings_sweep.txt) although it executes FEWER instructions.  What is it?

    python tools/bc1_nosplit_probe.py time          fraction of 8 TB/s, forward / inverse, 4 GiB of BC1, Variant1, split_colour 1 / 0:
                                                      aligned tiles at 64 / 128 / 256 / 512 lanes (is it the 256-lane tile shape?),
                                                      halo / shifted tiles forced on ALIGNED data (the kernel without any misalignment),
                                                      2^29 + {1, 3, 16, 17, 33, 63} blocks (which shifts?)
    python tools/bc1_nosplit_probe.py pmc256 SC     the same for the ALIGNED kernel at 256 lanes on 2^29 blocks (the shape without a halo)
    python tools/bc1_nosplit_probe.py pmc SC        the workload for rocprofv3 --pmc passes (tools/pmc_passes.py): 2^29 + 1 blocks,
                                                      split_colour = SC, three forward + inverse pairs
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import dxt_lossless_transform_amd as pkg  # noqa: E402

dev = torch.device("cuda:0")
BASE = 1 << 29


def settings(sc):
    return pkg.Bc1TransformSettings(pkg.YCoCgVariant(1), bool(sc))


def timed(fn, steps=30):
    for _ in range(3):
        fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(steps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / steps


def measure(n, sc, threads=0, force=0):
    x = torch.empty(8 * n, dtype=torch.uint8, device=dev)
    pkg.fill_splitmix64(x, 0xB1)
    y, z = torch.empty_like(x), torch.empty_like(x)
    st = settings(sc)
    pkg.set_tuning(threads, force)
    try:
        f = lambda: pkg.transform_bc1_with_settings(x, y, st)
        g = lambda: pkg.untransform_bc1_with_settings(y, z, st)
        tf, ti = timed(f), timed(g)
        tf, ti = min(tf, timed(f)), min(ti, timed(g))
        assert torch.equal(x, z)
    finally:
        pkg.set_tuning(0, 0)
    return 16 * n / (tf * 1e-3) / 8e12, 16 * n / (ti * 1e-3) / 8e12


if sys.argv[1] == "pmc256":
    # the ALIGNED kernel at 256 lanes, where the deficit shows without any halo: 2^29 blocks, split_colour = SC
    sc = int(sys.argv[2])
    x = torch.empty(8 * BASE, dtype=torch.uint8, device=dev)
    pkg.fill_splitmix64(x, 0xB1)
    y, z = torch.empty_like(x), torch.empty_like(x)
    pkg.set_tuning(256, 0)
    for _ in range(3):
        pkg.transform_bc1_with_settings(x, y, settings(sc))
        pkg.untransform_bc1_with_settings(y, z, settings(sc))
    torch.cuda.synchronize()
    pkg.set_tuning(0, 0)
    assert torch.equal(x, z)
    sys.exit(0)
if sys.argv[1] == "pmc":
    sc = int(sys.argv[2])
    n = BASE + 1
    x = torch.empty(8 * n, dtype=torch.uint8, device=dev)
    pkg.fill_splitmix64(x, 0xB1)
    y, z = torch.empty_like(x), torch.empty_like(x)
    for _ in range(3):
        pkg.transform_bc1_with_settings(x, y, settings(sc))
        pkg.untransform_bc1_with_settings(y, z, settings(sc))
    torch.cuda.synchronize()
    assert torch.equal(x, z)
    sys.exit(0)

warm = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
for _ in range(50):       # clock ramp
    warm.add_(1)
torch.cuda.synchronize()
del warm
print("BC1 Variant1, 4 GiB, fraction of 8 TB/s on 2 * len; columns: split_colour=1 fwd / inv, split_colour=0 fwd / inv")
rows = [(f"aligned tiles, {t:3d} lanes", BASE, t, 0) for t in (64, 128, 256, 512)]
rows += [("halo / shifted tiles forced on aligned data", BASE, 0, 2)]
rows += [(f"2^29 + {e:2d} blocks", BASE + e, 0, 0) for e in (1, 3, 16, 17, 33, 63)]
for label, n, threads, force in rows:
    a = measure(n, 1, threads, force)
    b = measure(n, 0, threads, force)
    print(f"{label:46s} {a[0]:.4f} / {a[1]:.4f}    {b[0]:.4f} / {b[1]:.4f}    (no split - split: fwd {b[0] - a[0]:+.4f}, inv {b[1] - a[1]:+.4f})", flush=True)
