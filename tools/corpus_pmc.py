"""Workload for PMC passes (tools/pmc_passes.py): the corpus-shaped batch (bench.py run_corpus_leg's textures, PROBE_SCALE of
them) forward + inverse x3 through dxtlt_transform_batch_device, and beside it ONE buffer of the same bytes with a mip-chain
block count (odd, = 23 mod 64) through the single-buffer call: halo tiles forward, shifted tiles inverse."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import dxt_lossless_transform_amd as pkg
from dxt_lossless_transform_amd import batch

fmt = sys.argv[1] if len(sys.argv) > 1 else "bc3"
scale = float(os.environ.get("PROBE_SCALE", "0.25"))
B = pkg.BLOCK_BYTES[fmt]
st = pkg.Bc3TransformSettings() if fmt == "bc3" else pkg.Bc1TransformSettings()
texs = bench.corpus_textures(scale)
if fmt != "bc1":
    texs = texs[::2]
offs, arena = bench.corpus_layout(texs, B)
dev = torch.device("cuda:0")
x = torch.empty(arena, dtype=torch.uint8, device=dev); pkg.fill_splitmix64(x, 3)
y = torch.empty_like(x); z = torch.empty_like(x)
views = [(x[o:o + n * B], y[o:o + n * B], z[o:o + n * B]) for (_, _, n), o in zip(texs, offs)]
pf = batch.prepare_batch([(fmt, False, a, b, st) for a, b, _ in views])
pi = batch.prepare_batch([(fmt, True, b, c, st) for _, b, c in views])
total = sum(n for _, _, n in texs)
one = total // 64 * 64 + 23            # one buffer of the same volume, block count = 23 mod 64 like a square mip chain
for _ in range(3):
    batch.run_prepared_batch(pf); batch.run_prepared_batch(pi)
    getattr(pkg, f"transform_{fmt}_with_settings")(x[:one * B], y[:one * B], st)
    getattr(pkg, f"untransform_{fmt}_with_settings")(y[:one * B], z[:one * B], st)
torch.cuda.synchronize()
print("corpus bytes", total * B, "textures", len(texs), "single buffer blocks", one)
