"""One big odd buffer (block count = 23 mod 64, like a square mip chain), forward / inverse fractions of the HBM peak:
single-buffer call; batch call with that ONE buffer (regular-array path, no table); the same with DXTLT_BATCH_NO_STRIDED=1
semantics emulated by two unequal buffers (table lookup path).  PROBE_FMT=bc3 PROBE_GIB=2"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dxt_lossless_transform_amd as pkg
from dxt_lossless_transform_amd import batch
dev = torch.device("cuda:0")
fmt = os.environ.get("PROBE_FMT", "bc3")
B = pkg.BLOCK_BYTES[fmt]
st = pkg.Bc3TransformSettings() if fmt == "bc3" else pkg.Bc1TransformSettings()
blocks = int(float(os.environ.get("PROBE_GIB", "2")) * 2**30) // B // 64 * 64 + int(os.environ.get("PROBE_MOD", "23"))
n = blocks * B
x = torch.empty(n + 4096, dtype=torch.uint8, device=dev); pkg.fill_splitmix64(x, 5)
y = torch.empty_like(x); z = torch.empty_like(x)

def timed(fn, reps=10):
    for _ in range(6):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3

f = getattr(pkg, f"transform_{fmt}_with_settings"); g = getattr(pkg, f"untransform_{fmt}_with_settings")
out = {}
out["single"] = [round(2 * n / timed(lambda: f(x[:n], y[:n], st)) / 8e12, 4), round(2 * n / timed(lambda: g(y[:n], z[:n], st)) / 8e12, 4)]
p1f = batch.prepare_batch([(fmt, False, x[:n], y[:n], st)]); p1i = batch.prepare_batch([(fmt, True, y[:n], z[:n], st)])
out["batch_one_buffer"] = [round(2 * n / timed(lambda: batch.run_prepared_batch(p1f)) / 8e12, 4), round(2 * n / timed(lambda: batch.run_prepared_batch(p1i)) / 8e12, 4)]
# two unequal buffers with mip-like counts: the table path
b1 = (blocks // 3) // 64 * 64 + 23
b2 = (blocks - b1 - 64) // 64 * 64 + int(os.environ.get('PROBE_MOD2', '23'))
o2 = (b1 * B + 255) // 256 * 256
n2 = (b1 + b2) * B
items_f = [(fmt, False, x[:b1 * B], y[:b1 * B], st), (fmt, False, x[o2:o2 + b2 * B], y[o2:o2 + b2 * B], st)]
items_i = [(fmt, True, y[:b1 * B], z[:b1 * B], st), (fmt, True, y[o2:o2 + b2 * B], z[o2:o2 + b2 * B], st)]
p2f, p2i = batch.prepare_batch(items_f), batch.prepare_batch(items_i)
out["batch_two_unequal"] = [round(2 * n2 / timed(lambda: batch.run_prepared_batch(p2f)) / 8e12, 4), round(2 * n2 / timed(lambda: batch.run_prepared_batch(p2i)) / 8e12, 4)]
# three equal buffers at irregular offsets: every buffer owns the same number of workgroups (entry = wg / wgs per buffer, ONE
# table load) but the pointers are no arithmetic progression (no regular-array path)
b3 = (blocks // 3 - 4096) // 64 * 64 + 23
offs = [0, (b3 * B + 255) // 256 * 256 + 256 * 7, 0]
offs[2] = offs[1] + (b3 * B + 255) // 256 * 256 + 256 * 29
n3 = 3 * b3 * B
p3f = batch.prepare_batch([(fmt, False, x[o:o + b3 * B], y[o:o + b3 * B], st) for o in offs])
p3i = batch.prepare_batch([(fmt, True, y[o:o + b3 * B], z[o:o + b3 * B], st) for o in offs])
out["batch_three_equal_irregular"] = [round(2 * n3 / timed(lambda: batch.run_prepared_batch(p3f)) / 8e12, 4),
                                      round(2 * n3 / timed(lambda: batch.run_prepared_batch(p3i)) / 8e12, 4)]
print(os.environ.get("DXTLT_LIB_PATH", "default").split("/")[-1], json.dumps(out))
