"""EXPERIMENT (needs a library built with -DDXTLT_WG_TIMING: tools/ab_build_rev.sh WORKTREE timing): per-workgroup durations of one
batch launch -- from the workgroup's first instruction to the acknowledgement of its last store, 100 MHz ticks -- by kind of tile.
PROBE_CASE=fmt:count:blocks:stride  PROBE_INVERSE=0/1"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import dxt_lossless_transform_amd as pkg
from dxt_lossless_transform_amd import batch, _lib
case = os.environ.get("PROBE_CASE", "bc3:4096:16383:262144").split(":")
inverse = os.environ.get("PROBE_INVERSE", "0") == "1"
dev = torch.device("cuda:0")
if case[0] == "corpus":          # corpus:<fmt>:<scale>: bench.py's corpus textures (a fraction of them: the timing array holds 2^20 workgroups)
    import bench
    fmt, scale = case[1], float(case[2])
    B = 8 if fmt == "bc1" else 16
    st = pkg.Bc1TransformSettings() if fmt == "bc1" else pkg.Bc3TransformSettings()
    texs = bench.corpus_textures(scale)
    if fmt != "bc1":
        texs = texs[::2]
    offs, arena = bench.corpus_layout(texs, B)
    big = torch.empty(arena, dtype=torch.uint8, device=dev); pkg.fill_splitmix64(big, 5)
    outb = torch.empty_like(big)
    prep = batch.prepare_batch([(fmt, inverse, big[o:o + n * B], outb[o:o + n * B], st) for (_, _, n), o in zip(texs, offs)])
    wgs = min(1 << 20, sum((n * B + 4095) // 4096 + 1 for _, _, n in texs))
else:
    fmt, count, blocks, stride = case[0], int(case[1]), int(case[2]), int(case[3])
    B = 8 if fmt == "bc1" else 16
    st = pkg.Bc1TransformSettings() if fmt == "bc1" else pkg.Bc3TransformSettings()
    n = blocks * B
    big = torch.empty(count * stride, dtype=torch.uint8, device=dev); pkg.fill_splitmix64(big, 5)
    outb = torch.empty_like(big)
    prep = batch.prepare_batch([(fmt, inverse, big[i * stride:i * stride + n], outb[i * stride:i * stride + n], st) for i in range(count)])
    wgs = min(1 << 20, count * ((blocks + (4096 // B) - 1) // (4096 // B) + 1))
for _ in range(20):
    batch.run_prepared_batch(prep)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros(4 * wgs, dtype=np.uint32)
lib.dxtlt_debug_read_wg_timing.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.dxtlt_debug_read_wg_timing(buf.ctypes.data, buf.size) == 0
t, kind, start, xcc = buf[0::4].astype(np.float64) * 0.01, buf[1::4], buf[2::4].astype(np.int64), buf[3::4]      # microseconds
names = {1: "aligned tile", 2: "whole halo / shifted tile", 3: "edge tile 0", 4: "edge tile at the end"}
print(os.environ.get("PROBE_CASE"), "inverse" if inverse else "forward")
for k in (1, 2, 3, 4):
    sel = t[kind == k]
    if sel.size:
        print(f"  {names[k]:28s} n {sel.size:7d}  mean {sel.mean():6.2f} us  median {np.median(sel):6.2f}  p90 {np.percentile(sel, 90):6.2f}  max {sel.max():7.2f}")
live = kind != 0
s0 = (start[live] - start[live].min()) & 0xFFFFFFFF
end = s0 * 0.01 + t[live]
span = end.max()
print(f"  launch span {span:.1f} us; sum of workgroup durations / span = {t[live].sum() / span:.0f} workgroups in flight on average (2048 slots)")
# workgroups in flight over time, in 20 slices of the span
edges = np.linspace(0, span, 21)
starts_us = s0 * 0.01
inflight = [int(((starts_us < (a + b) / 2) & (end > (a + b) / 2)).sum()) for a, b in zip(edges[:-1], edges[1:])]
print("  in flight at the middle of each twentieth of the span:", inflight)
# start order: how far behind its predecessor in the grid does a workgroup start (dispatch is in order)?
order = np.argsort(np.nonzero(live)[0])
d = np.diff(starts_us)
print(f"  start-to-start gap of consecutive workgroups: mean {d.mean() * 1000:.1f} ns, p99 {np.percentile(d, 99) * 1000:.0f} ns, max {d.max():.2f} us; workgroups started before their predecessor: {(d < 0).mean():.3f}")
k4 = np.nonzero(kind[live] == 4)[0]
if k4.size:
    nxt = k4[k4 + 1 < starts_us.size] + 1
    print(f"  gap in front of the workgroup that follows an end edge tile: mean {np.mean(starts_us[nxt] - starts_us[nxt - 1]) * 1000:.1f} ns")
print("  workgroups per XCC:", np.bincount(xcc[live]).tolist())
print("  sum of workgroup durations per XCC (ms):", [round(float(t[live][xcc[live] == x].sum()) / 1000, 2) for x in range(8)])

# phase marks of lane 0 (experiment build): 1 = lookup done / tile starts, 2 = loads arrived, 3 = behind the barrier, 4 = stores issued
marks = np.zeros(8 * wgs, dtype=np.uint32)
if hasattr(lib, "dxtlt_debug_read_wg_marks"):
    lib.dxtlt_debug_read_wg_marks.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    if lib.dxtlt_debug_read_wg_marks(marks.ctypes.data, marks.size) == 0:
        m = marks.reshape(-1, 8).astype(np.int64)
        for k in (1, 2):
            sel = (kind == k) & (m[:, 1] != 0)
            if not sel.any():
                continue
            st0 = start[sel]
            ph = [((m[sel, i] - st0) & 0xFFFFFFFF) * 0.01 for i in (1, 2, 3, 4)]
            total = t[sel]
            print(f"  {names[k]}: start -> tile begins {ph[0].mean():.2f} us -> loads arrived {ph[1].mean():.2f} -> behind the barrier {ph[2].mean():.2f} -> stores issued {ph[3].mean():.2f} -> all acknowledged {total.mean():.2f}")
