"""configs[4] slice (32 alternating 256 MiB BC1 / BC3 textures, default settings): one call per texture on 1 / 2 / 4 HIP
streams round robin -- do the tails of consecutive kernels overlap?  Fraction of 8 TB/s on 2 * bytes per direction."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dxt_lossless_transform_amd as pkg
dev = torch.device("cuda:0")
tex, k = 256 << 20, 32
x = torch.empty(k * tex, dtype=torch.uint8, device=dev); pkg.fill_splitmix64(x, 5)
y = torch.empty_like(x); z = torch.empty_like(x)
xs, ys, zs = (list(t.view(k, tex).unbind(0)) for t in (x, y, z))
fm = ["bc1" if i % 2 == 0 else "bc3" for i in range(k)]
st = {"bc1": pkg.Bc1TransformSettings(), "bc3": pkg.Bc3TransformSettings()}
F = {f: getattr(pkg, f"transform_{f}_with_settings") for f in st}
G = {f: getattr(pkg, f"untransform_{f}_with_settings") for f in st}
for ns in (1, 2, 4, 1, 2):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    def run(fn, src, dst):
        for i in range(k):
            with torch.cuda.stream(streams[i % ns]):
                fn[fm[i]](src[i], dst[i], st[fm[i]])
    res = []
    for fn, src, dst in ((F, xs, ys), (G, ys, zs)):
        for _ in range(2): run(fn, src, dst)
        torch.cuda.synchronize()
        a = torch.cuda.Event(enable_timing=True); a.record()
        steps = 10
        ends = []
        for s in streams: s.wait_event(a)
        for _ in range(steps): run(fn, src, dst)
        for s in streams:
            e = torch.cuda.Event(enable_timing=True); e.record(s); ends.append(e)
        torch.cuda.synchronize()
        t = max(a.elapsed_time(e) for e in ends) / steps * 1e-3
        res.append(round(2 * k * tex / t / 8e12, 4))
    assert torch.equal(x, z)
    print(f"streams {ns}: fwd {res[0]} inv {res[1]}", flush=True)
