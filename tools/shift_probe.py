"""Shifted-tile kernels by alignment class of the stream bases (4 GiB, fractions of 8 TB/s on 2*len): which part of the
odd-block-count cost is the kernel's own structure and which is misalignment.  force bits: 2 = shifted tiles even when
aligned, 0x20 = generic (switch per access) LDS scatter/gather, 0x40 = no write-through on whole lines, 0x10 = partial segments left out (timing only), 0x100 / 0x200 = XCD-contiguous
tile order off / on.  PROBE_ONLY=<substring> selects cases."""
import json, sys, os
os.environ.setdefault("DXTLT_TIMING_EXPERIMENTS", "1")   # the 0x10 cases (wrong output, timing only)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import dxt_lossless_transform_amd as pkg
dev = torch.device("cuda:0")
def timed(fn, steps=40):
    for _ in range(2): fn()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(steps): fn()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / steps
res = {}
rounds = 2
for fmt, B in (("bc3", 16), ("bc1", 8)):
    if os.environ.get("PROBE_FMT") and os.environ["PROBE_FMT"] != fmt:
        continue
    base = (4 << 30) // B
    for label, n, force in (("aligned128", base, 0), ("aligned128_forced_shift", base, 2), ("aligned128_forced_shift_generic_lds", base, 2 | 0x20),
                            ("aligned16_only", base + 16, 0), ("aligned16_forced_shift", base + 16, 2),
                            ("aligned128_forced_shift_no_xcd_remap", base, 2 | 0x100), ("aligned128_xcd_remap", base, 0x200),
                            ("odd", base + 1, 0), ("odd_first_form_partial_segments", base + 1, 0x400), ("odd_no_halo_load_WRONG_OUTPUT", base + 1, 0x10), ("odd_nt_stores_only", base + 1, 0x800), ("aligned128_forced_shift_no_halo_load", base, 2 | 0x10), ("aligned128_forced_shift_nt_only", base, 2 | 0x800), ("odd_generic_lds", base + 1, 0x20), ("odd_no_xcd_remap", base + 1, 0x100),
                            ("odd_no_line_policy", base + 1, 0x40), ("odd_plain_shared_lines", base + 1, 0x80),  ("odd_without_partial_segments_WRONG_OUTPUT", base + 1, 0x10), ("odd3", base + 3, 0), ("plus8", base + 8, 0),
                            ("plus24", base + 24, 0), ("plus40", base + 40, 0), ("plus63", base + 63, 0),
                            ("plus24_xcd_contiguous", base + 24, 0x200), ("plus40_identity_order", base + 40, 0x100), ("plus63_identity_order", base + 63, 0x100), ("plus8_identity_order", base + 8, 0x100),
                            ("odd3_xcd_contiguous", base + 3, 0x200), ("plus63_xcd_contiguous", base + 63, 0x200), ("odd_identity_order", base + 1, 0x100)):
        if os.environ.get("PROBE_ONLY") and os.environ["PROBE_ONLY"] not in label:
            continue
        x = torch.empty(n * B, dtype=torch.uint8, device=dev); pkg.fill_splitmix64(x, 1)
        y = torch.empty_like(x); z = torch.empty_like(x)
        pkg.set_tuning(0, force)
        f = getattr(pkg, f"transform_{fmt}_with_settings"); g = getattr(pkg, f"untransform_{fmt}_with_settings")
        tf = timed(lambda: f(x, y)); ti = timed(lambda: g(y, z))
        assert force & 0x10 or torch.equal(x, z)
        tf = min(tf, timed(lambda: f(x, y))); ti = min(ti, timed(lambda: g(y, z)))   # best of two passes of 40
        res[f"{fmt}_{label}"] = [round(2 * n * B / (tf * 1e-3) / 8e12, 4), round(2 * n * B / (ti * 1e-3) / 8e12, 4)]
        pkg.set_tuning(0, 0)
        del x, y, z
for k, v in res.items():
    print(f"{k:44s} fwd {v[0]:.3f}  inv {v[1]:.3f}")
