"""CPU restatement of transform_bcN_auto (TEST INFRASTRUCTURE ONLY) on top of the C oracle.

Follows /root/reference/src/core/dxt-lossless-transform-bc1/src/transform/transform_auto.rs:200-270 (BC2 twin
:196-, BC3 :196-294) with the test orders of settings.rs (bc1/bc2 :81-98, bc3 :91-121): full transform per
candidate, estimator on the endpoint section(s) only, strict `<`, re-transform when the best was not the last."""
from __future__ import annotations

from . import oracle_c

# (variant, split_alpha, split_colour), core numbering
FAST_12 = [(0, 0, 0), (0, 0, 1), (1, 0, 0), (1, 0, 1)]
ALL_12 = [(2, 0, 0), (0, 0, 0), (0, 0, 1), (3, 0, 0), (3, 0, 1), (2, 0, 1), (1, 0, 0), (1, 0, 1)]
FAST_3 = [(1, 1, 0), (1, 1, 1), (0, 1, 0), (0, 0, 1), (0, 1, 1), (1, 0, 1), (0, 0, 0), (1, 0, 0)]
ALL_3 = [(2, 1, 0), (2, 1, 1), (3, 1, 1), (3, 1, 0), (1, 1, 0), (3, 0, 1), (1, 1, 1), (2, 0, 1),
         (2, 0, 0), (3, 0, 0), (0, 1, 0), (0, 0, 1), (0, 1, 1), (1, 0, 1), (0, 0, 0), (1, 0, 0)]


def test_order(fmt: str, use_all: bool):
    if fmt == "bc3":
        return ALL_3 if use_all else FAST_3
    return ALL_12 if use_all else FAST_12


def transform_auto(fmt: str, data, estimate, use_all: bool):
    """estimate(bytes_like) -> int.  Returns ((variant, split_alpha, split_colour), output, calls) where calls is
    the list of (offset, length) sections handed to the estimator, in order."""
    n = len(data)
    blocks = n // oracle_c.BLOCK[fmt]
    best = (1, 1 if fmt == "bc3" else 0, 1)
    best_size = None
    last = best
    out = None
    calls = []
    for cand in test_order(fmt, use_all):
        v, sa, sc = cand
        out = oracle_c.transform(fmt, data, v, sc, sa)
        last = cand
        if fmt == "bc1":
            sections = [(0, n // 2)]
        elif fmt == "bc2":
            sections = [(n // 2, n // 4)]
        else:
            sections = [(0, blocks * 2), (n // 2, blocks * 4)]
        size = 0
        for off, ln in sections:
            calls.append((off, ln))
            size += estimate(out[off:off + ln])
        if best_size is None or size < best_size:
            best_size, best = size, cand
    if best != last:
        out = oracle_c.transform(fmt, data, best[0], best[2], best[1])
    return best, out, calls


def transform_bc1_auto_with_normalization(data, estimate, use_all: bool):
    """experimental/normalize_blocks/transform.rs:222-333.  estimate(bytes_like) -> int, or raises to signal an
    estimator error (the candidate is then skipped, transform.rs:403).  Returns ((norm, variant, split), output, calls)
    with calls = the estimator call lengths in order; when no block can be normalised the plain auto transform runs
    (its calls are (offset, length) pairs, see transform_auto)."""
    n = len(data)
    outs, any_normalized = oracle_c.normalize_bc1_blocks_all_modes(data)
    if not any_normalized:
        (v, _sa, sc), out, calls = transform_auto("bc1", data, estimate, use_all)
        return (0, v, sc), out, calls
    best, best_size, calls = (0, 1, 1), None, []
    for norm in range(3):
        for v, _sa, sc in test_order("bc1", use_all):
            cand = oracle_c.transform("bc1", outs[norm], v, sc)
            calls.append(n // 2)
            try:
                size = estimate(cand[: n // 2])
            except Exception:
                continue
            if best_size is None or size < best_size:
                best_size, best = size, (norm, v, sc)
    out = oracle_c.transform_bc1_with_normalize_blocks(data, best[0], best[1], bool(best[2]))
    return best, out, calls
