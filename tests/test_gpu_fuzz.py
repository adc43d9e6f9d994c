"""Seeded random sweep over what a caller can vary at once: format, settings, block count (1 .. ~70 000, biased towards
tile boundaries), the alignment of both device pointers, whole buffer or two ranges of it, and the experiment switches
that select kernel families.  Every case: forward == oracle byte for byte, nothing written outside the output, inverse
gives the source back.  The structured tests cover each axis on its own; this one covers their combinations (it is the
kind of test that caught a store hazard which only showed once a store sat inside an unrolled loop)."""
import numpy as np
import pytest

from helpers import BLOCK, all_settings, pkg_settings, settings_id

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

CASES = 1500
TILE = {"bc1": 512, "bc2": 256, "bc3": 256}


def pick_count(rng, fmt):
    t = TILE[fmt]
    kind = rng.integers(0, 4)
    if kind == 0:
        return int(rng.integers(1, 70))
    if kind == 1:
        return int(rng.integers(1, 40) * t + rng.integers(-17, 18))
    if kind == 2:
        return int(rng.integers(1, 70_000))
    return int(rng.integers(1, 300) * 16 + rng.integers(0, 2) * rng.integers(0, 16))


def test_random_combinations(pkg, oracle):
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0xF0220)
    guard = 64
    try:
        for case in range(CASES):
            fmt = ("bc1", "bc2", "bc3")[int(rng.integers(0, 3))]
            B = BLOCK[fmt]
            settings = list(all_settings(fmt))
            s = settings[int(rng.integers(0, len(settings)))]
            st = pkg_settings(pkg, fmt, s)
            n = max(1, pick_count(rng, fmt))
            in_off = int(rng.choice([0, 0, 0, 16, 8, 4, 1, 3]))
            out_off = int(rng.choice([0, 0, 0, 16, 8, 4, 2, 1, 5]))
            force = int(rng.choice([0, 0, 0, 1, 2, 0x20, 2 | 0x20, 0x100, 0x200]))
            x = rng.integers(0, 256, n * B, dtype=np.uint8)
            want = oracle.transform(fmt, x, s[0], s[2], s[1])
            src = torch.zeros(n * B + 2 * guard, dtype=torch.uint8, device=dev)
            src[in_off:in_off + n * B] = torch.from_numpy(x).to(dev)
            dst = torch.full((n * B + 2 * guard,), 0xA5, dtype=torch.uint8, device=dev)
            back = torch.full((n * B + 2 * guard,), 0x5A, dtype=torch.uint8, device=dev)
            xin, yout, zout = src[in_off:in_off + n * B], dst[out_off:out_off + n * B], back[in_off:in_off + n * B]
            tag = (case, fmt, settings_id(s), n, in_off, out_off, hex(force))
            pkg.set_tuning(0, force)
            if rng.integers(0, 3) == 0 and n >= 2:
                cut = int(rng.integers(1, n))     # two ranges of one buffer, as a sharded caller issues them
                for first, count in ((0, cut), (cut, n - cut)):
                    pkg.transform_range(fmt, False, xin[first * B:], yout, n, first, count, st)
                for first, count in ((cut, n - cut), (0, cut)):
                    pkg.transform_range(fmt, True, yout, zout[first * B:], n, first, count, st)
            else:
                getattr(pkg, f"transform_{fmt}_with_settings")(xin, yout, st)
                getattr(pkg, f"untransform_{fmt}_with_settings")(yout, zout, st)
            got = dst.cpu().numpy()
            assert np.array_equal(got[out_off:out_off + n * B], want), ("forward",) + tag
            assert (got[:out_off] == 0xA5).all() and (got[out_off + n * B:] == 0xA5).all(), ("forward wrote outside",) + tag
            rt = back.cpu().numpy()
            assert np.array_equal(rt[in_off:in_off + n * B], x), ("inverse",) + tag
            assert (rt[:in_off] == 0x5A).all() and (rt[in_off + n * B:] == 0x5A).all(), ("inverse wrote outside",) + tag
    finally:
        pkg.set_tuning(0, 0)
