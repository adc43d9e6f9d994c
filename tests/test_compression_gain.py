"""What the transform is for: a generic compressor does better on the transformed bytes.  The reference states it on
texture corpora (api/dxt-lossless-transform-bc1-api/README.MD:259-266, bc3-api/README.MD:33-59: zstd, zlib, 7z all gain);
here the reference's own 256x256 test textures (tests/golden/, from src/assets/tests/r2-256-bc*.dds) go through the
oracle and through zlib / lzma and -- when the system libzstd is there (tools/zstd_ratio.py) -- zstd -3 / -19, the
levels BASELINE.json configs[4] names.  The GPU leg of the same check is bench.py's archive workload."""
import lzma
import os
import zlib

import numpy as np
import pytest

from tools import zstd_ratio

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def payload(fmt):
    return np.fromfile(os.path.join(GOLDEN, f"r2-256-{fmt}.payload.bin"), dtype=np.uint8)


def sizes_of(data):
    raw = data.tobytes()
    out = {"zlib6": len(zlib.compress(raw, 6)), "lzma": len(lzma.compress(raw, preset=6))}
    if zstd_ratio.available():
        out["zstd3"] = zstd_ratio.compressed_size(raw, 3)
        out["zstd19"] = zstd_ratio.compressed_size(raw, 19)
    return out


@pytest.mark.parametrize("fmt", ["bc1", "bc2", "bc3"])
def test_transform_helps_every_compressor_on_the_reference_texture(oracle, fmt):
    x = payload(fmt)
    plain = sizes_of(x)
    default = oracle.transform(fmt, x, 1, True, True)
    assert np.array_equal(oracle.transform(fmt, default, 1, True, True, inverse=True), x)
    got = sizes_of(default)
    # default settings: every compressor gains, except the fast zstd level on the noise-like BC1 texture (+1 %) -- the
    # case transform_auto exists for: the best of the candidate settings gains there too
    for name in plain:
        if (fmt, name) != ("bc1", "zstd3"):
            assert got[name] < plain[name], (fmt, name, plain[name], got[name])
    best = dict(got)
    for variant in (0, 1):
        for split in (False, True):
            cand = sizes_of(oracle.transform(fmt, x, variant, split, split))
            best = {k: min(best[k], cand[k]) for k in best}
    for name in plain:
        assert best[name] < plain[name], (fmt, name, plain[name], best[name])


def test_zstd_binding_round_trips():
    if not zstd_ratio.available():
        pytest.skip("no libzstd in this image")
    x = payload("bc1")
    for level in (1, 3, 19):
        blob = zstd_ratio.compress(x, level)
        assert zstd_ratio.decompress(blob, x.size) == x.tobytes()
    assert zstd_ratio.compressed_size(x, 19) <= zstd_ratio.compressed_size(x, 3)
