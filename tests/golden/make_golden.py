#!/usr/bin/env python3
"""Regenerates the committed golden fixtures in this directory.

    python tests/golden/make_golden.py

The reference (Rust) cannot be built or run in this image, and it holds no transformed-byte vectors of its
own, so the expected outputs here are produced by the numpy restatement in oracle/oracle_np.py (written
independently of the C oracle and of the HIP kernels).  What the reference DOES pin is reproduced verbatim as
data: its three generator known-answer vectors and the split-endpoints vector (see KNOWN_ANSWERS below), and
the BC payloads of its three test textures (r2-256-bc{1,2,3}.payload.bin, extracted from
/root/reference/src/assets/tests/*.dds at offset 128 by the snippet in this docstring's sibling README).

Outputs:
  vectors.json   small cases: input and expected output bytes (hex) for every settings combination
  digests.json   sha256 of expected outputs for the larger seeded cases and the real textures
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle_np as onp  # noqa: E402

BLOCK = {"bc1": 8, "bc2": 16, "bc3": 16}

# --- data the reference's own tests pin (copied as data, not code) -------------------------------------
KNOWN_ANSWERS = {
    # dxt-lossless-transform-bc1/src/test_prelude.rs:107-119 (generate_bc1_test_data(3))
    "bc1_generator_3": "0001020380818283" "0405060784858687" "08090a0b88898a8b",
    # dxt-lossless-transform-bc2/src/test_prelude.rs:586-606 (generate_bc2_test_data(3))
    "bc2_generator_3": "000102030405060780818283c0c1c2c3" "08090a0b0c0d0e0f84858687c4c5c6c7"
                       "101112131415161788898a8bc8c9cacb",
    # dxt-lossless-transform-bc3/src/test_prelude.rs:1058-1078 (generate_bc3_test_data(3)), decimal there
    "bc3_generator_3": bytes([0, 1, 32, 33, 34, 35, 36, 37, 128, 129, 130, 131, 192, 193, 194, 195,
                              2, 3, 38, 39, 40, 41, 42, 43, 132, 133, 134, 135, 196, 197, 198, 199,
                              4, 5, 44, 45, 46, 47, 48, 49, 136, 137, 138, 139, 200, 201, 202, 203]).hex(),
    # dxt-lossless-transform-common/src/transforms/split_565_color_endpoints/tests.rs:140-152
    "split_565_input": "000110110405141508091819",
    "split_565_output": "000104050809101114151819",
    # colours exercised by the reference's YCoCg tests (decorrelate.rs:413-446 and
    # intrinsics/color_565/decorrelate/avx2.rs:194-266); the property pinned there is recorrelate(decorrelate(c)) == c
    "ycocg_roundtrip_colours": [0x0000, 0xFFFF, 0xF800, 0x07E0, 0x001F, 0x1234, 0x5678, 0x9ABC, 0xDEF0, 0x2468,
                                0x1357, 0xACE0, 0x1359, 0x7531, 0x9BDF, 0x4682, 0xCEA8],
}


def gen_bc1(n: int) -> np.ndarray:
    """bc1 test_prelude.rs:81-105"""
    b = np.arange(n, dtype=np.int64)[:, None] * 4 + np.arange(4)[None, :]
    out = np.empty((n, 8), dtype=np.uint8)
    out[:, 0:4] = (b % 256).astype(np.uint8)
    out[:, 4:8] = ((128 + b) % 256).astype(np.uint8)
    return out.reshape(-1)


def gen_bc2(n: int) -> np.ndarray:
    """bc2 test_prelude.rs:151-186"""
    i = np.arange(n, dtype=np.int64)[:, None]
    out = np.empty((n, 16), dtype=np.uint8)
    out[:, 0:8] = ((i * 8 + np.arange(8)[None, :]) % 256).astype(np.uint8)
    out[:, 8:12] = ((0x80 + i * 4 + np.arange(4)[None, :]) % 256).astype(np.uint8)
    out[:, 12:16] = ((0xC0 + i * 4 + np.arange(4)[None, :]) % 256).astype(np.uint8)
    return out.reshape(-1)


def gen_bc3(n: int) -> np.ndarray:
    """bc3 test_prelude.rs:45-101: each band wraps inside itself"""
    i = np.arange(n, dtype=np.int64)[:, None]
    out = np.empty((n, 16), dtype=np.uint8)
    out[:, 0:2] = ((i * 2) % 32 + np.arange(2)[None, :]).astype(np.uint8)
    out[:, 2:8] = (32 + (i * 6) % 96 + np.arange(6)[None, :]).astype(np.uint8)
    out[:, 8:12] = (128 + (i * 4) % 64 + np.arange(4)[None, :]).astype(np.uint8)
    out[:, 12:16] = (192 + (i * 4) % 64 + np.arange(4)[None, :]).astype(np.uint8)
    return out.reshape(-1)


GEN = {"bc1": gen_bc1, "bc2": gen_bc2, "bc3": gen_bc3}


def settings_of(fmt: str):
    for v in range(4):
        for sc in (0, 1):
            for sa in ((0, 1) if fmt == "bc3" else (0,)):
                yield v, sa, sc


def seeded(fmt: str, blocks: int, seed: int) -> np.ndarray:
    q = blocks * BLOCK[fmt] // 8
    return onp.splitmix64(seed, 0, q).astype("<u8").view(np.uint8)


def main() -> None:
    vectors = []
    for fmt in ("bc1", "bc2", "bc3"):
        for n in (1, 2, 3, 5, 17, 64):
            x = GEN[fmt](n)
            for v, sa, sc in settings_of(fmt):
                y = onp.transform(fmt, x, v, bool(sc), bool(sa))
                assert np.array_equal(onp.transform(fmt, y, v, bool(sc), bool(sa), inverse=True), x)
                vectors.append({"fmt": fmt, "source": "generator", "blocks": n, "variant": v, "split_alpha": sa,
                                "split_colour": sc, "input": x.tobytes().hex(), "output": y.tobytes().hex()})
        # a few random blocks so that every bit of the colour arithmetic is exercised
        x = seeded(fmt, 33, 0xD17A0000 + BLOCK[fmt])
        for v, sa, sc in settings_of(fmt):
            y = onp.transform(fmt, x, v, bool(sc), bool(sa))
            vectors.append({"fmt": fmt, "source": "splitmix64", "blocks": 33, "variant": v, "split_alpha": sa,
                            "split_colour": sc, "input": x.tobytes().hex(), "output": y.tobytes().hex()})

    digests = []
    big = {"bc1": (1 << 20, 0x0BC10001), "bc2": (1 << 18, 0x0BC20001), "bc3": (1 << 18, 0x0BC30001)}
    for fmt, (blocks, seed) in big.items():
        x = seeded(fmt, blocks, seed)
        for v, sa, sc in settings_of(fmt):
            y = onp.transform(fmt, x, v, bool(sc), bool(sa))
            digests.append({"fmt": fmt, "source": "splitmix64", "seed": seed, "blocks": blocks, "variant": v,
                            "split_alpha": sa, "split_colour": sc, "input_sha256": hashlib.sha256(x).hexdigest(),
                            "output_sha256": hashlib.sha256(y).hexdigest()})
        # ragged size: not a multiple of any tile
        blocks_r = 100003
        x = seeded(fmt, blocks_r, seed + 7)
        for v, sa, sc in settings_of(fmt):
            y = onp.transform(fmt, x, v, bool(sc), bool(sa))
            digests.append({"fmt": fmt, "source": "splitmix64", "seed": seed + 7, "blocks": blocks_r, "variant": v,
                            "split_alpha": sa, "split_colour": sc, "input_sha256": hashlib.sha256(x).hexdigest(),
                            "output_sha256": hashlib.sha256(y).hexdigest()})
        p = np.fromfile(os.path.join(HERE, f"r2-256-{fmt}.payload.bin"), dtype=np.uint8)
        for v, sa, sc in settings_of(fmt):
            y = onp.transform(fmt, p, v, bool(sc), bool(sa))
            digests.append({"fmt": fmt, "source": f"r2-256-{fmt}.payload.bin", "blocks": p.size // BLOCK[fmt],
                            "variant": v, "split_alpha": sa, "split_colour": sc,
                            "input_sha256": hashlib.sha256(p).hexdigest(),
                            "output_sha256": hashlib.sha256(y).hexdigest()})

    with open(os.path.join(HERE, "vectors.json"), "w") as f:
        json.dump({"known_answers": KNOWN_ANSWERS, "vectors": vectors}, f, indent=0, separators=(",", ":"))
        f.write("\n")
    with open(os.path.join(HERE, "digests.json"), "w") as f:
        json.dump(digests, f, indent=0, separators=(",", ":"))
        f.write("\n")
    print(f"{len(vectors)} vectors, {len(digests)} digests")


if __name__ == "__main__":
    main()
