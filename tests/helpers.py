"""Shared helpers for the test-suite (settings enumeration, golden access)."""
from __future__ import annotations

import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BLOCK = {"bc1": 8, "bc2": 16, "bc3": 16}
FORMATS = ("bc1", "bc2", "bc3")


def all_settings(fmt: str):
    """(variant, split_alpha, split_colour) -- 8 combos for BC1/BC2, 16 for BC3
    (bc1 settings.rs:68, bc3 settings.rs:74)."""
    for v in range(4):
        for sc in (0, 1):
            for sa in ((0, 1) if fmt == "bc3" else (0,)):
                yield v, sa, sc


def settings_id(s) -> str:
    return f"v{s[0]}-sa{s[1]}-sc{s[2]}"


_vec = None


def golden_vectors():
    global _vec
    if _vec is None:
        with open(os.path.join(GOLDEN, "vectors.json")) as f:
            _vec = json.load(f)
    return _vec


def golden_digests():
    with open(os.path.join(GOLDEN, "digests.json")) as f:
        return json.load(f)


def payload(fmt: str) -> np.ndarray:
    return np.fromfile(os.path.join(GOLDEN, f"r2-256-{fmt}.payload.bin"), dtype=np.uint8)


def pkg_settings(pkg, fmt: str, s):
    v, sa, sc = s
    if fmt == "bc1":
        return pkg.Bc1TransformSettings(pkg.YCoCgVariant(v), bool(sc))
    if fmt == "bc2":
        return pkg.Bc2TransformSettings(pkg.YCoCgVariant(v), bool(sc))
    return pkg.Bc3TransformSettings(pkg.YCoCgVariant(v), bool(sa), bool(sc))
