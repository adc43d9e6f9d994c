"""Builds libdxtlt_gfx950.so in-tree with hipcc (gfx950 only; cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libdxtlt_gfx950.so")

SOURCES = ["bcn_kernels.hip", "dxtlt_api.cpp", "c_api_core.cpp", "c_api_stable.cpp", "auto_transform.cpp", "file_format.cpp",
           "bc7_kernels.hip", "bc7_api.cpp", "bc7_sharded.cpp", "bc1_normalize.hip", "normalize_api.cpp", "batch_api.cpp", "bc23_normalize.hip",
           "normalize23_api.cpp", "color565_ops.hip", "color565_api.cpp", "bcn_decode.hip", "decode_api.cpp"]
HEADERS = ["bcn_launch.h", "ycocg_swar.h", "host_common.h", "bc7_launch.h", "bc1_normalize.h", "bc23_normalize.h", "bcn_decode.h", "launch_grid.h", "streaming_store.h"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libdxtlt_gfx950.so cannot be built (set HIPCC)")


def _inputs():
    inc = os.path.join(os.path.dirname(_HERE), "include")
    files = [os.path.join(CSRC, s) for s in SOURCES + HEADERS if os.path.exists(os.path.join(CSRC, s))]
    if os.path.isdir(inc):
        files += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    return files


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(f) > t for f in _inputs())


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP/C++ source of the package into one shared library for gfx950."""
    if not force and not is_stale():
        return LIB_PATH
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [
        _hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
        "-Wall", "-Wextra", "-Wno-unused-command-line-argument",
        "-x", "hip",
    ] + srcs + ["-o", LIB_PATH + ".tmp", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
