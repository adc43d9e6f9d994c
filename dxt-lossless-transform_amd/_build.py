"""Builds libdxtlt_gfx950.so in-tree with hipcc (gfx950 only; cross-compiles without a GPU).

One object per source under build/obj (compiled in parallel, rebuilt only when the source or a header is newer), then one
link step.  The shared library is the only artefact that matters; build/ is scratch."""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libdxtlt_gfx950.so")
OBJ_DIR = os.path.join(os.path.dirname(_HERE), "build", "obj")

SOURCES = ["bcn_kernels.hip", "batch_kernels.hip", "dxtlt_api.cpp", "c_api_core.cpp", "c_api_stable.cpp", "auto_transform.cpp", "file_format.cpp",
           "bc7_kernels.hip", "bc7_api.cpp", "bc7_sharded.cpp", "bc1_normalize.hip", "normalize_api.cpp", "batch_api.cpp", "bc23_normalize.hip",
           "normalize23_api.cpp", "color565_ops.hip", "color565_api.cpp", "bcn_decode.hip", "decode_api.cpp",
           "auto_kernels.hip", "numa_affinity.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-command-line-argument"]
# A/B experiments (tools/ab_build_rev.sh): extra compiler flags for a side build, never set for the shipped library
FLAGS += os.environ.get("DXTLT_EXTRA_HIPCC_FLAGS", "").split()


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libdxtlt_gfx950.so cannot be built (set HIPCC)")


def _sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _headers():
    inc = os.path.join(os.path.dirname(_HERE), "include")
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if os.path.isdir(inc):
        files += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    return files


def is_stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(f) > t for f in _sources() + _headers())


def _obj(src: str) -> str:
    return os.path.join(OBJ_DIR, os.path.basename(src) + ".o")


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP/C++ source of the package into one shared library for gfx950."""
    if not force and not is_stale():
        return LIB_PATH
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    newest_header = max((os.path.getmtime(h) for h in _headers()), default=0.0)
    todo = []
    for src in _sources():
        o = _obj(src)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(src), newest_header):
            todo.append(src)

    def compile_one(src: str) -> None:
        cmd = [hipcc] + FLAGS + ["-x", "hip", "-c", src, "-o", _obj(src) + ".tmp"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(_obj(src) + ".tmp", _obj(src))

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        list(pool.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [_obj(s) for s in _sources()] + ["-o", LIB_PATH + ".tmp", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
