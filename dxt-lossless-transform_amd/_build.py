"""Builds libdxtlt_gfx950.so in-tree with hipcc (gfx950 only; cross-compiles without a GPU).

One object per source under build/obj (compiled in parallel, rebuilt only when the source or a header is newer), then one
link step.  The shared library is the only artefact that matters; build/ is scratch."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libdxtlt_gfx950.so")
_BUILD_DIR = os.path.join(os.path.dirname(_HERE), "build")

SOURCES = ["bcn_kernels.hip", "batch_kernels.hip", "dxtlt_api.cpp", "c_api_core.cpp", "c_api_stable.cpp", "auto_transform.cpp", "file_format.cpp",
           "bc7_kernels.hip", "bc7_api.cpp", "bc7_sharded.cpp", "bc1_normalize.hip", "normalize_api.cpp", "batch_api.cpp", "bc23_normalize.hip",
           "normalize23_api.cpp", "color565_ops.hip", "color565_api.cpp", "bcn_decode.hip", "decode_api.cpp",
           "auto_kernels.hip", "numa_affinity.cpp"]
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra", "-Wno-unused-command-line-argument"]


def _extra_flags(extra_flags=None) -> list:
    """Extra compiler flags of a SIDE build (A/B experiments: -DDXTLT_EXPERIMENTS, -DDXTLT_WG_TIMING): ONLY the argument.  The
    environment is never read here: build() / __graft_entry__.build() always bring the SHIPPED library up to date, whatever is
    left set in a shell; $DXTLT_EXTRA_HIPCC_FLAGS is honoured by tools/ab_build_rev.sh alone, which passes it in explicitly.  A
    build with extra flags never touches the shipped library or its objects: it goes to build/side-<hash of the flags>/ (objects
    and library), and objects of one flag set are never linked into another."""
    return list(extra_flags or [])


def _dirs(extra: list):
    """(object directory, library path) of a build with the extra flags `extra`"""
    if not extra:
        return os.path.join(_BUILD_DIR, "obj"), LIB_PATH
    tag = hashlib.sha256(" ".join(extra).encode()).hexdigest()[:12]
    side = os.path.join(_BUILD_DIR, "side-" + tag)
    return os.path.join(side, "obj"), os.path.join(side, "libdxtlt_gfx950.so")


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
    """The SHIPPED library against its sources (side builds are always asked for explicitly)."""
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(f) > t for f in _sources() + _headers())


def build(force: bool = False, verbose: bool = False, extra_flags=None) -> str:
    """Compile every HIP/C++ source of the package into one shared library for gfx950 and return its path: the shipped
    library, or -- with extra flags -- a side build under build/side-*/ (see _extra_flags)."""
    extra = _extra_flags(extra_flags)
    obj_dir, lib_path = _dirs(extra)
    if not extra and not force and not is_stale():
        return LIB_PATH
    hipcc = _hipcc()
    os.makedirs(obj_dir, exist_ok=True)
    flags = BASE_FLAGS + extra
    newest_header = max((os.path.getmtime(h) for h in _headers()), default=0.0)

    def obj(src: str) -> str:
        return os.path.join(obj_dir, os.path.basename(src) + ".o")

    todo = []
    for src in _sources():
        o = obj(src)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(src), newest_header):
            todo.append(src)

    def compile_one(src: str) -> None:
        cmd = [hipcc] + flags + ["-x", "hip", "-c", src, "-o", obj(src) + ".tmp"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(obj(src) + ".tmp", obj(src))

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        list(pool.map(compile_one, todo))
    if not todo and os.path.exists(lib_path) and all(os.path.getmtime(obj(s)) <= os.path.getmtime(lib_path) for s in _sources()):
        return lib_path
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [obj(s) for s in _sources()] + ["-o", lib_path + ".tmp", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(lib_path + ".tmp", lib_path)
    return lib_path


if __name__ == "__main__":
    print(build(force=True, verbose=True))
