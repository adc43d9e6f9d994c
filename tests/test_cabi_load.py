"""CPU-side checks of the C-ABI library: it loads without a GPU, exports every symbol the headers declare, and
its argument validation (which runs before any device work) behaves like the reference's."""
import ctypes as C
import glob
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text = open(h).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        text = re.sub(r"//[^\n]*", "", text)
        for m in re.finditer(r"\b((?:dxtlt|dltbc\d(?:core)?)_\w+|is_dds|parse_dds)\s*\(", text):
            names.add(m.group(1))
    return sorted(names)


def test_headers_declare_something():
    syms = declared_symbols()
    assert "dxtlt_transform_bc1_with_settings" in syms
    assert len(syms) >= 19


def test_library_exports_every_declared_symbol(pkg):
    lib = C.CDLL(pkg._lib.lib_path())
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in include/*.h but not exported: {missing}"


def test_version_and_last_error(pkg):
    l = pkg.load()
    assert l.dxtlt_version().decode().startswith("dxtlt-gfx950")
    assert isinstance(pkg._lib.last_error(), str)


def test_validation_precedes_device_work(pkg):
    """Length / argument errors are reported without touching a device (so they work on a CPU-only box)."""
    l = pkg.load()
    buf = np.zeros(64, dtype=np.uint8)
    out = np.zeros(64, dtype=np.uint8)
    assert l.dxtlt_transform_bc1_with_settings(buf.ctypes.data, out.ctypes.data, 12, 1, True) == 1  # INVALID_LENGTH
    assert l.dxtlt_transform_bc3_with_settings(buf.ctypes.data, out.ctypes.data, 24, 1, True, True) == 1
    assert l.dxtlt_transform_bc1_with_settings(buf.ctypes.data, out.ctypes.data, 16, 9, True) == 2  # bad mode
    assert l.dxtlt_transform_bc1_with_settings(None, out.ctypes.data, 16, 1, True) == 2  # NULL
    assert "dxtlt" in pkg._lib.last_error()
    # zero blocks is a no-op and needs no device
    assert l.dxtlt_transform_bc2_with_settings(buf.ctypes.data, out.ctypes.data, 0, 1, True) == 0
    assert l.dxtlt_transform_range_device(1, False, None, None, 10, 8, 4, 1, False, True, None) == 2  # bad range


def test_python_wrapper_validation(pkg):
    x = np.zeros(24, dtype=np.uint8)
    with pytest.raises(pkg.InvalidLength):
        pkg.transform_bc1_with_settings(x[:12], np.zeros(12, dtype=np.uint8))
    with pytest.raises(pkg.OutputBufferTooSmall) as e:
        pkg.transform_bc1_with_settings(x, np.zeros(16, dtype=np.uint8))
    assert (e.value.needed, e.value.actual) == (24, 16)
    # length is checked before size (bc1 safe/transform_with_settings.rs:93-105)
    with pytest.raises(pkg.InvalidLength):
        pkg.transform_bc3_with_settings(x, np.zeros(1, dtype=np.uint8))


def test_no_cpu_fallback_without_device(pkg):
    """On a box without a GPU a real transform must fail loudly, never silently compute on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    x = np.zeros(16, dtype=np.uint8)
    with pytest.raises(pkg.DeviceError) as e:
        pkg.transform_bc1_with_settings(x, np.zeros(16, dtype=np.uint8))
    assert e.value.code in (3, 4)


def test_settings_defaults_and_combinations(pkg):
    # bc1 settings.rs:35-43, bc3 settings.rs:39-48
    assert pkg.Bc1TransformSettings() == pkg.Bc1TransformSettings(pkg.YCoCgVariant.Variant1, True)
    assert pkg.Bc3TransformSettings() == pkg.Bc3TransformSettings(pkg.YCoCgVariant.Variant1, True, True)
    assert len(set(pkg.Bc1TransformSettings.all_combinations())) == 8
    assert len(set(pkg.Bc2TransformSettings.all_combinations())) == 8
    assert len(set(pkg.Bc3TransformSettings.all_combinations())) == 16
    assert [int(v) for v in pkg.YCoCgVariant] == [0, 1, 2, 3]


def test_stream_table_and_shard_plan(pkg):
    assert pkg.stream_table("bc1", pkg.Bc1TransformSettings()) == [(0, 2), (2, 2), (4, 4)]
    assert pkg.stream_table("bc1", pkg.Bc1TransformSettings(pkg.YCoCgVariant.NONE, False)) == [(0, 4), (4, 4)]
    assert pkg.stream_table("bc2", pkg.Bc2TransformSettings()) == [(0, 8), (8, 2), (10, 2), (12, 4)]
    assert pkg.stream_table("bc3", pkg.Bc3TransformSettings()) == [(0, 1), (1, 1), (2, 6), (8, 2), (10, 2), (12, 4)]
    assert pkg.stream_table("bc3", pkg.Bc3TransformSettings(pkg.YCoCgVariant.NONE, False, False)) == \
        [(0, 2), (2, 6), (8, 4), (12, 4)]
    plan = pkg.plan_shards(10_000_019, 8)
    assert plan[0][0] == 0 and sum(n for _, n in plan) == 10_000_019
    assert all(a + n == b for (a, n), (b, _) in zip(plan, plan[1:]))
    assert all(first % 2048 == 0 for first, _ in plan)
    assert pkg.plan_shards(5, 8)[-1] == (0, 5)


@pytest.mark.parametrize("header", sorted(os.path.basename(h) for h in glob.glob(os.path.join(ROOT, "include", "*.h"))))
def test_public_headers_are_valid_c_and_cxx(header):
    """Every public C header compiles on its own as C11 and as C++17 (the cbindgen `style = "both"` contract)."""
    import subprocess

    path = os.path.join(ROOT, "include", header)
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-fsyntax-only", "-x", "c", path])
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-fsyntax-only", "-x", "c++", path])


def test_new_entry_points_validate_without_a_device(pkg):
    """Normalisation, batch and BC7 sharding entry points: argument errors and empty inputs need no device."""
    import ctypes

    l = pkg.load()
    buf = np.zeros(64, dtype=np.uint8)
    p = buf.ctypes.data
    assert l.dxtlt_bc1_normalize_blocks(p, p, 12, 1) == 1
    assert l.dxtlt_bc1_normalize_blocks(p, p, 16, 7) == 2
    assert l.dxtlt_bc1_normalize_blocks(p, p, 0, 1) == 0
    assert l.dxtlt_bc1_normalize_blocks(p, p, 16, 0) == 0          # mode None in place: nothing to do
    assert l.dxtlt_bc1_normalize_split_blocks_in_place(p, p, 0, 1) == 0
    assert l.dxtlt_transform_bc1_with_normalize_blocks(p, p, None, 12, 1, 1, True) == 1
    assert l.dxtlt_transform_bc1_with_normalize_blocks(p, p, None, 16, 3, 1, True) == 2
    assert l.dxtlt_transform_batch_device(None, 0, None) == 0
    assert l.dxtlt_transform_batch_device(None, 3, None) == 2
    l.dxtlt_transform_bc7_sharded.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int32]
    assert l.dxtlt_transform_bc7_sharded(p, p, 24, 2) == 1
    assert l.dxtlt_transform_bc7_sharded(p, p, 0, 2) == 0


def test_rust_sys_crate_declares_only_exported_symbols(pkg):
    """rust/dxt-lossless-transform-gfx950-sys/src/lib.rs (source only: no Rust toolchain here) must not drift from the
    library: every `pub fn` of its extern block is an exported symbol, and every argument list has the C prototype's
    number of parameters."""
    import ctypes
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "rust", "dxt-lossless-transform-gfx950-sys", "src", "lib.rs")).read()
    header = open(os.path.join(root, "include", "dxtlt_gfx950.h")).read()
    lib = ctypes.CDLL(pkg._lib.lib_path())
    decls = re.findall(r"pub fn (dxtlt_\w+)\(([^)]*)\)", src)
    assert len(decls) >= 20
    for name, args in decls:
        assert hasattr(lib, name), name
        m = re.search(r"\b" + name + r"\(([^)]*)\)", header)
        assert m, name
        c_args = [a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]
        rust_args = [a for a in args.split(",") if a.strip()]
        assert len(c_args) == len(rust_args), (name, len(c_args), len(rust_args))


def test_host_route_threshold_default_setter_and_environment(pkg):
    """dxtlt_host_route_threshold_bytes: the host-pointer crossover a size-routing caller uses (rust/core-bodies).  Default
    32 MiB, a setter, and $DXTLT_HOST_ROUTE_THRESHOLD_BYTES on top of both; no device needed."""
    import ctypes
    import os
    import subprocess
    import sys

    lib = ctypes.CDLL(pkg._lib.lib_path())
    lib.dxtlt_host_route_threshold_bytes.restype = ctypes.c_size_t
    lib.dxtlt_set_host_route_threshold_bytes.argtypes = [ctypes.c_size_t]
    if "DXTLT_HOST_ROUTE_THRESHOLD_BYTES" not in os.environ:
        assert lib.dxtlt_host_route_threshold_bytes() == 32 << 20
        lib.dxtlt_set_host_route_threshold_bytes(1 << 20)
        assert lib.dxtlt_host_route_threshold_bytes() == 1 << 20
        lib.dxtlt_set_host_route_threshold_bytes(32 << 20)
    code = ("import ctypes; l = ctypes.CDLL(%r); l.dxtlt_host_route_threshold_bytes.restype = ctypes.c_size_t; "
            "print(l.dxtlt_host_route_threshold_bytes())" % pkg._lib.lib_path())
    for value, want in (("0", 0), ("65536", 65536), ("not a number", 32 << 20)):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120,
                           env=dict(os.environ, DXTLT_HOST_ROUTE_THRESHOLD_BYTES=value))
        assert r.returncode == 0 and int(r.stdout.strip()) == want, (value, r.stdout, r.stderr)


def test_rust_bodies_route_by_size_and_glue_is_no_std():
    """The shipped Rust bodies (source only) send small inputs to the crate's own CPU dispatch before the FFI call, keep the
    loud panic for device failures, and the glue they share uses nothing from std (the core crates make std optional) and
    puts no bound beyond the reference's on `vtable`."""
    import os
    import re

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rust", "core-bodies")
    for n in (1, 2, 3):
        src = open(os.path.join(root, f"bc{n}_transform_with_settings.rs")).read()
        for fn in (f"transform_bc{n}_with_settings", f"untransform_bc{n}_with_settings"):
            body = src[src.index(f"pub unsafe fn {fn}("):]
            body = body[:body.index("\n}\n") + 3]
            route, ffi = body.index("if stays_on_cpu(len)"), body.index(f"dxtlt_{fn}(")
            assert route < ffi and f"return {fn}_cpu(" in body[route:ffi], fn
            assert "abort_on_device_failure" in body[ffi:] and "device_is_absent(rc)" in body[ffi:], fn
        auto = open(os.path.join(root, f"bc{n}_transform_auto.rs")).read()
        assert auto.index("if stays_on_cpu(len)") < auto.index(f"dxtlt_transform_bc{n}_auto(")
        # the reference's bounds, nothing added
        assert re.search(r"where\s+T: SizeEstimationOperations,\s*\{", auto), n
    glue = open(os.path.join(root, "gfx950_glue.rs")).read()
    code = "\n".join(l for l in glue.splitlines() if not l.lstrip().startswith("//"))
    assert "std::" not in code and "Mutex" not in code
    assert re.search(r"pub\(crate\) fn vtable<T: SizeEstimationOperations>\(", code)
    assert "dxtlt_host_route_threshold_bytes()" in code
    sys_src = open(os.path.join(os.path.dirname(root), "dxt-lossless-transform-gfx950-sys", "src", "lib.rs")).read()
    assert "#![no_std]" in sys_src
