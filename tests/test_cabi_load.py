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


def _strip_c_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


_C_SCALARS = {"uint8_t": "u8", "uint16_t": "u16", "uint32_t": "u32", "uint64_t": "u64", "int32_t": "i32", "int64_t": "i64",
              "size_t": "usize", "bool": "bool", "double": "f64", "float": "f32", "char": "c_char", "void": "c_void"}


def c_type_to_rust(decl: str, named: bool = True) -> str:
    """The Rust spelling a C parameter / field / return declaration must have in an `extern "C"` block: fixed-width integers by
    width, size_t -> usize, `const T *` -> `*const T`, `T *` -> `*mut T`, `T name[N]` -> `[T; N]`, struct names unchanged."""
    decl = " ".join(decl.replace("*", " * ").split())
    arr = re.search(r"\[(\d+)\]$", decl)
    if arr:
        decl = decl[:arr.start()].strip()
    toks = decl.split()
    if named and len(toks) > 1 and toks[-1] != "*":
        toks = toks[:-1]                      # the parameter / field name
    const = "const" in toks
    toks = [t for t in toks if t not in ("const", "struct")]
    stars = toks.count("*")
    base = [t for t in toks if t != "*"]
    assert len(base) == 1, decl
    ty = _C_SCALARS.get(base[0], base[0])
    for k in range(stars):
        ty = ("*const " if const and k == 0 else "*mut ") + ty
    if arr:
        ty = f"[{ty}; {arr.group(1)}]"
    return ty


def rust_type(t: str) -> str:
    return " ".join(t.split()).replace("core::ffi::", "")


def _split_args(args: str):
    return [a.strip() for a in args.split(",") if a.strip() and a.strip() != "void"]


def test_c_type_to_rust_mapping():
    assert c_type_to_rust("const uint8_t *input_ptr") == "*const u8"
    assert c_type_to_rust("uint8_t *output_ptr") == "*mut u8"
    assert c_type_to_rust("size_t len") == "usize"
    assert c_type_to_rust("const DltSizeEstimator *estimator") == "*const DltSizeEstimator"
    assert c_type_to_rust("void *hip_stream") == "*mut c_void"
    assert c_type_to_rust("const void *d_input") == "*const c_void"
    assert c_type_to_rust("bool *out_split") == "*mut bool"
    assert c_type_to_rust("uint8_t reserved[3]") == "[u8; 3]"
    assert c_type_to_rust("const char *", named=False) == "*const c_char"
    assert c_type_to_rust("int32_t", named=False) == "i32"


def test_rust_sys_crate_matches_the_c_prototypes_by_type(pkg):
    """rust/dxt-lossless-transform-gfx950-sys/src/lib.rs has never been through rustc (no toolchain in this image), and a wrong
    width in an `extern "C"` declaration is undefined behaviour that compiles.  So every `pub fn` of its extern block is checked
    against the prototype in include/dxtlt_gfx950.h: exported by the library, the same parameters IN ORDER with the same names and
    the types a fixed C -> Rust mapping gives (const uint8_t * <-> *const u8, size_t <-> usize, bool, uint8_t <-> u8,
    const DltSizeEstimator * <-> *const DltSizeEstimator, ...), and the same return type (int32_t <-> i32) -- the shape of the
    reference's own C entry points (bc1 c_api/transform_with_settings.rs:73,119).  The #[repr(C)] structs are checked field by
    field against the header's typedefs."""
    import ctypes

    src = open(os.path.join(ROOT, "rust", "dxt-lossless-transform-gfx950-sys", "src", "lib.rs")).read()
    src_nc = re.sub(r"//[^\n]*", "", src)
    header = _strip_c_comments(open(os.path.join(ROOT, "include", "dxtlt_gfx950.h")).read())
    est = _strip_c_comments(open(os.path.join(ROOT, "include", "dlt_size_estimator.h")).read())
    lib = ctypes.CDLL(pkg._lib.lib_path())
    block = src_nc[src_nc.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    decls = re.findall(r"pub fn (dxtlt_\w+)\(([^)]*)\)\s*(?:->\s*([^;]+))?;", block)
    assert len(decls) >= 30
    for name, args, ret in decls:
        assert hasattr(lib, name), name
        m = re.search(r"([\w ]+?[\s\*]+)\b" + name + r"\s*\(([^)]*)\)\s*;", header)
        assert m, f"{name}: no prototype in include/dxtlt_gfx950.h"
        c_ret = c_type_to_rust(m.group(1), named=False)
        assert (rust_type(ret) if ret else "c_void") == c_ret, (name, "return", ret, m.group(1))
        c_args, rust_args = _split_args(m.group(2)), _split_args(args)
        assert len(c_args) == len(rust_args), (name, c_args, rust_args)
        for ca, ra in zip(c_args, rust_args):
            rname, rty = [x.strip() for x in ra.split(":", 1)]
            assert rust_type(rty) == c_type_to_rust(ca), (name, ca, ra)
            assert rname == ca.replace("*", " ").split()[-1], (name, "parameter name", ca, ra)

    def c_struct(text, name):
        body = re.search(r"typedef struct " + name + r"\s*\{(.*?)\}\s*" + name + r"\s*;", text, re.S).group(1)
        return [f.strip() for f in body.split(";") if f.strip()]

    def rust_struct(name):
        body = re.search(r"pub struct " + name + r"\s*\{(.*?)\n\}", src_nc, re.S).group(1)
        fields, depth, cur = [], 0, ""
        for ch in body:                       # split at top-level commas (fn pointer types hold commas of their own)
            depth += ch == "("
            depth -= ch == ")"
            if ch == "," and depth == 0:
                fields.append(cur)
                cur = ""
            else:
                cur += ch
        fields.append(cur)
        return [" ".join(f.split()) for f in fields if f.strip()]

    for name in ("DxtltBatchItem", "DxtltShardStat"):
        cf, rf = c_struct(header, name), rust_struct(name)
        assert len(cf) == len(rf), (name, cf, rf)
        for c, r in zip(cf, rf):
            rname, rty = [x.strip() for x in r.replace("pub ", "", 1).split(":", 1)]
            cname = re.sub(r"\[\d+\]$", "", c.replace("*", " ").split()[-1])
            assert rname == cname and rust_type(rty) == c_type_to_rust(c), (name, c, r)
    # DltSizeEstimator: a context pointer and two callbacks whose signatures are the header's function-pointer typedefs
    fn_types = {n: (c_type_to_rust(r, named=False), [c_type_to_rust(a) for a in _split_args(a_)])
                for r, n, a_ in re.findall(r"typedef\s+([\w ]+?)\s*\(\*(\w+)\)\(([^)]*)\);", est)}
    cf, rf = c_struct(est, "DltSizeEstimator"), rust_struct("DltSizeEstimator")
    assert len(cf) == len(rf) == 3
    assert rf[0] == "pub context: *mut c_void" and c_type_to_rust(cf[0]) == "*mut c_void"
    for c, r in zip(cf[1:], rf[1:]):
        want_ret, want_args = fn_types[c.split()[0]]
        m = re.match(r"pub \w+: unsafe extern \"C\" fn\((.*)\) -> (\w+)$", r)
        assert m, r
        got_args = [rust_type(a.split(":", 1)[1]) for a in _split_args(m.group(1))]
        assert got_args == want_args and m.group(2) == want_ret, (c, r)
    # status codes
    for cname, value in re.findall(r"#define (DXTLT_(?:OK|E_\w+))\s+(\d+)", header):
        assert re.search(r"pub const " + cname + r": i32 = " + value + ";", src), cname


def _call_args(text, start):
    """the top-level comma-separated arguments of the call whose '(' is at text[start]"""
    depth, cur, args = 0, "", []
    for ch in text[start:]:
        if ch in "([{":
            depth += 1
            if depth == 1:
                continue
        if ch in ")]}":
            depth -= 1
            if depth == 0:
                break
        if ch == "," and depth == 1:
            args.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        args.append(cur.strip())
    return args


def test_rust_bodies_call_the_sys_crate_with_the_declared_argument_lists():
    """Every `dxtlt_*(...)` call in rust/core-bodies/*.rs (source only) passes as many arguments as the `-sys` crate's `extern "C"`
    declaration of that function takes -- and calls nothing the crate does not declare.  (The declarations themselves are checked
    against the C prototypes by type above; together: bodies -> extern block -> header -> exported symbol.)"""
    sys_src = re.sub(r"//[^\n]*", "", open(os.path.join(ROOT, "rust", "dxt-lossless-transform-gfx950-sys", "src", "lib.rs")).read())
    declared = {n: len(_split_args(a)) for n, a in re.findall(r"pub fn (dxtlt_\w+)\(([^)]*)\)", sys_src)}
    seen = 0
    for path in glob.glob(os.path.join(ROOT, "rust", "core-bodies", "*.rs")):
        src = re.sub(r"//[^\n]*", "", open(path).read())
        for m in re.finditer(r"\b(dxtlt_\w+)\s*\(", src):
            name = m.group(1)
            if src[max(0, m.start() - 3):m.start()] == "fn ":
                continue
            assert name in declared, (os.path.basename(path), name)
            args = _call_args(src, m.end() - 1)
            assert len(args) == declared[name], (os.path.basename(path), name, args, declared[name])
            seen += 1
    assert seen >= 12      # six transforms, three autos, the threshold, the error text, the per-thread estimator cap (x 2) ...


def test_shipped_library_contains_no_experiment_code(pkg):
    """The experiment switches of rounds 1-4 (element-granular kernel, first-form shifted tiles, a wrong-output timing switch, the
    per-workgroup timing arrays) are compiled only with -DDXTLT_EXPERIMENTS / -DDXTLT_WG_TIMING, into side builds under
    build/side-*/ that never replace the shipped library (_build.py).  Checked on the shipped file itself: the tuning mask, the
    kernel names in its device code object, the debug exports."""
    if os.environ.get("DXTLT_LIB_PATH"):
        pytest.skip("another build of the library is under test")
    l = pkg.load()
    assert l.dxtlt_tuning_mask() == 0x22
    blob = open(pkg._lib.lib_path(), "rb").read()
    for needle in (b"generic_kernel", b"fwd_tiled_shift", b"g_wg_marks", b"g_wg_timing", b"dxtlt_debug_read_wg"):
        assert needle not in blob, needle
    assert b"fwd_tiled_halo" in blob and b"inv_tiled_shift" in blob      # (the check can see kernel names)
    from dxt_lossless_transform_amd import _build
    assert _build._dirs([])[1] == _build.LIB_PATH
    assert _build._dirs(["-DDXTLT_EXPERIMENTS"])[1] != _build.LIB_PATH and "side-" in _build._dirs(["-DDXTLT_EXPERIMENTS"])[0]
    # a variable left set in a shell never redirects the package's own build away from the shipped library: only an
    # explicit argument makes a side build, and tools/ab_build_rev.sh is the one place that turns the variable into that argument
    os.environ["DXTLT_EXTRA_HIPCC_FLAGS"] = "-DDXTLT_EXPERIMENTS"
    try:
        assert _build._extra_flags() == [] and _build._extra_flags(None) == [] and _build._extra_flags(["-DX"]) == ["-DX"]
        assert _build.build(force=False) == _build.LIB_PATH
    finally:
        del os.environ["DXTLT_EXTRA_HIPCC_FLAGS"]


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


def test_rust_bodies_default_to_the_device_and_glue_is_no_std():
    """The shipped Rust bodies (source only): with the DEFAULT feature list every call is the FFI call -- no size detour, no
    device-absent detour (both exist only behind opt-in features) -- device failures panic loudly, and the glue
    they share uses nothing from std (the core crates make std optional) and puts no bound beyond the reference's on `vtable`."""
    import os
    import re

    import tomli

    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rust", "core-bodies")
    features = tomli.load(open(os.path.join(root, "Cargo.features.toml"), "rb"))["features"]
    assert features["default"] == ["std"], "the default route must be the tested device path: no cpu feature by default"
    assert features["cpu"] == [] and features["cpu-below-threshold"] == ["cpu"] and features["cpu-without-device"] == ["cpu"]

    def without_optional_code(src):
        """What is left of a body with the default features: an attribute `#[cfg(feature = "cpu…")]` removes the statement, match arm
        or item that follows it (here always: one line, or a braced block that closes at the attribute's indentation)."""
        out, lines, i = [], src.splitlines(), 0
        while i < len(lines):
            m = re.match(r'(\s*)#\[cfg\(feature = "(cpu[a-z-]*)"\)\]\s*$', lines[i])
            if not m:
                out.append(lines[i])
                i += 1
                continue
            assert m.group(2) in ("cpu-below-threshold", "cpu-without-device"), lines[i]
            i += 1
            while lines[i].lstrip().startswith("#["):      # further attributes of the same item
                i += 1
            if lines[i].rstrip().endswith("{"):
                while lines[i] != m.group(1) + "}":
                    i += 1
            i += 1
        return "\n".join(out)

    for n in (1, 2, 3):
        src = open(os.path.join(root, f"bc{n}_transform_with_settings.rs")).read()
        for fn in (f"transform_bc{n}_with_settings", f"untransform_bc{n}_with_settings"):
            body = src[src.index(f"pub unsafe fn {fn}("):]
            body = body[:body.index("\n}\n") + 3]
            # opt-in: the size route stands in front of the FFI call, the device-absent route behind it, each under its own feature
            route, ffi = body.index("if stays_on_cpu(len)"), body.index(f"dxtlt_{fn}(")
            assert route < ffi and f"return {fn}_cpu(" in body[route:ffi], fn
            assert '#[cfg(feature = "cpu-below-threshold")]\n    if stays_on_cpu(len)' in body, fn
            assert '#[cfg(feature = "cpu-without-device")]\n        if device_is_absent(rc)' in body[ffi:], fn
            # default: nothing but the FFI call and the loud failure
            default = without_optional_code(body)
            assert "_cpu(" not in default and "stays_on_cpu" not in default and "device_is_absent" not in default, fn
            assert f"dxtlt_{fn}(" in default and "abort_on_device_failure" in default, fn
        auto = open(os.path.join(root, f"bc{n}_transform_auto.rs")).read()
        assert auto.index("if stays_on_cpu(len)") < auto.index(f"dxtlt_transform_bc{n}_auto(")
        default = without_optional_code(auto)
        assert "_cpu(" not in default.replace(f"transform_bc{n}_auto_cpu`", "") and "stays_on_cpu" not in default and "device_is_absent" not in default, n
        assert f"dxtlt_transform_bc{n}_auto(" in default and "abort_on_device_failure" in default
        # the reference's bounds, nothing added
        assert re.search(r"where\s+T: SizeEstimationOperations,\s*\{", auto), n
    glue = open(os.path.join(root, "gfx950_glue.rs")).read()
    code = "\n".join(l for l in glue.splitlines() if not l.lstrip().startswith("//"))
    assert "std::" not in code and "Mutex" not in code
    assert re.search(r"pub\(crate\) fn vtable<T: SizeEstimationOperations>\(", code)
    assert "dxtlt_host_route_threshold_bytes()" in code
    default_glue = without_optional_code(glue)
    assert "stays_on_cpu" not in default_glue.replace("`stays_on_cpu`", "") and "fn device_is_absent" not in default_glue
    assert 'feature = "cpu"' not in code, "only the two opt-in features gate code; `cpu` is internal (the kept bodies)"
    sys_src = open(os.path.join(os.path.dirname(root), "dxt-lossless-transform-gfx950-sys", "src", "lib.rs")).read()
    assert "#![no_std]" in sys_src


def test_plain_makefile_builds_what_the_python_build_builds():
    """csrc/Makefile (for maintainers who build from build.rs or a shell, without Python) names the same sources and compiler flags as
    _build.py; `make -n` resolves every rule."""
    import subprocess

    from dxt_lossless_transform_amd import _build

    mk = open(os.path.join(_build.CSRC, "Makefile")).read().replace("\\\n", " ")
    sources = re.search(r"^SOURCES\s*:=\s*(.*)$", mk, re.M).group(1).split()
    flags = re.search(r"^FLAGS\s*:=\s*(.*)$", mk, re.M).group(1).split()
    assert sources == _build.SOURCES
    assert flags == _build.BASE_FLAGS
    r = subprocess.run(["make", "-n", "-B", "-C", _build.CSRC, "LIB=/tmp/never_written.so", "OBJ_DIR=/tmp/never_written_obj"],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count(" -x hip -c ") == len(_build.SOURCES) and "-shared -fPIC" in r.stdout


def test_every_rust_source_has_balanced_brackets():
    """No Rust toolchain in this image: the least a source-only file can be held to is that every bracket closes (strings, chars,
    lifetimes and comments skipped).  rust/ (the shim) and tests/golden/reference_kit/ (the pin-on-arrival kit)."""
    import glob

    paths = glob.glob(os.path.join(ROOT, "rust", "**", "*.rs"), recursive=True) + \
        glob.glob(os.path.join(ROOT, "tests", "golden", "reference_kit", "**", "*.rs"), recursive=True)
    assert len(paths) >= 10
    for path in paths:
        src = open(path).read()
        code = re.sub(r'/\*.*?\*/', "", src, flags=re.S)
        code = re.sub(r'//[^\n]*|b?"(?:\\.|[^"\\])*"|b?\'(?:\\.|[^\'\\])\'', "", code)
        stack, pairs = [], {")": "(", "]": "[", "}": "{"}
        for n, ch in enumerate(code):
            if ch in "([{":
                stack.append(ch)
            elif ch in pairs:
                assert stack and stack.pop() == pairs[ch], (os.path.relpath(path, ROOT), code[max(0, n - 60):n + 1])
        assert not stack, os.path.relpath(path, ROOT)
