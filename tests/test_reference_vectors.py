"""The pin: this build's oracle and HIP path against bytes the REFERENCE itself produced.

The reference is Rust and this image has no toolchain, and the reference's own tests hold no transformed-byte vector
(round trips only: bc1 test_prelude.rs:154-317), so nothing in the repository can pin forward bytes until somebody runs

    tests/golden/reference_kit/run.sh <checkout of Sewer56/dxt-lossless-transform>

on a machine with cargo.  That writes tests/golden/reference_out/ (MANIFEST.txt: one `<file> <length> <crc32>` line per output of
transform_bcN_with_settings -- bc1 transform_with_settings.rs:31, bc2 :30, bc3 :32 -- for every all_combinations() setting over the
inputs of tests/golden/make_reference_inputs.py, plus 288 small outputs as whole files).  Until that directory exists the tests
below are dormant: ONE skip whose message is the command.  No output file is ever written by this repository itself.

With the directory present:
  * CPU (-m "not gpu"): the C oracle AND the numpy oracle reproduce every manifest line (length + CRC-32), every whole file byte
    for byte, and the manifest is complete (every case x every settings combination);
  * GPU (-m gpu): the HIP path through the C ABI does the same;
  * and when the kit's second program ran too (AUTO_MANIFEST.txt: what transform_bcN_auto chose, with a CRC estimator, for every
    input and both candidate lists), oracle_auto on the CPU and dltbcNcore_transform_auto on the GPU make the same choices and bytes.
The comparer itself is exercised either way against a throw-away directory under tmp_path."""
import os
import sys
import zlib

import numpy as np
import pytest

from helpers import GOLDEN

sys.path.insert(0, GOLDEN)
import make_reference_inputs as inputs  # noqa: E402

REF_OUT = os.path.join(GOLDEN, "reference_out")
KIT = os.path.join(GOLDEN, "reference_kit")
HOW = ("no reference-produced vectors in this tree (forward-byte parity stays unpinned): on a machine with cargo run "
       "`tests/golden/reference_kit/run.sh <checkout of Sewer56/dxt-lossless-transform>` and commit tests/golden/reference_out/")


def read_manifest(directory):
    rows = []
    with open(os.path.join(directory, "MANIFEST.txt")) as f:
        for line in f:
            if line.strip():
                name, length, crc = line.split()
                rows.append((name, int(length), int(crc, 16)))
    return rows


def parse_output_name(name):
    """`<case>.<fmt>.v<V>[a<A>]c<C>.out` -> (case, fmt, (variant, split_alpha, split_colour))"""
    case, fmt, sid, ext = name.rsplit(".", 3)
    assert ext == "out" and fmt in inputs.FORMATS, name
    ids = {s: key for key, s in inputs.settings_ids(fmt)}
    assert sid in ids, name
    return case, fmt, ids[sid]


def expected_names():
    return {f"{case}.{fmt}.{sid}.out" for fmt in inputs.FORMATS for case in inputs.case_names(fmt) for _, sid in inputs.settings_ids(fmt)}


def compare(directory, produce, what):
    """Every manifest line of `directory` against produce(fmt, input, (variant, split_alpha, split_colour)) -> uint8 array."""
    rows = read_manifest(directory)
    assert {r[0] for r in rows} == expected_names(), "the manifest does not cover every case x settings combination"
    whole = 0
    for name, length, crc in rows:
        case, fmt, s = parse_output_name(name)
        got = np.ascontiguousarray(produce(fmt, inputs.case(case, fmt), s))
        assert got.size == length, (what, name, "length")
        assert zlib.crc32(got.tobytes()) == crc, (what, name, "crc32")
        path = os.path.join(directory, name)
        if os.path.exists(path):
            want = np.fromfile(path, dtype=np.uint8)
            assert np.array_equal(got, want), (what, name, "first differing byte", int(np.argmax(got != want[: got.size])) if want.size == got.size else "length")
            whole += 1
    return len(rows), whole


def read_auto_manifest(directory):
    """AUTO_MANIFEST.txt: `<case>.<fmt>.all<0|1> <settings id> <length> <crc32>` -> (case, fmt, use_all, (variant, split_alpha, split_colour), length, crc)"""
    rows = []
    with open(os.path.join(directory, "AUTO_MANIFEST.txt")) as f:
        for line in f:
            if line.strip():
                name, sid, length, crc = line.split()
                case, fmt, mode = name.rsplit(".", 2)
                ids = {s: key for key, s in inputs.settings_ids(fmt)}
                assert mode in ("all0", "all1") and sid in ids, line
                rows.append((case, fmt, mode == "all1", ids[sid], int(length), int(crc, 16)))
    return rows


def crc_estimate(section) -> int:
    """The kit's estimator (src/bin/auto.rs): CRC-32 of the section & 0xFFFFF -- every byte shown to the estimator matters"""
    return zlib.crc32(bytes(section)) & 0xFFFFF


def compare_auto(directory, produce, what):
    """Every AUTO_MANIFEST line against produce(fmt, input, use_all) -> ((variant, split_alpha, split_colour), output array)."""
    rows = read_auto_manifest(directory)
    want_names = {(case, fmt, use_all) for fmt in inputs.FORMATS for case in inputs.case_names(fmt) for use_all in (False, True)}
    assert {r[:3] for r in rows} == want_names, "the auto manifest does not cover every case x use_all_decorrelation_modes"
    for case, fmt, use_all, choice, length, crc in rows:
        got_choice, out = produce(fmt, inputs.case(case, fmt), use_all)
        assert tuple(int(v) for v in got_choice) == choice, (what, case, fmt, use_all, "choice", got_choice, choice)
        out = np.ascontiguousarray(out)
        assert out.size == length and zlib.crc32(out.tobytes()) == crc, (what, case, fmt, use_all, "output")
    return len(rows)


def oracle_auto_produce():
    from oracle import oracle_auto

    def produce(fmt, x, use_all):
        choice, out, _calls = oracle_auto.transform_auto(fmt, x, crc_estimate, use_all)
        return choice, out
    return produce


def oracle_c_produce(oracle):
    return lambda fmt, x, s: oracle.transform(fmt, x, s[0], bool(s[2]), bool(s[1]))


def oracle_np_produce():
    from oracle import oracle_np

    return lambda fmt, x, s: oracle_np.transform(fmt, x, s[0], bool(s[2]), bool(s[1]))


HAS_AUTO_MANIFEST = os.path.exists(os.path.join(REF_OUT, "AUTO_MANIFEST.txt"))
needs_reference_out = pytest.mark.skipif(not os.path.exists(os.path.join(REF_OUT, "MANIFEST.txt")), reason=HOW)


# ------------------------------------------------------------------------------------------------------------
# dormant until tests/golden/reference_out/ arrives
# ------------------------------------------------------------------------------------------------------------
@needs_reference_out
def test_oracles_equal_the_reference_bytes(oracle):
    n, whole = compare(REF_OUT, oracle_c_produce(oracle), "C oracle")
    assert whole >= 1, "commit at least the small whole-file outputs (make_reference_inputs.py --commit-list)"
    compare(REF_OUT, oracle_np_produce(), "numpy oracle")
    if HAS_AUTO_MANIFEST:      # the kit's second program (src/bin/auto.rs): the reference's transform_bcN_auto choices and bytes
        assert compare_auto(REF_OUT, oracle_auto_produce(), "oracle_auto") > 0


@pytest.mark.gpu
@needs_reference_out
def test_hip_path_equals_the_reference_bytes(pkg):
    torch = pytest.importorskip("torch")
    from helpers import pkg_settings

    assert torch.cuda.is_available(), "gpu tests need a GPU"

    def produce(fmt, x, s):
        xd = torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0")
        yd = torch.full_like(xd, 0xA5)
        getattr(pkg, f"transform_{fmt}_with_settings")(xd, yd, pkg_settings(pkg, fmt, s))
        torch.cuda.synchronize()
        back = torch.full_like(xd, 0x5A)
        getattr(pkg, f"untransform_{fmt}_with_settings")(yd, back, pkg_settings(pkg, fmt, s))
        torch.cuda.synchronize()
        assert torch.equal(back, xd), (fmt, s, "inverse")
        return yd.cpu().numpy()

    compare(REF_OUT, produce, "HIP")
    if HAS_AUTO_MANIFEST:
        # dltbcNcore_transform_auto (host pointers, the estimator as a C callback) against the reference's own choices and output bytes
        import ctypes as C

        import cabi

        lib = cabi.bind(C.CDLL(pkg._lib.lib_path()))
        core_settings = {"bc1": cabi.CoreSettings2, "bc2": cabi.CoreSettings2, "bc3": cabi.CoreSettings3}

        def produce_auto(fmt, x, use_all):
            est, _ = cabi.make_estimator("crc")
            x = np.ascontiguousarray(x)
            y = np.zeros_like(x)
            out = core_settings[fmt]()
            r = getattr(lib, f"dltbc{fmt[2]}core_transform_auto")(x.ctypes.data, x.size, y.ctypes.data, y.size, C.byref(est),
                                                                 cabi.AutoSettings(use_all), C.byref(out))
            assert r.ErrorCode == 0, (fmt, use_all, r.ErrorCode)
            return (out.DecorrelationMode, int(out.SplitAlphaEndpoints) if fmt == "bc3" else 0, int(out.SplitColourEndpoints)), y

        assert compare_auto(REF_OUT, produce_auto, "C API auto") > 0


# ------------------------------------------------------------------------------------------------------------
# always on: the kit is whole, its inputs are what they claim to be, and the comparer catches a wrong byte
# ------------------------------------------------------------------------------------------------------------
def test_kit_is_complete_and_stays_off_the_gpu_box():
    for f in ("run.sh", "Cargo.toml.in", "src/main.rs", "src/bin/auto.rs"):
        assert os.path.exists(os.path.join(KIT, f)), f
    src = open(os.path.join(KIT, "src", "main.rs")).read()
    for fmt in inputs.FORMATS:       # exactly the hot path's entry points, every settings combination, std only
        assert f"transform_{fmt}_with_settings(input.as_ptr(), out.as_mut_ptr(), len, s)" in src
        assert f"Bc{fmt[2]}TransformSettings::all_combinations()" in src
    assert "extern crate" not in src and all(line.split()[1].startswith(("std", "dxt_lossless_transform_bc")) for line in src.splitlines() if line.startswith("use "))
    # no compiler here: at least every bracket of the source closes (strings, chars and comments skipped)
    import re

    code = re.sub(r'//[^\n]*|"(?:\\.|[^"\\])*"|\'(?:\\.|[^\'\\])\'', "", src)
    stack, pairs = [], {")": "(", "]": "[", "}": "{"}
    for ch in code:
        if ch in "([{":
            stack.append(ch)
        elif ch in pairs:
            assert stack and stack.pop() == pairs[ch], "unbalanced bracket in the kit's main.rs"
    assert not stack
    manifest = open(os.path.join(KIT, "Cargo.toml.in")).read()
    deps = manifest.split("[dependencies]")[1].split("[")[0]
    assert [line.split()[0] for line in deps.strip().splitlines() if not line.startswith("#")] == \
        [f"dxt-lossless-transform-{fmt}" for fmt in inputs.FORMATS] + ["dxt-lossless-transform-api-common"]
    auto = open(os.path.join(KIT, "src", "bin", "auto.rs")).read()
    for fmt in inputs.FORMATS:
        assert f"transform_{fmt}_auto(input.as_ptr(), out.as_mut_ptr(), len, &options)" in auto
    assert "Ok((crc32(section) & 0x000F_FFFF) as usize)" in auto and "for use_all in [false, true]" in auto
    assert all(line.split()[1].startswith(("std", "dxt_lossless_transform_")) for line in auto.splitlines() if line.startswith("use "))
    root = os.path.dirname(os.path.dirname(GOLDEN))
    assert "tests/golden/reference_kit/" in open(os.path.join(root, ".gpurunignore")).read().split()
    if not os.path.exists(REF_OUT):  # this repository never writes reference outputs itself
        assert not any(f.endswith(".out") for f in os.listdir(GOLDEN))


def test_kit_crc_is_zlib_crc32():
    """The kit's bitwise CRC-32 (main.rs: reflected 0xEDB88320, initial and final complement), restated; the manifest is checked with zlib."""
    def crc32_bitwise(data):
        crc = 0xFFFFFFFF
        for b in data:
            crc ^= b
            for _ in range(8):
                crc = (crc >> 1) ^ (0xEDB88320 & (-(crc & 1) & 0xFFFFFFFF))
        return crc ^ 0xFFFFFFFF

    src = open(os.path.join(KIT, "src", "main.rs")).read()
    assert "let mut crc = !0u32;" in src and "(crc >> 1) ^ (0xEDB8_8320 & (crc & 1).wrapping_neg())" in src and "    !crc\n" in src
    for data in (b"", b"123456789", bytes(range(256)) * 3):
        assert crc32_bitwise(data) == zlib.crc32(data)
    assert zlib.crc32(b"123456789") == 0xCBF43926


def test_inputs_are_the_reference_generators_and_cover_every_tail(oracle):
    names = {fmt: list(inputs.case_names(fmt)) for fmt in inputs.FORMATS}
    assert len(expected_names()) == sum(len(v) for v in (names["bc1"], names["bc2"])) * 8 + len(names["bc3"]) * 16
    for fmt in inputs.FORMATS:
        for n in range(1, 66):
            assert np.array_equal(inputs.case(f"gen-n{n:03d}", fmt), oracle.generate_test_data(fmt, n))
            assert inputs.case(f"splitmix-n{n:03d}", fmt).size == n * inputs.BLOCK[fmt]
        assert inputs.case("splitmix-n100003", fmt).size == 100003 * inputs.BLOCK[fmt]
        assert np.array_equal(inputs.seeded(fmt, 9, 77), oracle.fill_splitmix64(9 * inputs.BLOCK[fmt], 77))
    c = inputs.case("colours", "bc1").view("<u2").reshape(-1, 4)
    assert np.array_equal(c[:, 0], np.arange(65536, dtype=np.uint16)) and len(set(c[:, 1].tolist())) == 65536
    # the three known answers the reference's tests hold for its generators (test_prelude.rs: bc1 :107-119, bc2 :586-606, bc3 :1058-1078)
    from helpers import golden_vectors

    for fmt in inputs.FORMATS:
        assert inputs.case("gen-n003", fmt).tobytes().hex() == golden_vectors()["known_answers"][f"{fmt}_generator_3"]


def test_comparer_on_a_throw_away_directory(tmp_path, oracle):
    """The comparer's own check: a manifest written from the ORACLE under tmp_path (never in the tree, never a reference claim) passes;
    one flipped byte in a whole file, one wrong CRC and one missing line each fail."""
    d = str(tmp_path)
    produce = oracle_c_produce(oracle)
    lines = []
    for name in sorted(expected_names()):
        case, fmt, s = parse_output_name(name)
        y = produce(fmt, inputs.case(case, fmt), s)
        lines.append(f"{name} {y.size} {zlib.crc32(y.tobytes()):08x}")
        if case in inputs.COMMIT_WHOLE:
            y.tofile(os.path.join(d, name))
    manifest = os.path.join(d, "MANIFEST.txt")
    open(manifest, "w").write("\n".join(lines) + "\n")
    n, whole = compare(d, oracle_np_produce(), "numpy oracle")
    assert (n, whole) == (len(lines), 288)

    victim = os.path.join(d, "gen-n017.bc3.v1a1c1.out")
    good = open(victim, "rb").read()
    open(victim, "wb").write(good[:40] + bytes([good[40] ^ 1]) + good[41:])
    with pytest.raises(AssertionError, match="first differing byte"):
        compare(d, produce, "C oracle")
    open(victim, "wb").write(good)

    bad = [ln if not ln.startswith("colours.bc1.v2c0.out") else ln[:-8] + f"{(int(ln[-8:], 16) ^ 0x10):08x}" for ln in lines]
    open(manifest, "w").write("\n".join(bad) + "\n")
    with pytest.raises(AssertionError, match="crc32"):
        compare(d, produce, "C oracle")
    open(manifest, "w").write("\n".join(lines[1:]) + "\n")
    with pytest.raises(AssertionError, match="does not cover"):
        compare(d, produce, "C oracle")


def test_auto_comparer_on_a_throw_away_directory(tmp_path):
    """compare_auto's own check, like the one above: an AUTO_MANIFEST written from oracle_auto under tmp_path passes; a changed choice,
    a changed output CRC and a missing line each fail.  Also: the estimator makes real choices (more than one setting wins somewhere)."""
    d = str(tmp_path)
    produce = oracle_auto_produce()
    sid = {fmt: {key: s for key, s in inputs.settings_ids(fmt)} for fmt in inputs.FORMATS}
    lines, winners = [], set()
    for fmt in inputs.FORMATS:
        for case in inputs.case_names(fmt):
            for use_all in (False, True):
                choice, out = produce(fmt, inputs.case(case, fmt), use_all)
                winners.add((fmt, choice))
                lines.append(f"{case}.{fmt}.all{int(use_all)} {sid[fmt][tuple(choice)]} {out.size} {zlib.crc32(out.tobytes()):08x}")
    assert len(winners) >= 20          # 8 + 8 + 16 settings exist; the CRC estimator spreads its choices over them
    path = os.path.join(d, "AUTO_MANIFEST.txt")
    open(path, "w").write("\n".join(lines) + "\n")
    assert compare_auto(d, produce, "oracle_auto") == len(lines) == 2 * 397

    k = next(i for i, ln in enumerate(lines) if ln.startswith("r2-256.bc3.all1 "))
    name, s, length, crc = lines[k].split()
    other = "v0a0c0" if s != "v0a0c0" else "v1a1c1"
    open(path, "w").write("\n".join(lines[:k] + [f"{name} {other} {length} {crc}"] + lines[k + 1:]) + "\n")
    with pytest.raises(AssertionError, match="choice"):
        compare_auto(d, produce, "oracle_auto")
    open(path, "w").write("\n".join(lines[:k] + [f"{name} {s} {length} {(int(crc, 16) ^ 1):08x}"] + lines[k + 1:]) + "\n")
    with pytest.raises(AssertionError, match="output"):
        compare_auto(d, produce, "oracle_auto")
    open(path, "w").write("\n".join(lines[:-1]) + "\n")
    with pytest.raises(AssertionError, match="does not cover"):
        compare_auto(d, produce, "oracle_auto")
