"""Two properties of the gfx950 code that cost 0.03-0.08 of peak when the compiler drops them -- and it did, silently, when the
code AROUND the hot path changed (profiles/r04_batch_edge_tiles.txt).  Checked on the device assembly hipcc emits (cross-compiled
here, no GPU needed), for the default-settings kernels of the batch path:

  * wave 0 of an inverse shifted tile issues its main load and its tail load back to back: no `s_waitcnt vmcnt` between them
    (with one, every workgroup waits out a memory round trip before asking for its tail segments: BC3 inverse 0.77 -> 0.70);
  * the batch kernel's table lookup fetches every field of its entry in ONE group of scalar loads (no scalar load between the
    entry's first wait and the tile's first vector load except inside the rare walk loop)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dxt-lossless-transform_amd", "csrc")


def _device_asm(name):
    src = os.path.join(CSRC, name + ".hip")
    out = os.path.join(ROOT, "build", "isa", name + ".s")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        hipcc = os.environ.get("HIPCC") or "/opt/rocm/bin/hipcc"
        if not os.path.exists(hipcc):
            pytest.skip("no hipcc")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "--cuda-device-only", "-S", src,
                               "-o", out + ".tmp", "-Wno-unused-command-line-argument"], cwd="/tmp")
        os.replace(out + ".tmp", out)
    return open(out).read()


@pytest.fixture(scope="module")
def batch_asm():
    return _device_asm("batch_kernels")


@pytest.fixture(scope="module")
def single_asm():
    return _device_asm("bcn_kernels")


def kernel_body(asm: str, mangled_fragment: str, prefix: str = "_ZN5dxtlt12batch_kernel") -> list:
    m = re.search(r"^(" + prefix + re.escape(mangled_fragment) + r"\w*):\s*(?:;.*)?$", asm, re.M)
    assert m, mangled_fragment
    body = asm[m.end():]
    body = body[:body.index(".Lfunc_end")]
    return [l.strip() for l in body.splitlines() if l.strip() and not l.strip().startswith(";")]


@pytest.mark.parametrize("which, kernel", [("batch", "ILi3ELi1ELb1ELb1ELb1E"), ("batch", "ILi1ELi1ELb0ELb1ELb1E"),   # BC3 / BC1 default settings, inverse
                                           ("single", "ILi3ELi1ELb1ELb1E"), ("single", "ILi1ELi1ELb0ELb1E"), ("single", "ILi2ELi1ELb0ELb1E")])
def test_wave0_of_the_inverse_shifted_tile_issues_both_loads_before_it_waits(batch_asm, single_asm, which, kernel):
    lines = kernel_body(batch_asm, kernel) if which == "batch" else kernel_body(single_asm, kernel, "_ZN5dxtlt15inv_tiled_shift")
    nt_loads = [i for i, l in enumerate(lines) if l.startswith("global_load_dwordx4") and l.endswith(" nt")]
    assert len(nt_loads) >= 2
    back_to_back = 0
    for a, b in zip(nt_loads, nt_loads[1:]):
        between = lines[a + 1:b]
        if not any(l.startswith("s_waitcnt vmcnt") or l.startswith("s_barrier") or l.startswith("s_endpgm") for l in between):
            back_to_back += 1
    assert back_to_back >= 1, "no two nt loads without a vmcnt wait between them: wave 0's main and tail loads are serialised"


@pytest.mark.parametrize("kernel", ["ILi3ELi1ELb1ELb1ELb0E", "ILi3ELi1ELb1ELb1ELb1E", "ILi1ELi1ELb0ELb1ELb0E"])
def test_batch_lookup_fetches_the_whole_entry_at_once(batch_asm, kernel):
    """Behind the index loads (the only scalar loads with a register offset) the entry arrives as one group of loads followed by
    one wait; the next scalar load may only be the bisection's (one dword, BatchEntry::end_wg) or the entry it ends on."""
    lines = kernel_body(batch_asm, kernel)
    # base[wg / 4096] (the only scalar load with a register offset) and the dword of delta[wg / 64] (its address depends on the
    # index form, byte or 16-bit deltas, so it is computed first), issued together: no wait between them
    reg = [i for i, l in enumerate(lines) if l.startswith("s_load_dword ") and re.search(r", s\d+ offset:", l)]
    assert len(reg) == 1, reg
    nxt = next(i for i in range(reg[0] + 1, len(lines)) if lines[i].startswith("s_load_dword"))
    assert lines[nxt].startswith("s_load_dword "), lines[nxt]
    idx = [reg[0], nxt]
    assert not any(l.startswith("s_waitcnt") for l in lines[idx[0] + 1:idx[1]])
    wait = next(i for i in range(idx[1], len(lines)) if lines[i].startswith("s_waitcnt lgkmcnt(0)"))
    group = []
    i = wait + 1
    while not lines[i].startswith("s_waitcnt lgkmcnt(0)"):
        if lines[i].startswith("s_load"):
            group.append(lines[i])
        i += 1
    bytes_loaded = sum(4 * int(re.match(r"s_load_dword(x(\d+))?", l).group(2) or 1) for l in group)
    # sizeof(BatchEntry) = 96, of which a format with three streams needs 72 (the bases of streams it does not have are dead):
    # nothing the tile needs is left for a later round trip
    assert 72 <= bytes_loaded <= 96, (bytes_loaded, group)


@pytest.mark.parametrize("prefix, kernel", [("_ZN5dxtlt14fwd_tiled_halo", "ILi3ELi1ELb1ELb1ELi0ELb1E"), ("_ZN5dxtlt15inv_tiled_shift", "ILi3ELi1ELb1ELb1E"),
                                            ("_ZN5dxtlt14fwd_tiled_halo", "ILi1ELi1ELb0ELb1ELi0ELb1E"), ("_ZN5dxtlt15inv_tiled_shift", "ILi1ELi1ELb0ELb1E")])
def test_single_call_halo_and_shifted_kernels_fetch_their_arguments_in_one_round_trip(single_asm, prefix, kernel):
    """fwd_tiled_halo / inv_tiled_shift take a 110-byte Shifts argument.  Left alone the compiler fetches an argument in the block
    that first uses it -- two or three dependent scalar round trips in front of the tile's load, 0.77 against 0.80 of peak
    (profiles/r04_batch_edge_tiles.txt section 3).  shifts_fetched_at_once() pins them: between the first wait and the first
    vector memory instruction no further load from the kernel-argument segment may appear."""
    lines = kernel_body(single_asm, kernel, prefix)
    first_load = next(l for l in lines if l.startswith("s_load_dword"))
    karg = re.search(r", (s\[\d+:\d+\]), ", first_load).group(1)
    wait = next(i for i, l in enumerate(lines) if l.startswith("s_waitcnt lgkmcnt(0)"))
    vmem = next(i for i, l in enumerate(lines) if l.startswith(("global_load", "global_store", "flat_", "buffer_")))
    assert wait < vmem
    late = [l for l in lines[wait:vmem] if l.startswith("s_load_dword") and f", {karg}, " in l]
    assert not late, late
    assert not any(l.startswith("flat_") for l in lines), "a pinned pointer lost its address space: flat accesses in the tile"


@pytest.mark.parametrize("prefix, kernel, round4_scalars", [
    ("_ZN5dxtlt14fwd_tiled_halo", "ILi3ELi1ELb1ELb1ELi0ELb1E", 535), ("_ZN5dxtlt14fwd_tiled_halo", "ILi1ELi1ELb0ELb1ELi0ELb1E", 474),
    ("_ZN5dxtlt15inv_tiled_shift", "ILi3ELi1ELb1ELb1E", 728), ("_ZN5dxtlt15inv_tiled_shift", "ILi1ELi1ELb0ELb1E", 828)])
def test_product_kernels_carry_no_experiment_scalars(single_asm, prefix, kernel, round4_scalars):
    """Round 4's halo / shifted kernels pinned 19 Shifts fields into SGPRs, three of them experiment switches (tile order, store
    policy, a wrong-output timing switch) that every workgroup fetched and branched on.  Those now exist only with
    -DDXTLT_EXPERIMENTS: the product kernels' static scalar-instruction counts must stay below the round-4 figures (halo tiles:
    535 -> 486 and 474 -> 427; shifted tiles, which only lost the tile-order switch: 728 -> 721 and 828 -> 822)."""
    lines = kernel_body(single_asm, kernel, prefix)
    scalars = sum(1 for l in lines if l.startswith("s_") and not l.endswith(":"))
    assert scalars < round4_scalars, (scalars, round4_scalars)
    if "halo" in prefix:
        assert scalars <= round4_scalars - 40, (scalars, round4_scalars)
    assert not any("generic_kernel" in l or "fwd_tiled_shift" in l for l in single_asm.splitlines() if l.startswith("_ZN"))
