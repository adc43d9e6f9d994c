"""AddressSanitizer + UBSan over the host side of the library, on the CPU (the reference has no
sanitizer lane, its safety rests on tests with +-1-byte misalignment; GPU sanitizers are not available on the pool).
tools/asan_host_check.sh rebuilds every .cpp of the product with clang -fsanitize=address,undefined, links them with
the normal kernel objects and runs the device-free C-ABI suites (argument validation, DDS parser with hostile headers,
TransformHeader bits, BC7 shard placement, batch planning, no-device error paths) against that build."""
import glob
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(os.environ.get("DXTLT_LIB_PATH") is not None, reason="already inside the sanitizer run")
def test_host_side_is_clean_under_asan_and_ubsan(pkg):
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++") or not glob.glob(
            "/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so") or shutil.which("bash") is None:
        pytest.skip("no clang sanitizer runtime in this image")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan_host_check.sh")], cwd=ROOT, capture_output=True, text=True,
                       timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
