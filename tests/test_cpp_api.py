"""Builds and runs the C++ host-API test program (tests/cpp/test_cpp_api.cpp) against the in-tree library."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "test_cpp_api")


@pytest.fixture(scope="module")
def exe(pkg):
    libdir = os.path.dirname(pkg._lib.lib_path())
    src = os.path.join(ROOT, "tests", "cpp", "test_cpp_api.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-o", EXE, src, f"-L{libdir}",
                           "-ldxtlt_gfx950", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"])
    return EXE


def test_cpp_api_validation_paths(exe):
    r = subprocess.run([exe, "cpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
def test_cpp_api_on_device(exe):
    r = subprocess.run([exe, "gpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
