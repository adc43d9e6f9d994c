"""bench.py and __graft_entry__.py are run by the driver on a GPU box only; these CPU checks keep them importable and
their command line stable (the contract: --gpus / --steps / --warmup, plus this build's --format / --workload)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_help_lists_the_contract_flags():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    for flag in ("--gpus", "--steps", "--warmup", "--format", "--workload", "--size-gib", "--no-cpu-baseline"):
        assert flag in r.stdout, flag
    assert "bc7" in r.stdout and "archive" in r.stdout


def test_bench_defaults():
    sys.path.insert(0, ROOT)
    import bench

    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        a = bench.parse_args()
    finally:
        sys.argv = argv
    assert (a.gpus, a.format, a.workload) == (1, "bc1", "buffer") and a.steps > 0 and a.warmup >= 0


def test_graft_entry_has_build_and_smoke():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g

    assert callable(g.build) and callable(g.smoke)
