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


def test_bench_self_launches_its_ranks_from_a_bare_shell():
    """`python bench.py --gpus 2` with no launcher environment must start torch.distributed.run itself (as a child,
    before anything touches a GPU), relay rank 0's JSON line and return the child's exit code.  --rendezvous-only
    stops after the rank plumbing (gloo barrier + MAX-reduce), so this runs without a GPU."""
    import json

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["DXTLT_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec == {"rendezvous": "ok", "n_gpus": 2, "max_over_ranks": 2.0, "backend": "gloo"}


def test_bench_self_launch_returns_the_childs_exit_code():
    """No GPU here: the ranks of a real run fail at 'bench.py needs a GPU'; the parent must report that failure."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["DXTLT_BENCH_BACKEND"] = "gloo"
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("needs a box without a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert "needs a GPU" in r.stderr


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rendezvous-only"],
                       capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr
