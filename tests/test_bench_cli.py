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
    assert {k: rec[k] for k in ("rendezvous", "n_gpus", "max_over_ranks", "backend", "world_size_seen", "per_rank")} == {
        "rendezvous": "ok", "n_gpus": 2, "max_over_ranks": 2.0, "backend": "gloo", "world_size_seen": 2, "per_rank": [[1.0, 0.0], [2.0, 10.0]]}
    assert rec["config"]["total_blocks"] == 2 << 30 and rec["ranges"] == [[0, 0, 1 << 30], [1, 1 << 30, 1 << 30]]


def test_stdout_of_a_launcher_run_is_one_json_line():
    """The driver's own shape for N > 1: `python -m torch.distributed.run ... bench.py --gpus N`.  The gloo transport
    prints "[Gloo] Rank r is connected to ..." on stdout from C++; bench.py moves that to stderr, so stdout is the one
    JSON line of rank 0 and nothing else.  Three ranks, --rendezvous-only (no GPU)."""
    import json
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["DXTLT_BENCH_BACKEND"] = "gloo"
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
                        "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "3", "--rendezvous-only"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert {k: rec[k] for k in ("rendezvous", "n_gpus", "max_over_ranks", "backend", "world_size_seen", "per_rank")} == {
        "rendezvous": "ok", "n_gpus": 3, "max_over_ranks": 3.0, "backend": "gloo", "world_size_seen": 3,
        "per_rank": [[1.0, 0.0], [2.0, 10.0], [3.0, 20.0]]}


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


# ---- on the GPU box: the bench itself, small -----------------------------------------------------------------------
import pytest  # noqa: E402


def _run_bench(extra, env=None, timeout=600):
    import json

    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                        "--size-gib", "0.25", *extra], capture_output=True, text=True, timeout=timeout, env=e)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_has_the_contract_fields_and_the_host_array_leg():
    d = _run_bench(["--host-array-gib", "0.25"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["dtype"] == "u8" and d["vs_baseline"] is None
    assert d["config"]["bit_exact_roundtrip_and_oracle_window"] is True
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    # a 0.25 GiB run is not the workload of profiles/pmc_traffic.json: no carried traffic figure
    assert d["roofline"]["traffic"] is None and d["roofline"]["traffic_source"] is None
    h = d["sharded_host_array"]
    assert h["devices"] >= 1 and h["bit_exact_roundtrip_and_oracle_windows_across_shard_boundaries"] is True


@pytest.mark.gpu
def test_default_line_carries_the_other_single_gpu_configs_as_legs():
    """BC3, BC7 (uniform and skewed mode mix) and the archive slice ride in the default BC1 line as `legs`; here on a
    small buffer.  Each leg has both directions' times, a roofline fraction and its own exactness flags."""
    d = _run_bench(["--host-array-gib", "0", "--small-legs", "--leg-steps", "2"])
    legs = d["legs"]
    assert set(legs) == {"bc3", "bc2", "bc7_uniform", "bc7_skewed", "archive", "corpus", "corpus_bc3"}
    for name, leg in legs.items():
        assert leg["bit_exact_roundtrip"] is True, name
        assert leg["fwd_ms"] > 0 and leg["inv_ms"] > 0 and 0 < leg["roofline"]["frac"] < 1, name
        assert 0 < leg["roofline"]["inverse_kernel"]["frac"] < 1, name
    assert legs["bc3"]["oracle_window_exact"] and legs["bc2"]["oracle_window_exact"] and legs["archive"]["oracle_windows_exact"]
    for name in ("corpus", "corpus_bc3"):     # the reference's benchmark shape in miniature: mip-chained, odd block counts, one batch call
        assert legs[name]["oracle_textures_exact"] is True and legs[name]["textures"] >= 9
        assert legs[name]["forward_gaps_exact"] is True and legs[name]["oracle_textures_checked"] >= 3
        assert legs[name]["largest_blocks"] == 1398103 and legs[name]["smallest_blocks"] % 2 == 1   # 4096 x 4096 with mips; odd counts
    assert legs["bc7_uniform"]["oracle_prefix_exact"] and legs["bc7_skewed"]["oracle_prefix_exact"]
    assert legs["bc7_skewed"]["mode_counts"][6] > 2 * legs["bc7_uniform"]["mode_counts"][6]
    # the driver's record keeps `config` and drops `legs`: every leg rides there in short, nested and as flat scalars
    cfg = d["config"]
    assert set(cfg["legs_summary"]) == set(legs) and cfg["legs_all_exact"] is True and cfg["legs_inexact"] == []
    assert_driver_keeps_every_config(list(cfg))
    assert 0 < cfg["inv_frac"] < 1 and cfg["inv_frac"] == d["roofline"]["inv_frac"] == d["roofline"]["inverse_kernel"]["frac"]
    for name, leg in legs.items():
        f, i, ok = cfg["legs_summary"][name]
        assert f == leg["roofline"]["frac"] == cfg[f"leg_{name}_fwd_frac"], name
        assert i == leg["roofline"]["inverse_kernel"]["frac"] == cfg[f"leg_{name}_inv_frac"], name
        assert ok is True and cfg[f"leg_{name}_exact"] is True, name
    # without the switch a small buffer has no legs (they are defined on the BASELINE sizes)
    assert "legs" not in _run_bench(["--host-array-gib", "0"])


@pytest.mark.gpu
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_self_launched_on_one_gpu(scaling):
    """`python bench.py --gpus 2` from a bare shell: the parent starts the ranks; here they share the one GPU over gloo.
    Strong scaling = one array, each rank its block range through dxtlt_transform_range_device."""
    d = _run_bench(["--gpus", "2", "--scaling", scaling, "--host-array-gib", "0"], env={"DXTLT_BENCH_BACKEND": "gloo"})
    assert d["n_gpus"] == 2 and d["scaling"] == scaling
    assert d["world_size_seen"] == 2 and [r["rank"] for r in d["per_rank"]] == [0, 1]
    assert all(r["fwd_ms"] > 0 and r["inv_ms"] > 0 and r["elapsed_s"] > 0 for r in d["per_rank"])
    assert all(r["round_trip_exact"] is True and r["oracle_window_exact"] is True for r in d["per_rank"])      # every rank, its own data
    assert d["config"]["bit_exact_roundtrip_and_oracle_window"] is True
    total = d["config"]["total_blocks"]
    assert total == (d["config"]["blocks_per_gpu"] * 2)      # strong: the array split in two; weak: two shards of one array
    assert ("ONE logical" in d["config"]["workload"]) == (scaling == "strong")


@pytest.mark.gpu
def test_bench_bc7_and_archive_lines():
    d = _run_bench(["--format", "bc7"])
    assert d["config"]["format"] == "bc7" and d["roofline"]["pipeline_bytes"] == d["roofline"]["algorithmic_bytes_per_launch"]
    a = _run_bench(["--workload", "archive", "--size-gib", "1", "--gpus", "2", "--archive-split", "range"],
                   env={"DXTLT_BENCH_BACKEND": "gloo"})
    assert a["config"]["archive_split"] == "range" and a["config"]["bit_exact_roundtrip_and_oracle_windows"] is True
    assert len(a["per_rank"]) == 2 and all(r["round_trip_exact"] is True and r["oracle_windows_exact"] is True for r in a["per_rank"])
    b = _run_bench(["--format", "bc7", "--gpus", "2"], env={"DXTLT_BENCH_BACKEND": "gloo"})
    assert len(b["per_rank"]) == 2 and all(r["round_trip_exact"] is True and r["oracle_prefix_exact"] is True for r in b["per_rank"])


def test_physical_cores_counts_cores_not_threads():
    """cpu_baseline quotes physical cores (north_star: "core count stated"), not SMT threads."""
    sys.path.insert(0, ROOT)
    import bench

    n = bench.physical_cores()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_eight_rank_rendezvous_the_driver_shape():
    """The driver's N = 8 launch, without a GPU: `python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8
    --rendezvous-only` -- eight ranks rendezvous over gloo on 127.0.0.1, barrier, MAX-reduce, rank 0 prints the one line."""
    import json
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8",
                        "--rendezvous-only"], capture_output=True, text=True, timeout=600, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    # world_size_seen is the process group's own count; per_rank = every rank's figures in rank order (all_gather): the keys a
    # real N > 1 line carries so that the first SCALE record shows stragglers and proves the group saw N ranks
    rec = json.loads(lines[0])
    assert {k: rec[k] for k in ("rendezvous", "n_gpus", "max_over_ranks", "backend", "world_size_seen", "per_rank")} == {
        "rendezvous": "ok", "n_gpus": 8, "max_over_ranks": 8.0, "backend": "gloo", "world_size_seen": 8,
        "per_rank": [[float(r + 1), float(10 * r)] for r in range(8)]}
    # the numbers the first real SCALE record must carry at N = 8 (weak scaling, BC1, 8 GiB per GPU): eight per_rank rows, the
    # process group's own count, 8 x 2^30 blocks in all, rank r on blocks [r * 2^30, (r + 1) * 2^30) -- job_shape() is the
    # function the real run takes its ranges from
    assert rec["world_size_seen"] == 8 and len(rec["per_rank"]) == 8 and rec["scaling"] == "weak"
    assert rec["config"] == {"format": "bc1", "total_blocks": 8 * 2**30, "blocks_per_gpu": 2**30}
    assert rec["ranges"] == [[r, r * 2**30, 2**30] for r in range(8)]
    # every rank verified a window of ITS OWN range (rank-dependent data) and the flags were all-gathered: N rows, N x 2 true flags
    assert [row["rank"] for row in rec["per_rank_flags"]] == list(range(8))
    assert all(row["round_trip_exact"] is True and row["oracle_window_exact"] is True for row in rec["per_rank_flags"])
    assert rec["bit_exact_roundtrip_and_oracle_window"] is True


@pytest.mark.parametrize("world,scaling", [(2, "weak"), (4, "strong")])
def test_every_rank_reports_its_own_exactness_flags(world, scaling):
    """bench.py --gpus N from a bare shell (self-launch), gloo, no device: the N > 1 record's `per_rank` rows each carry the rank's
    own round-trip and oracle-window flags, and the line's flag is their AND."""
    import json

    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--rendezvous-only", "--scaling", scaling,
                        "--format", "bc3", "--size-gib", "0.25"], capture_output=True, text=True, timeout=600, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][0])
    assert rec["world_size_seen"] == world and len(rec["per_rank_flags"]) == world
    assert sum(row["round_trip_exact"] is True for row in rec["per_rank_flags"]) == world
    assert sum(row["oracle_window_exact"] is True for row in rec["per_rank_flags"]) == world
    assert rec["bit_exact_roundtrip_and_oracle_window"] is True


def test_all_ranks_exact_is_false_when_one_rank_disagrees():
    sys.path.insert(0, ROOT)
    import bench

    rows = {"per_rank": [{"rank": 0, "round_trip_exact": True, "oracle_window_exact": True},
                         {"rank": 1, "round_trip_exact": True, "oracle_window_exact": False}]}
    assert bench.all_ranks_exact(rows, ("round_trip_exact",)) is True
    assert bench.all_ranks_exact(rows, ("round_trip_exact", "oracle_window_exact")) is False
    assert bench.all_ranks_exact({"per_rank": [{"rank": 0}]}, ("round_trip_exact",)) is False


def test_config_is_ordered_for_the_24_members_the_driver_keeps():
    """The driver's BENCH record keeps the first 24 members of `config`: after `workload` they are `legs_all_exact`, `legs_inexact`
    and both roofline fractions of every BASELINE config (configs[2], configs[3] on both mode mixes, configs[4]'s share, the corpus
    legs, BC2), then the headline's own flag / fractions / times; the descriptive members come behind."""
    sys.path.insert(0, ROOT)
    import bench

    names = ("bc3", "bc2", "bc7_uniform", "bc7_skewed", "archive", "corpus", "corpus_bc3")       # the order run_legs produces them in
    legs = {n: {"roofline": {"frac": 0.8, "inverse_kernel": {"frac": 0.79}}, "bit_exact_roundtrip": True, "oracle_window_exact": n != "bc2"}
            for n in names}
    cfg = {"workload": "w", "mode": "m", "format": "bc1", "blocks_per_gpu": 1, "bytes_per_gpu": 8, "total_blocks": 1, "seed": "0x1",
           "sharding": "s", "bit_exact_roundtrip_and_oracle_window": True, "fwd_ms": 1.0, "inv_ms": 1.0, "fwd_GiBps": 1.0, "inv_GiBps": 1.0,
           "fwd_frac": 0.8, "inv_frac": 0.8}
    out = {"legs": legs, "config": dict(cfg)}
    bench.summarize_legs_into_config(out)
    keys = list(out["config"])
    assert_driver_keeps_every_config(keys)
    assert out["config"]["legs_inexact"] == ["bc2"] and out["config"]["legs_all_exact"] is False
    assert set(cfg) <= set(keys) and all(out["config"][k] == v for k, v in cfg.items())            # nothing lost, only reordered
    assert out["config"]["legs_summary"]["corpus"] == [0.8, 0.79, True] and out["config"]["leg_bc2_exact"] is False


def assert_driver_keeps_every_config(keys):
    sys.path.insert(0, ROOT)
    import bench

    first = keys[:bench.DRIVER_KEEPS]
    assert bench.DRIVER_KEEPS == 24 and first[:3] == ["workload", "legs_all_exact", "legs_inexact"]
    for name in ("bc3", "bc7_uniform", "bc7_skewed", "archive", "corpus", "corpus_bc3", "bc2"):
        assert f"leg_{name}_fwd_frac" in first and f"leg_{name}_inv_frac" in first, name
    for k in ("bit_exact_roundtrip_and_oracle_window", "fwd_frac", "inv_frac", "fwd_ms", "inv_ms"):
        assert k in first, k
    for k in ("fwd_GiBps", "inv_GiBps", "mode", "sharding", "blocks_per_gpu", "bytes_per_gpu"):
        assert k in keys and k not in first, k


def test_strong_scaling_ranges_partition_the_array():
    """--scaling strong over eight ranks: the block ranges of job_shape() tile [0, total) without gap or overlap."""
    import argparse

    sys.path.insert(0, ROOT)
    import bench
    import dxt_lossless_transform_amd as pkg

    for size_gib, drop in ((8.0, 0), (8.0, 1), (0.25, 3)):
        a = argparse.Namespace(size_gib=size_gib, drop_blocks=drop, scaling="strong")
        shapes = [bench.job_shape(a, 8, 8, r, pkg.plan_shards) for r in range(8)]
        total = shapes[0][0]
        assert all(s[0] == total for s in shapes) and total == int(size_gib * 2**30) // 8 - drop
        at = 0
        for _, first, blocks in shapes:
            assert first == at and blocks > 0
            at += blocks
        assert at == total


def test_corpus_leg_has_the_shape_of_the_references_published_benchmark():
    """legs.corpus rebuilds the reference's own benchmark shape (api/dxt-lossless-transform-bc1-api/README.MD:286-311: "2130
    real files (8692.9 MiB)" of BC1 DDS textures): 2130 textures with full mip chains whose block bytes add up to 8692.9 MiB,
    every block count odd (the 4 + 1 + 1 + 1 blocks of the 8x8 .. 1x1 levels), laid out at 256-byte boundaries."""
    sys.path.insert(0, ROOT)
    import bench

    assert bench.mip_chain_blocks(4, 4) == 1 + 1 + 1                      # 4x4, 2x2, 1x1: a block each
    assert bench.mip_chain_blocks(256, 256) == (4 ** 7 - 1) // 3 + 2      # 64^2 + 32^2 + ... + 1, + the 2x2 and 1x1 levels
    assert bench.mip_chain_blocks(4096, 2048) == 699053
    texs = bench.corpus_textures()
    assert len(texs) == 2130 and all(n % 2 == 1 for _, _, n in texs)
    assert round(sum(n for _, _, n in texs) * 8 / 2**20, 1) == 8692.9
    assert texs != sorted(texs, key=lambda t: t[2])                       # a directory walk, not a size-sorted list
    offs, arena = bench.corpus_layout(texs, 8)
    assert all(o % 256 == 0 for o in offs) and arena >= sum(n for _, _, n in texs) * 8
    assert all(b - a >= n * 8 for (a, b), (_, _, n) in zip(zip(offs, offs[1:] + [arena]), texs))
