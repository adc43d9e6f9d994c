"""N > 1 path on CPU: two gloo ranks, contiguous block-range shards, no data-path collective.

What is under test is the host logic the multi-GPU paths share -- shard planning (plan_shards), the stream table and
the per-stream placement of a shard's result (sharding.py, mirrored by shard_worker in csrc/dxtlt_api.cpp) -- plus
the bench's rendezvous pattern (barrier, MAX-reduce of the elapsed time).  The per-shard transform itself is done by
the CPU oracle here (no GPU in this container); on the GPU box the same composition is exercised with the HIP kernels
by tests/test_gpu_parity.py::test_range_calls_compose_to_the_whole_buffer and ::test_sharded_entry_point_on_one_gpu.
The gather at the end is test plumbing to compare against the whole-buffer oracle, not part of the product path."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fmt, total_blocks, settings, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import time

    import torch
    import torch.distributed as dist

    import dxt_lossless_transform_amd as pkg
    from dxt_lossless_transform_amd import sharding
    from helpers import BLOCK, pkg_settings
    from oracle import oracle_c

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        st = pkg_settings(pkg, fmt, settings)
        table = pkg.stream_table(fmt, st)
        plan = pkg.plan_shards(total_blocks, world)
        first, count = plan[rank]
        B = BLOCK[fmt]
        # every rank generates ITS slice of the one logical block array (as bench.py does on the device)
        mine = oracle_c.fill_splitmix64(count * B, 0x0A5C0005, first * B // 8)

        dist.barrier()
        t0 = time.perf_counter()
        shard_soa = oracle_c.transform(fmt, mine, settings[0], settings[2], settings[1])
        elapsed = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
        dist.barrier()
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)  # bench.py: time = max over ranks

        # ---- test plumbing: collect shards on rank 0 and compare with the whole-buffer oracle ----
        sizes = [n * B for _, n in plan]
        if rank == 0:
            parts = [torch.empty(s, dtype=torch.uint8) for s in sizes]
            parts[0] = torch.from_numpy(shard_soa)
            for r in range(1, world):
                dist.recv(parts[r], src=r)
            whole_in = oracle_c.fill_splitmix64(total_blocks * B, 0x0A5C0005)
            whole = np.zeros(total_blocks * B, dtype=np.uint8)
            for r, (f, n) in enumerate(plan):
                sharding.scatter_shard_streams(whole, parts[r].numpy(), total_blocks, f, n, table)
            want = oracle_c.transform(fmt, whole_in, settings[0], settings[2], settings[1])
            ok_fwd = bool(np.array_equal(whole, want))
            # inverse: hand every rank its packed slice of every stream
            for r, (f, n) in enumerate(plan):
                packed = sharding.gather_shard_streams(want, total_blocks, f, n, table)
                if r == 0:
                    my_packed = packed
                else:
                    dist.send(torch.from_numpy(packed), dst=r)
        else:
            dist.send(torch.from_numpy(shard_soa), dst=0)
            buf = torch.empty(count * B, dtype=torch.uint8)
            dist.recv(buf, src=0)
            my_packed = buf.numpy()
            ok_fwd = True
        back = oracle_c.transform(fmt, my_packed, settings[0], settings[2], settings[1], inverse=True)
        ok_inv = bool(np.array_equal(back, mine))
        flags = torch.tensor([int(ok_fwd), int(ok_inv)], dtype=torch.int32)
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)
        if rank == 0:
            q.put((int(flags[0]), int(flags[1]), float(elapsed.item()) > 0.0))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fmt,total,settings", [
    ("bc1", 50_000, (1, 0, 1)),        # shard boundary 24 576: tile-aligned
    ("bc3", 10_007, (1, 1, 1)),        # odd total: last shard takes the ragged remainder
    ("bc2", 4_100, (2, 0, 0)),
    ("bc3", 4_099, (0, 0, 1)),
])
def test_two_rank_shards_compose(fmt, total, settings):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, fmt, total, settings, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    ok_fwd, ok_inv, timed = q.get(timeout=5)
    assert ok_fwd == 1 and ok_inv == 1 and timed


def test_mixed_bc1_bc3_archive_placement(oracle, pkg):
    """BASELINE.json configs[4] in miniature: an archive of alternating BC1 / BC3 textures, each transformed with its
    format's default settings, block ranges split over 8 'GPUs'; placement must reproduce the unsharded result."""
    from dxt_lossless_transform_amd import sharding
    from helpers import BLOCK

    rng = np.random.default_rng(5)
    for i in range(6):
        fmt = "bc1" if i % 2 == 0 else "bc3"
        st = pkg.Bc1TransformSettings() if fmt == "bc1" else pkg.Bc3TransformSettings()
        blocks = int(rng.integers(3_000, 40_000))
        x = oracle.fill_splitmix64(blocks * BLOCK[fmt], 0x0A5C0005 + i)
        want = oracle.transform(fmt, x)
        table = pkg.stream_table(fmt, st)
        whole = np.zeros_like(x)
        for first, count in pkg.plan_shards(blocks, 8):
            shard = oracle.transform(fmt, x[first * BLOCK[fmt]:(first + count) * BLOCK[fmt]])
            sharding.scatter_shard_streams(whole, shard, blocks, first, count, table)
        assert np.array_equal(whole, want)
