"""N > 1 path on CPU: two and four gloo ranks, contiguous block-range shards, no data-path collective.

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


def _run_ranks(target, world, args):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, *args, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    return q.get(timeout=5)


@pytest.mark.parametrize("world,fmt,total,settings", [
    (2, "bc1", 50_000, (1, 0, 1)),        # shard boundary 24 576: tile-aligned
    (2, "bc3", 10_007, (1, 1, 1)),        # odd total: last shard takes the ragged remainder
    (2, "bc2", 4_100, (2, 0, 0)),
    (2, "bc3", 4_099, (0, 0, 1)),
    (4, "bc1", 70_001, (1, 0, 1)),        # four ranks: three whole shares and a ragged last one
    (4, "bc3", 33_333, (1, 1, 1)),
])
def test_rank_shards_compose(world, fmt, total, settings):
    ok_fwd, ok_inv, timed = _run_ranks(_worker, world, (fmt, total, settings))
    assert ok_fwd == 1 and ok_inv == 1 and timed


def _bc7_worker(rank, world, port, total_blocks, kind, q):
    """BC7 over `world` ranks: shards are runs of whole 1024-block granules (the last one takes the tail part), each rank
    transforms its run as a stand-alone buffer and the placement table (dxtlt_bc7_shard_pieces, host code) says where its
    pieces go in the whole transformed buffer -- no counters to exchange, no collective on the data path."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist

    import dxt_lossless_transform_amd as pkg  # noqa: F401  (loads the library: the placement table is host code in it)
    from dxt_lossless_transform_amd import bc7
    from oracle import oracle_c
    from tests.test_bc7 import make_blocks

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        G = bc7.sort_granule()
        granules = (total_blocks + G - 1) // G
        share = granules // world
        first = rank * share * G
        count = (total_blocks - first) if rank == world - 1 else share * G
        whole_in = make_blocks(oracle_c, total_blocks, kind, 99)          # every rank can regenerate the array; it uses its slice
        mine = whole_in[16 * first: 16 * (first + count)]
        local = oracle_c.transform_bc7(mine)
        g, l, b = bc7.shard_pieces(total_blocks, first, count)
        # ---- test plumbing: rank 0 assembles and compares ----
        if rank == 0:
            got = np.full(16 * total_blocks, 0xEE, dtype=np.uint8)
            covered = np.zeros(16 * total_blocks, dtype=np.int32)
            for r in range(world):
                if r == 0:
                    loc, gg, ll, bb = local, g, l, b
                else:
                    meta = torch.empty(27, dtype=torch.int64)
                    dist.recv(meta, src=r)
                    gg, ll, bb = meta[:9].tolist(), meta[9:18].tolist(), meta[18:].tolist()
                    buf = torch.empty(sum(bb), dtype=torch.uint8)
                    dist.recv(buf, src=r)
                    loc = buf.numpy()
                for p in range(9):
                    got[gg[p]: gg[p] + bb[p]] = loc[ll[p]: ll[p] + bb[p]]
                    covered[gg[p]: gg[p] + bb[p]] += 1
            want = oracle_c.transform_bc7(whole_in)
            ok = bool((covered == 1).all()) and bool(np.array_equal(got, want))
            # inverse: every rank gets its pieces back and must recover its slice
            for r in range(1, world):
                dist.send(torch.from_numpy(want), dst=r)
            mine_t = want
        else:
            dist.send(torch.tensor(list(g) + list(l) + list(b), dtype=torch.int64), dst=0)
            dist.send(torch.from_numpy(local), dst=0)
            t = torch.empty(16 * total_blocks, dtype=torch.uint8)
            dist.recv(t, src=0)
            mine_t = t.numpy()
            ok = True
        packed = np.zeros(16 * count, dtype=np.uint8)
        for p in range(9):
            packed[l[p]: l[p] + b[p]] = mine_t[g[p]: g[p] + b[p]]
        ok_inv = bool(np.array_equal(oracle_c.transform_bc7(packed, inverse=True), mine))
        flags = torch.tensor([int(ok), int(ok_inv)], dtype=torch.int32)
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)
        if rank == 0:
            q.put((int(flags[0]), int(flags[1])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,total,kind", [(2, 5_000, "uniform"), (4, 9 * 1024 + 77, "skewed"), (4, 8 * 1024, "uniform")])
def test_bc7_granule_shards_over_ranks(world, total, kind):
    ok_fwd, ok_inv = _run_ranks(_bc7_worker, world, (total, kind))
    assert ok_fwd == 1 and ok_inv == 1


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
