"""Multi-GPU sharding of the BC7 mode-split transform (per-mode counts exchanged on the host, no
collective).  CPU part: the placement table (dxtlt_bc7_shard_pieces, host code) assembles per-shard ORACLE transforms
into exactly the whole-buffer oracle transform, for ragged partitions and empty mode classes.  GPU part: the sharded
entry points with more shards than devices (round robin on the one GPU of the test box) equal the oracle."""
import numpy as np
import pytest

from oracle import oracle_np as onp
from tests.test_bc7 import make_blocks


@pytest.fixture(scope="module")
def bc7(pkg):
    from dxt_lossless_transform_amd import bc7 as mod

    return mod


def partition(n, cuts):
    edges = [0] + sorted(cuts) + [n]
    return [(a, b - a) for a, b in zip(edges[:-1], edges[1:])]


def mode_counts(blocks):
    modes = onp.bc7_modes(blocks.reshape(-1, 16)[:, 0])
    return [int((modes == m).sum()) for m in range(9)]


@pytest.mark.parametrize("kind", ["uniform", "mode6", "skewed", "raw"])
def test_placement_table_assembles_the_whole_transform(pkg, bc7, oracle, kind):
    n = 5000
    x = make_blocks(oracle, n, kind, 21)
    want = oracle.transform_bc7(x)
    for cuts in ([], [2500], [1, 2, 4999], [1024, 2048, 3072, 4096], [777, 778, 3001]):
        parts = partition(n, cuts)
        counts = [mode_counts(x[16 * f: 16 * (f + c)]) for f, c in parts]
        firsts, nums = [f for f, _ in parts], [c for _, c in parts]
        got = np.full(x.size, 0xEE, dtype=np.uint8)
        covered = np.zeros(x.size, dtype=np.int32)
        for s, (f, c) in enumerate(parts):
            local = oracle.transform_bc7(x[16 * f: 16 * (f + c)])
            g, l, b = bc7.shard_pieces(counts, s, firsts, nums, n)
            assert sum(b) == 16 * c
            for p in range(19):
                got[g[p]: g[p] + b[p]] = local[l[p]: l[p] + b[p]]
                covered[g[p]: g[p] + b[p]] += 1
        assert (covered == 1).all(), (kind, cuts)           # the pieces tile the buffer exactly once
        assert np.array_equal(got, want), (kind, cuts)


def test_placement_table_rejects_inconsistent_input(pkg, bc7):
    counts = [[1, 0, 0, 0, 0, 0, 0, 0, 0], [0, 2, 0, 0, 0, 0, 0, 0, 0]]
    g, l, b = bc7.shard_pieces(counts, 1, [0, 1], [1, 2], 3)
    assert b[0] == 2 and g[0] == 1 and l[0] == 0            # `first` piece of shard 1: two bytes at offset 1
    assert b[2] == 2 * 9 and g[2] == 3 + 15 and b[11] == 2 * 6   # head_1 after mode 0's 15 bytes; tail_1
    with pytest.raises(pkg.DeviceError):                     # counts do not add up to the shard's size
        bc7.shard_pieces(counts, 0, [0, 1], [2, 2], 4)
    with pytest.raises(pkg.DeviceError):                     # ranges are not contiguous
        bc7.shard_pieces(counts, 0, [0, 2], [1, 2], 4)
    with pytest.raises(pkg.DeviceError):
        bc7.shard_pieces(counts, 2, [0, 1], [1, 2], 3)


torch = pytest.importorskip("torch")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["uniform", "skewed", "raw"])
def test_gpu_sharded_equals_oracle(pkg, bc7, oracle, kind):
    for n in (1, 5, 1023, 1024, 4099, 300_001):
        x = make_blocks(oracle, n, kind, n + 3)
        want = oracle.transform_bc7(x)
        for shards in (0, 1, 2, 3, 7, 64):
            y = np.zeros_like(x)
            bc7.transform_bc7_sharded(x, y, shards)
            assert np.array_equal(y, want), (kind, n, shards)
            z = np.zeros_like(x)
            bc7.transform_bc7_sharded(y, z, shards, inverse=True)
            assert np.array_equal(z, x), (kind, n, shards, "inverse")
    with pytest.raises(pkg.InvalidLength):
        bc7.transform_bc7_sharded(np.zeros(24, np.uint8), np.zeros(24, np.uint8), 2)


@pytest.mark.gpu
def test_gpu_sharded_256_mib(pkg, bc7, oracle):
    n = (256 << 20) // 16
    x = make_blocks(oracle, n, "uniform", 77)
    y, z = np.zeros_like(x), np.zeros_like(x)
    bc7.transform_bc7_sharded(x, y, 5)
    whole = np.zeros_like(x)
    bc7.transform_bc7(x, whole)
    assert np.array_equal(y, whole)
    bc7.transform_bc7_sharded(y, z, 3, inverse=True)
    assert np.array_equal(z, x)
