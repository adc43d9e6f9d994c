"""Multi-GPU sharding of the BC7 granule-sorted field split (contiguous ranges, no collective; in
version 1 of the format no counters are exchanged either -- a granule's output depends on that granule alone).  CPU
part: the placement table (dxtlt_bc7_shard_pieces, host code) assembles per-shard ORACLE transforms into exactly the
whole-buffer oracle transform.  GPU part: the sharded entry points with more shards than devices (round robin on the
one GPU of the test box) equal the oracle."""
import numpy as np
import pytest

from tests.test_bc7 import make_blocks


@pytest.fixture(scope="module")
def bc7(pkg):
    from dxt_lossless_transform_amd import bc7 as mod

    return mod


def partition(n, cuts):
    edges = [0] + sorted(cuts) + [n]
    return [(a, b - a) for a, b in zip(edges[:-1], edges[1:])]


@pytest.mark.parametrize("kind", ["uniform", "mode6", "skewed", "raw"])
def test_placement_table_assembles_the_whole_transform(pkg, bc7, oracle, kind):
    for n in (5000, 4096, 1024 * 3 + 1):
        x = make_blocks(oracle, n, kind, 21)
        want = oracle.transform_bc7(x)
        for cuts in ([], [2048], [1024, 2048, 3072], [1024], [3072]):
            parts = partition(n, [c for c in cuts if c < n])
            got = np.full(x.size, 0xEE, dtype=np.uint8)
            covered = np.zeros(x.size, dtype=np.int32)
            for f, c in parts:
                local = oracle.transform_bc7(x[16 * f: 16 * (f + c)])
                g, l, b = bc7.shard_pieces(n, f, c)
                assert sum(b) == 16 * c
                for p in range(9):
                    got[g[p]: g[p] + b[p]] = local[l[p]: l[p] + b[p]]
                    covered[g[p]: g[p] + b[p]] += 1
            assert (covered == 1).all(), (kind, n, cuts)           # the pieces tile the buffer exactly once
            assert np.array_equal(got, want), (kind, n, cuts)


def test_placement_table_rejects_unaligned_shards(pkg, bc7):
    g, l, b = bc7.shard_pieces(3000, 1024, 1024)
    assert b[:8] == [8192, 2048, 1024, 1024, 1024, 1024, 1024, 1024] and b[8] == 0
    assert g[0] == 8 * 1024 and g[1] == 8 * 2048 + 2 * 1024 and g[7] == 15 * 2048 + 1024 and l[1] == 8 * 1024
    g, l, b = bc7.shard_pieces(3000, 2048, 952)          # the shard that reaches the end: only the tail part
    assert b[:8] == [0] * 8 and b[8] == 16 * 952 and g[8] == 16 * 2048 and l[8] == 0
    with pytest.raises(pkg.DeviceError):                     # starts inside a granule
        bc7.shard_pieces(3000, 100, 1024)
    with pytest.raises(pkg.DeviceError):                     # ends inside a granule, not at the end
        bc7.shard_pieces(3000, 0, 1500)
    with pytest.raises(pkg.DeviceError):
        bc7.shard_pieces(3000, 2048, 2000)


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
