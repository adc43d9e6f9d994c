"""The `_device` entry points of the block transforms, normalisation, colour-array operations and decoders only enqueue
kernels on the caller's stream (no allocation, no synchronisation, no host-side state), so a caller can capture them
into a HIP graph and replay it -- the launch-bound case of many small textures per frame.  BC7 too since version 1 of
its format (one pass, no workspace, no host-side totals).  (The batch call plans on the host and is documented as not
capturable.)"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")


@pytest.mark.gpu
def test_gpu_device_entry_points_replay_from_a_hip_graph(pkg, oracle):
    from dxt_lossless_transform_amd import bc7, color565, decode, normalize

    dev = torch.device("cuda:0")
    n = 4096 + 3
    x1 = torch.from_numpy(oracle.generate_test_data("bc1", n)).to(dev)
    x3 = torch.from_numpy(oracle.generate_test_data("bc3", n)).to(dev)
    y1, z1 = torch.zeros_like(x1), torch.zeros_like(x1)
    y3, z3 = torch.zeros_like(x3), torch.zeros_like(x3)
    norm1 = torch.zeros_like(x1)
    cols = torch.zeros(4 * n, dtype=torch.uint8, device=dev)
    pixels = torch.zeros(64 * n, dtype=torch.uint8, device=dev)
    s3 = pkg.Bc3TransformSettings(pkg.YCoCgVariant.Variant2, True, False)
    n7 = 5 * 1024 + 77                            # main part + tail part: two launches per direction
    x7 = torch.from_numpy(np.random.default_rng(7).integers(0, 256, 16 * n7, dtype=np.uint8)).to(dev)
    y7, z7 = torch.zeros_like(x7), torch.zeros_like(x7)

    def work():
        pkg.transform_bc1_with_settings(x1, y1)
        pkg.untransform_bc1_with_settings(y1, z1)
        pkg.transform_bc3_with_settings(x3, y3, s3)
        pkg.untransform_bc3_with_settings(y3, z3, s3)
        normalize.normalize_blocks(x1, norm1, normalize.ColorNormalizationMode.COLOR0_ONLY)
        color565.recorrelate_ycocg_r(y1[:4 * n], cols, 1)
        decode.decode_blocks("bc1", z1, pixels)
        bc7.transform_bc7(x7, y7)
        bc7.untransform_bc7(y7, z7)

    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        work()                                    # warm-up outside capture (module load, first launch)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)

    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        work()
    # new inputs in the same buffers, outputs cleared: only a replay can produce the right answers now
    fresh1 = np.random.default_rng(1).integers(0, 256, 8 * n, dtype=np.uint8)
    fresh3 = np.random.default_rng(3).integers(0, 256, 16 * n, dtype=np.uint8)
    x1.copy_(torch.from_numpy(fresh1))
    x3.copy_(torch.from_numpy(fresh3))
    fresh7 = np.random.default_rng(77).integers(0, 256, 16 * n7, dtype=np.uint8)
    x7.copy_(torch.from_numpy(fresh7))
    for t in (y1, z1, y3, z3, norm1, cols, pixels, y7, z7):
        t.zero_()
    graph.replay()
    torch.cuda.synchronize(dev)

    want1 = oracle.transform("bc1", fresh1, 1, True)
    assert np.array_equal(y1.cpu().numpy(), want1)
    assert np.array_equal(z1.cpu().numpy(), fresh1)
    assert np.array_equal(y3.cpu().numpy(), oracle.transform("bc3", fresh3, 2, False, True))
    assert np.array_equal(z3.cpu().numpy(), fresh3)
    assert np.array_equal(norm1.cpu().numpy(), oracle.normalize_bc1_blocks(fresh1, 1))
    # colours of the transformed buffer, recorrelated = the split endpoints of the source
    assert np.array_equal(cols.cpu().numpy(), oracle.split_565_color_endpoints(fresh1.reshape(n, 8)[:, :4].reshape(-1)))
    assert np.array_equal(pixels.cpu().numpy(), oracle.decode_blocks("bc1", fresh1))
    assert np.array_equal(y7.cpu().numpy(), oracle.transform_bc7(fresh7))
    assert np.array_equal(z7.cpu().numpy(), fresh7)
