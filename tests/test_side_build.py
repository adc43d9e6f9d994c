"""Tests of the EXPERIMENTS side build (-DDXTLT_EXPERIMENTS: csrc/bcn_experiments.h), which is never the shipped library.

Marked `side_build` and deselected by tests/conftest.py unless asked for by name, so that neither `-m gpu` nor `-m "not gpu"` carries
dead skips for them.  To run them on a GPU box:
    DXTLT_EXTRA_HIPCC_FLAGS=-DDXTLT_EXPERIMENTS tools/ab_build_rev.sh WORKTREE exp
    DXTLT_LIB_PATH=$PWD/ab/libdxtlt_exp.so python -m pytest tests/test_side_build.py -m side_build"""
import numpy as np
import pytest

from helpers import BLOCK, FORMATS, all_settings, settings_id
from test_gpu_parity import TILE, dev, fwd_oracle, run_device  # noqa: F401  (dev: the module's device fixture)

pytestmark = pytest.mark.side_build


@pytest.mark.parametrize("fmt", FORMATS)
def test_element_kernel_equals_tiled_kernel(pkg, oracle, dev, fmt):
    if not pkg.tuning_mask() & 1:
        pytest.skip("the element-granular kernel exists in the experiments side build only (-DDXTLT_EXPERIMENTS)")
    n = 4 * TILE[fmt]
    x = oracle.fill_splitmix64(n * BLOCK[fmt], 0xE1E)
    for s in all_settings(fmt):
        tiled = run_device(pkg, fmt, x, s, dev)
        try:
            pkg.set_tuning(0, 1)
            generic = run_device(pkg, fmt, x, s, dev)
            back = run_device(pkg, fmt, generic, s, dev, inverse=True)
        finally:
            pkg.set_tuning(0, 0)
        assert np.array_equal(tiled, generic), (fmt, settings_id(s))
        assert np.array_equal(back, x)
