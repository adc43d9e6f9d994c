import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "side_build: needs a GPU AND the -DDXTLT_EXPERIMENTS side build loaded through DXTLT_LIB_PATH; "
                                       "selected only by name (-m side_build), never by -m gpu or -m 'not gpu'")


def pytest_collection_modifyitems(config, items):
    """`side_build` tests exercise kernels the shipped library does not contain: they run only when asked for by name."""
    if "side_build" in (config.option.markexpr or ""):
        return
    keep, drop = [], []
    for item in items:
        (drop if item.get_closest_marker("side_build") else keep).append(item)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_c

    oracle_c.lib()
    return oracle_c


@pytest.fixture(scope="session")
def pkg():
    import dxt_lossless_transform_amd as p

    p.build()  # no-op when the in-tree .so is current
    p.load()
    return p
