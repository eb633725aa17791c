import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


def _gpu_present():
    try:
        from orb_slam2_ssd_semantic_amd import _ffi
        return _ffi.lib().orbfe_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # gpu-marked tests are deselected with -m "not gpu"; if someone runs them on a box without a GPU they
    # must FAIL (not skip): a silent fallback is exactly what the product must never have.
    pass


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_ffi
    oracle_ffi.lib()
    return oracle_ffi


@pytest.fixture(scope="session")
def have_gpu():
    return _gpu_present()


@pytest.fixture(scope="session")
def dev_lib():
    """The -DORBFE_DEVELOPER build of liborbfe (ab/liborbfe_dev.so, built by __graft_entry__.build()): the release library plus
    the measured-slower kernel variants (k_pyr_walk2, k_fast_pyr, k_blur_pyr) that the release build compiles out."""
    from orb_slam2_ssd_semantic_amd import _build, _ffi
    path = os.path.join(ROOT, "ab", "liborbfe_dev.so")
    if not os.path.exists(path):
        _build.build_variant("dev", ["-DORBFE_DEVELOPER"])
    return _ffi.load_variant(path)
