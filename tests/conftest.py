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
