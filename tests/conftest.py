import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _no_gpu_here() -> bool:
    """True only on a box that provably has no ROCm device (no /dev/kfd and the built engine reports 0 devices).  On a GPU
    box nothing is ever skipped: a missing or broken engine library must fail the gpu tests loudly."""
    if os.path.exists("/dev/kfd"):
        return False
    try:
        from tardis_amd import _lib

        return _lib.lib().tardis_mc_device_count() == 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    gpu_items = [it for it in items if "gpu" in it.keywords]
    if gpu_items and _no_gpu_here():
        skip = pytest.mark.skip(reason="no ROCm device on this box (gpu-marked tests run on the MI355X box)")
        for it in gpu_items:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o

    o.build()
    return o
