import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "gpu_slow: full-length legs of the GPU parity tests (-m gpu_slow on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """the `gpu_slow` legs run only when asked for by name (-m gpu_slow): neither the CPU suite (-m "not gpu") nor -m gpu takes them"""
    if "gpu_slow" in (config.getoption("-m") or ""):
        return
    skip = pytest.mark.skip(reason="full-length leg: run with -m gpu_slow on the GPU box")
    for item in items:
        if "gpu_slow" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def mano_model():
    from homan_amd.mano_assets import synthetic_mano
    return synthetic_mano(0)


@pytest.fixture(autouse=True)
def _release_device_objects():
    """hipGraphs, their private memory pools and the steppers' buffers go when the test's objects do: collect them right
    after every test instead of whenever the cycle collector runs (a process that piles up several dozen captured graphs
    has crashed inside the HIP runtime at a later replay)."""
    yield
    import gc
    try:
        import torch
        gpu = torch.cuda.is_available()
    except ImportError:
        gpu = False
    if gpu:
        torch.cuda.synchronize()         # (nothing of the test is in flight when its graphs are destroyed)
    gc.collect()
    if gpu:
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
