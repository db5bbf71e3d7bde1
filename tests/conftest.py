import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _tune_block_of_the_test(request):
    """Performance knobs are no longer process state (round 5): a test that forces a kernel family does it in the block bound to its own
    thread (teochat_amd._lib.tune_set).  That block is released when the test ends -- scoping, not a global reset."""
    yield
    if "gpu" in request.keywords and _has_gpu():
        from teochat_amd import _lib
        _lib.tune_release()
