import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (this container only)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    has_gpu = _has_gpu()
    has_ref = os.path.isdir("/root/reference/model")
    for item in items:
        if "gpu" in item.keywords and not has_gpu:
            item.add_marker(pytest.mark.skip(reason="no GPU in this container"))
        if "reference" in item.keywords and not has_ref:
            item.add_marker(pytest.mark.skip(reason="/root/reference not present"))


@pytest.fixture(autouse=True)
def _seed_torch(request):
    """Every test starts from the same torch generator state (module initialisers draw from it): a test's data must not
    depend on which tests ran before it (round 3: an 'intermittent' gradient mismatch was a weight draw that put one
    LeakyReLU pre-activation within summation noise of 0 -- it came and went with the suite's composition)."""
    try:
        import torch
        torch.manual_seed(0x5EED)
    except Exception:
        pass
    yield
