import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_available() -> bool:
    if os.environ.get("FYX_TEST_NO_TORCH"):      # sanitizer runs: torch's own HIP start-up does not survive a preloaded ASan
        return os.path.exists("/dev/kfd")
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _gpu_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "fyrox_unit_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def ctx():
    """A live fyx context on GPU 0 (gpu tests only)."""
    import fyrox_amd
    c = fyrox_amd.Context(0)
    yield c
    c.close()
