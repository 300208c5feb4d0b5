import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _device_runtime_first():
    """On a GPU box: torch's device runtime and allocator come up before the first library call of the session, whichever
    test file runs first (the order every full run of the suite has; profiles/r03/README.md, "open observation")."""
    if os.environ.get("GDX_TEST_NO_RUNTIME_FIRST") == "1":  # tools/stall_probe.sh: the library first, torch whenever a test wants it
        yield
        return
    try:
        import torch
    except ImportError:
        yield
        return
    if torch.cuda.is_available():
        torch.zeros(1, device="cuda")
        torch.cuda.synchronize()
    yield


@pytest.fixture(scope="session")
def kat():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "genedex_kat.json")) as f:
        return json.load(f)
