import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # pytest.ini's hard 40-minute limit per test is pytest-timeout's (tests/requirements.txt): without the plugin the option
    # is only an "unknown config option" warning and a stuck test waits for the box's own limit -- say so where it is seen
    if not config.pluginmanager.hasplugin("timeout"):
        import warnings

        warnings.warn("pytest-timeout is not installed: pytest.ini's `timeout = 2400` is not enforced "
                      "(faulthandler_timeout still dumps every thread's traceback after 15 minutes)")


@pytest.fixture(scope="session")
def kat():
    import json

    with open(os.path.join(ROOT, "tests", "golden", "genedex_kat.json")) as f:
        return json.load(f)
