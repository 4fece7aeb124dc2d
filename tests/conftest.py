import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


GPU_TEST_TIMEOUT_S = 300   # the whole GPU suite takes ~50 s; a test that sits this long is a hung kernel / stream


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """A hung GPU test must end the run, not sit on the box until the scheduler's limit: every `gpu` test gets a
    pytest-timeout watchdog (thread method: the process is blocked inside a HIP call when a stream hangs, where a signal
    handler never gets to run; the watchdog thread dumps all stacks and exits the process)."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
            item.add_marker(pytest.mark.timeout(GPU_TEST_TIMEOUT_S, method="thread"))


@pytest.fixture(scope="session")
def gpu():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
