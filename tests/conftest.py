import os
import sys

os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")   # before anything loads libgomp: the oracle's workers must not spin
# the dot metric's quantised flow is taken whatever the list-size skew (the library's default too; search_ms.hip: mscan_dot_ready has a guard for
# A/B runs -- the exact pair scan keeps its coverage through LANCE_HIP_NO_DOT_FLOW=1 children)
os.environ.setdefault("LANCE_HIP_DOT_FLOW_SKEW", "1e18")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a test that hangs (a rendezvous that never completes, a wedged child process) must fail, not stall the whole run:
    # pytest-timeout, when installed and no --timeout was given, limits every test to ten minutes
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 600


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def engine():
    """The HIP engine; GPU tests fail loudly (never skip silently) if it cannot load."""
    import lance_amd
    return lance_amd
