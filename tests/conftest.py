import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_ctx():
    """A libkiwigpu context on GPU 0.  No fallback: if the library or the device is
    missing on a GPU run, the tests FAIL (they are only selected with -m gpu)."""
    from flydog_sdr_gps_amd import Context
    ctx = Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def oracle():
    from oracle import kiwi_oracle
    kiwi_oracle.lib()
    return kiwi_oracle
