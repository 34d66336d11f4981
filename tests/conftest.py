import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def dev():
    """cuda:0 for the ``-m gpu`` tests.  On a box without an AMD GPU device node (this development
    container) the GPU tests are skipped; on a GPU box a card that torch or the library cannot use
    is a failure, never a silent skip."""
    import torch

    if not torch.cuda.is_available():
        if not os.path.exists("/dev/kfd"):
            pytest.skip("no GPU on this box (run with -m gpu on the MI355X)")
        raise AssertionError("/dev/kfd exists but torch sees no GPU")
    from nfst_amd import _lib

    assert _lib.lib.nfst_device_available() == 1, "libnfst_hip.so sees no GPU"
    return torch.device("cuda:0")
