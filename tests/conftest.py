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
def cuda_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu and needs a GPU; none is visible")
    return torch.device("cuda:0")


@pytest.fixture
def pass_sequencing():
    """Kernel-level tests that record / intercept the individual pass launches (``train.DEBUG_STASH``, monkeypatched
    ``ops.*``) run the step pass by pass from Python instead of through the fused per-direction entries — the two are
    bit-identical (tests/test_gpu_fused.py)."""
    from pointnetgpd_amd import train
    train.set_sequencing("passes")
    yield
    train.set_sequencing("fused")
