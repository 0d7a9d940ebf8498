import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    import torch

    assert torch.cuda.is_available(), "gpu-marked test run without a GPU"
    from textboxgan_amd import native

    native.lib()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")
