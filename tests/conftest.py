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


def ocr_oracle(max_steps=8, dtype=None, **kw):
    """the oracle's OWN implementation of the frozen OCR network (oracle/ref_ocr.py), loaded with the frozen synthetic weights of
    the product's network as DATA (a state_dict, as a checkpoint would be handed to two implementations).  float64 by default."""
    import torch
    from oracle.ref_ocr import OcrOracle
    from textboxgan_amd.aster import AsterLikeOCR
    return OcrOracle(AsterLikeOCR(max_steps=max_steps, **kw).state_dict(), max_steps=max_steps,
                     dtype=torch.float64 if dtype is None else dtype)


@pytest.fixture
def arith(request):
    """arithmetic of the MFMA contractions for this test (ops.compute_dtype scope): "f32" unless parametrised."""
    from textboxgan_amd import ops

    mode = getattr(request, "param", "f32")
    with ops.compute_dtype(mode):
        yield mode


def arith_modes(fn):
    """run an fp32 parity test in BOTH fp32 arithmetics -- exact v_mfma_f32_32x32x2_f32 and "f32x3" (three bf16 terms per
    operand on the bf16 pipe) -- at the SAME tolerance (the split path may carry the fp32
    configuration only if every fp32 test passes at the unchanged fp32 tolerances)."""
    return pytest.mark.usefixtures("arith")(pytest.mark.parametrize("arith", ["f32", "f32x3"], indirect=True)(fn))
