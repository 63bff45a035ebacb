import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def load_golden(name):
    """Golden vector -> dict of numpy arrays; float16 inputs are widened to float32 (exact)."""
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        out = {}
        for k in z.files:
            v = z[k]
            out[k] = v.astype(np.float32) if v.dtype == np.float16 else v
        return out


@pytest.fixture(scope="session")
def golden():
    return load_golden


@pytest.fixture(autouse=True)
def _inference_mode_for_gpu_tests(request):
    """The mirrors are inference-only and raise when autograd is recording (ops.inference_only); the GPU parity tests exercise
    them the way the reference's eval loop does, under torch.no_grad() (eval_manager_mm.py:195)."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    import torch
    with torch.no_grad():
        yield
