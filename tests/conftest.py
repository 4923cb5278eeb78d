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
def golden_blocks():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "blocks.npz"))


@pytest.fixture(scope="session")
def golden_full():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "full_640x360.npz"))


@pytest.fixture(autouse=True)
def _seed_global_rng():
    """Modules built inside tests (nn.Conv2d defaults etc.) draw their initial weights from torch's global
    generator: seed it so that every run of a test sees the same numbers.  (The one failure this once hid is
    explained and covered in tests/test_gpu_training.py::test_conv_bn_act_backward: a ReLU decision within
    rounding distance of zero.)"""
    import torch
    torch.manual_seed(20240917)
    yield
