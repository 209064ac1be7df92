import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "qqq_golden.npz"))


def gpu_dump(name, **arrays):
    """Save diagnostics under gpurun_out/ (merged back from the GPU box) so a failing GPU test can be
    analysed offline."""
    import numpy as np

    d = os.path.join(ROOT, "gpurun_out", "diag")
    os.makedirs(d, exist_ok=True)
    np.savez_compressed(os.path.join(d, name + ".npz"), **arrays)


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")
