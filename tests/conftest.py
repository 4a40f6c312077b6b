import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_required():
    if not _have_gpu():
        pytest.fail("this test is marked gpu but no HIP device is visible")


def cloth_params(dimx, dimz, pos=(0.0, -0.1, 0.0), stiff=(0.9, 0.9, 0.9), mass=0.5, flip=0):
    """scene_params[19] in the layout of flex_utils.py:332-342."""
    import numpy as np

    return np.array([pos[0], pos[1], pos[2], dimx, dimz, stiff[0], stiff[1], stiff[2], 2,
                     0, 2, 0, np.pi / 2, -np.pi / 2, 0, 720, 720, mass, flip], dtype=np.float64)
