import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

from yololite_amd import _lib as _yl_lib  # noqa: E402

_yl_lib.apply_runtime_knobs()  # before any test touches the GPU (see _lib.RUNTIME_KNOBS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return np.load(ROOT / "tests" / "golden" / "golden_v1.npz")


@pytest.fixture(scope="session")
def gpu_device():
    import torch

    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu but no GPU is visible")
    return torch.device("cuda:0")
