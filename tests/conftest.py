import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def pytest_sessionstart(session):
    """GPU runs: bring PyTorch's bundled HIP runtime up first (see coverm_amd.native._torch_hip_first); tests that hand
    torch device tensors to cov_push_batch_device would otherwise find "No HIP GPUs" once libcovermhip.so has loaded
    the system runtime.  A no-op on the CPU-only container."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
