import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """On a GPU box initialise torch's HIP context before the first test touches the device through the C ABI: torch's lazy
    initialisation later in a process whose HIP runtime is already busy has been seen to report "No HIP GPUs are available"
    (tests that use torch tensors after ctypes-only tests).  device_count() does not initialise anything, so CPU runs are unaffected."""
    try:
        import torch
        if torch.cuda.device_count() > 0:
            torch.cuda.init()
    except Exception:   # no torch / no driver: the gpu-marked tests will say so themselves
        pass
