import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: minutes of meta-training on the GPU (skipped unless L2O_RUN_SLOW=1)")


def pytest_collection_modifyitems(config, items):
    """GPU tests must not silently pass on a box without a GPU: skip them here
    unless a device is visible (the driver runs `-m "not gpu"` on CPU and
    `-m gpu` on the MI355X box)."""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
