import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: diagnostic reproducers that need extra builds (run with S2T_SLOW=1; tools/ runs them)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when no device is visible, e.g. a plain `pytest tests/` here.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if not os.environ.get("S2T_SLOW"):
        slow = pytest.mark.skip(reason="slow diagnostic (S2T_SLOW=1 runs it)")
        for it in items:
            if "slow" in it.keywords:
                it.add_marker(slow)
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
