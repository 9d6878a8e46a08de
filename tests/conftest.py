import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def _gpu_available() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must FAIL loudly rather than silently pass: only skip when the
    # user did not ask for gpu tests explicitly.
    if _gpu_available():
        return
    asked = "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or "")
    if asked:
        return
    skip = pytest.mark.skip(reason="no GPU in this container (run with gpurun)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture
def fast_mode(monkeypatch):
    """The library's default is the reference-order ("exact") arithmetic (fl_default_exact() == 1).  Modules that characterise the
    FAST kernels -- their tile configurations, fused forms and tolerances -- opt in with this fixture: models created inside start
    in fast mode (FL_FAST=1) and the operator-level entry points run the fast kernels (fl_set_op_mode(0))."""
    from fastllama_amd import hip
    L = hip.load()
    monkeypatch.setenv("FL_FAST", "1")
    monkeypatch.delenv("FL_EXACT", raising=False)
    L.fl_set_op_mode(0)
    yield
    L.fl_set_op_mode(-1)
