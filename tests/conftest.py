import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need the MI355X: without one they are skipped (a plain `pytest tests` on a CPU box stays green) — on the
    GPU box they run, and the product itself still fails loudly there if its HIP library is missing."""
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no GPU visible (gpu-marked tests run on the MI355X box)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def gold():
    return golden


def rel_rms(a, b):
    a = np.asarray(a, dtype=np.float64 if not np.iscomplexobj(a) else np.complex128)
    b = np.asarray(b, dtype=a.dtype)
    return float(np.sqrt(np.mean(np.abs(a - b) ** 2)) / (np.sqrt(np.mean(np.abs(b) ** 2)) + 1e-300))


def rms_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)))
