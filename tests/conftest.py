import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    def _load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    return _load


def rel_err(a, ref):
    """max |a-ref| / max |ref|  -- the tolerance metric used throughout
    (element-wise relative error is ill-conditioned for near-zero correlation
    outputs; SURVEY.md section 7 'hard parts' #1)."""
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    denom = np.abs(ref).max()
    return float(np.abs(a - ref).max() / (denom if denom > 0 else 1.0))


def l2_err(a, ref):
    """||a-ref||_2 / ||ref||_2 -- for gradients that pass through the warp: a last-bit change
    of a flow value can flip a floor() in the bilinear taps and move a handful of pixels by
    O(1e-3) of the maximum, so the max-norm is not a stable yardstick there."""
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    den = np.sqrt((ref * ref).sum())
    return float(np.sqrt(((a - ref) ** 2).sum()) / (den if den > 0 else 1.0))
