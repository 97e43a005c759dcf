"""Composite parity: examples/softmax_regression.py (thin-product and split-K GEMMs, column-operand fused
chains with axis ends, full chain reductions, eager binaries) against the same training run in numpy fp64."""
import importlib.util
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _load():
    path = Path(__file__).resolve().parent.parent / "examples" / "softmax_regression.py"
    spec = importlib.util.spec_from_file_location("softmax_regression", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("samples,features", [(5000, 784), (3001, 50), (2048, 128)])
def test_softmax_regression_tracks_numpy(samples, features, hip):
    ex = _load()
    X, Y = ex.make_problem(samples, features, 10, seed=7)
    W, losses = ex.train_gpu(X, Y, 8, 0.5)
    W_ref, ref_losses = ex.train_numpy(X, Y, 8, 0.5)
    assert np.allclose(losses, ref_losses, rtol=2e-5, atol=1e-6), (losses, ref_losses)
    assert losses[-1] < losses[0]
    scale = np.abs(W_ref).max()
    assert np.abs(W - W_ref).max() <= 1e-4 * scale
