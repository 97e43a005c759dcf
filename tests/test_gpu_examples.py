"""Composite parity: examples/softmax_regression.py (thin-product and split-K GEMMs, column-operand fused
chains with axis ends, full chain reductions, eager binaries) against the same training run in numpy fp64."""
import importlib.util
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _load(name="softmax_regression"):
    path = Path(__file__).resolve().parent.parent / "examples" / (name + ".py")
    spec = importlib.util.spec_from_file_location(name, path)
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


@pytest.mark.parametrize("n,d,k", [(20_000, 16, 8), (50_000, 64, 32), (7001, 3, 5)])
def test_kmeans_tracks_numpy(n, d, k, hip):
    """examples/kmeans.py: thin GEMM + a chain with a column AND a row operand + argmin over short rows + a one-hot
    built by a broadcast compare + split-K GEMM + column divide, against numpy fp64.  A point within fp32 rounding of
    two centroids may be assigned differently (and Lloyd's iteration amplifies a flip), so: one iteration is checked
    exactly — label agreement, and centroids equal to the fp64 means of the GPU's own labels — and a longer run by its
    inertia."""
    ex = _load("kmeans")
    X, _ = ex.make_points(n, d, k, seed=11)
    C0 = X[:k].copy()
    C1, labels = ex.kmeans_gpu(X, C0, 1)
    _, labels_ref = ex.kmeans_numpy(X, C0, 1)
    labels = labels.astype(np.int64)
    assert (labels == labels_ref).mean() >= 0.999
    H = np.eye(k)[labels]
    means = (H.T @ X.astype(np.float64)) / np.maximum(H.sum(0), 1.0)[:, None]
    assert np.abs(C1 - means).max() <= 1e-5 * max(1.0, np.abs(means).max())

    def inertia(C):
        Xd, Cd = X.astype(np.float64), C.astype(np.float64)
        D = (Xd * Xd).sum(1)[:, None] - 2.0 * Xd @ Cd.T + (Cd * Cd).sum(1)[None, :]
        return D.min(1).sum()

    C5, _ = ex.kmeans_gpu(X, C0, 5)
    C5_ref, _ = ex.kmeans_numpy(X, C0, 5)
    assert np.isfinite(C5).all() and C5.shape == (k, d)
    assert abs(inertia(C5) - inertia(C5_ref)) <= 2e-3 * inertia(C5_ref)
