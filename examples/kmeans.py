"""Lloyd's k-means on the NDArray surface, every array resident on the MI355X:

    D      = |x|^2 - 2 X . C^T + |c|^2       thin-product GEMM, then ONE fused chain with a column and a row operand
    labels = argmin(D, axis 1)               lane-group argreduce over short rows
    H      = (0 + labels_col) == arange_row  one-hot assignment matrix, one fused chain
    C      = (H^T . X) / counts_col          split-K GEMM (k x n . n x d), column-operand divide

`python examples/kmeans.py [points] [dims] [k] [iterations]`; tests/test_gpu_examples.py checks a small instance
against the same iterations in numpy fp64."""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from numpower_amd import synth                 # noqa: E402
from numpower_amd.lazy import Lazy             # noqa: E402,F401
from numpower_amd.ndarray import GPU, NDArray as nd  # noqa: E402


def make_points(n: int, d: int, k: int, seed: int = 3):
    centres = synth.uniform((k, d), seed, -4.0, 4.0)
    which = (synth.uniform((n,), seed + 1, 0.0, 1.0) * k).astype(np.int64) % k
    return (centres[which] + synth.uniform((n, d), seed + 2, -0.7, 0.7)).astype(np.float32), centres


def kmeans_gpu(X, C0, iterations: int):
    n, d = X.shape
    k = C0.shape[0]
    gX = nd.array(X).gpu()
    xn = nd.reshape((gX.lazy() * gX).sum(axis=1), [n, 1])                 # |x|^2 once, as a column
    ids = nd.arange(k, device=GPU)                                         # 0 .. k-1 as a row operand
    Z = nd.zeros([n, k]).gpu()
    C = nd.array(C0).gpu()
    labels = None
    for _ in range(iterations):
        cn = (C.lazy() * C).sum(axis=1)                                    # (k,) row operand
        G = nd.matmul(gX, nd.transpose(C))                                 # (n, k)
        D = ((G.lazy() * -2.0 + xn) + cn).eval()
        labels = nd.argmin(D, 1)
        H = (Z.lazy() + nd.reshape(labels, [n, 1])).equal(ids).eval()      # one-hot (n, k)
        counts = nd.reshape(nd.sum(H, 0), [k, 1])
        sums = nd.matmul(nd.transpose(H), gX)                              # (k, d)
        C = sums / nd.maximum(counts, nd.array(np.float32(1.0)).gpu())     # an empty cluster keeps a zero centroid
    return C.cpu().numpy(), labels.cpu().numpy()


def kmeans_numpy(X, C0, iterations: int):
    X = X.astype(np.float64)
    C = C0.astype(np.float64)
    labels = None
    for _ in range(iterations):
        D = (X * X).sum(1)[:, None] - 2.0 * X @ C.T + (C * C).sum(1)[None, :]
        labels = D.argmin(1)
        H = np.eye(C.shape[0])[labels]
        C = (H.T @ X) / np.maximum(H.sum(0), 1.0)[:, None]
    return C, labels


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    its = int(sys.argv[4]) if len(sys.argv) > 4 else 10
    X, centres = make_points(n, d, k)
    C0 = X[:k].copy()
    kmeans_gpu(X[:8192], C0, 2)
    t0 = time.perf_counter()
    C, labels = kmeans_gpu(X, C0, its)
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    kmeans_numpy(X, C0, 1)
    t_cpu = time.perf_counter() - t0
    print("%d points, %d dims, k = %d: %d iterations in %.1f ms on the GPU (incl. placing X: %.0f MB); numpy fp64: %.0f ms per iteration"
          % (n, d, k, its, 1e3 * t_gpu, X.nbytes / 1e6, 1e3 * t_cpu))


if __name__ == "__main__":
    main()
