"""A small end-to-end use of the NDArray surface the way a NumPower script would write it: multinomial
logistic regression by full-batch gradient descent, every array resident on the MI355X.

    logits = X . W                       (samples x 784) . (784 x 10): the thin-product MFMA kernel
    P      = softmax(logits)             row max, exp(logits - max), row sum, divide: fused chains with
                                         axis ends — two passes over `logits` instead of five
    grad   = X^T . (P - Y) / n           (784 x samples) . (samples x 10): split-K planner
    W     -= lr * grad

`python examples/softmax_regression.py [samples] [steps]` prints the loss curve, the time per step and the
same training run in numpy (fp64) beside it.  tests/test_gpu_examples.py runs a small instance as a
composite parity test."""
from __future__ import annotations

import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

from numpower_amd import synth                 # noqa: E402
from numpower_amd.lazy import Lazy             # noqa: E402,F401  (installs NDArray.lazy)
from numpower_amd.ndarray import NDArray as nd  # noqa: E402


def make_problem(samples: int, features: int = 784, classes: int = 10, seed: int = 1):
    X = synth.uniform((samples, features), seed, -1.0, 1.0)
    W_true = synth.uniform((features, classes), seed + 1, -1.0, 1.0)
    labels = np.argmax(X.astype(np.float64) @ W_true.astype(np.float64), axis=1)
    Y = np.zeros((samples, classes), np.float32)
    Y[np.arange(samples), labels] = 1.0
    return X, Y


def place(X, Y):
    """$x->gpu(): the arrays go to the device once."""
    gX, gY = nd.array(X).gpu(), nd.array(Y).gpu()
    return gX, gY, nd.transpose(gX)                          # X^T once: the gradient product reads it


def train_gpu(X, Y, steps: int, lr: float, placed=None):
    n, classes = Y.shape
    gX, gY, gXt = placed if placed is not None else place(X, Y)
    W = nd.zeros([X.shape[1], classes]).gpu()
    losses = []
    for _ in range(steps):
        logits = nd.matmul(gX, W)
        row_max = nd.reshape(nd.max(logits, 1), [n, 1])                       # (n, 1) column operand
        row_sum = nd.reshape((logits.lazy() - row_max).exp().sum(axis=1), [n, 1])   # ONE pass: exp(l - max) summed per row
        P = ((logits.lazy() - row_max).exp() / row_sum).eval()               # ONE pass: the probabilities
        # cross-entropy of the true class: -sum(Y * log P) / n, the product never materialised
        losses.append(-(P.lazy().log() * gY).sum() / n)
        grad = nd.matmul(gXt, P - gY)
        W = W - grad * (lr / n)
    return W.cpu().numpy(), losses


def train_numpy(X, Y, steps: int, lr: float):
    X64, Y64 = X.astype(np.float64), Y.astype(np.float64)
    n = X.shape[0]
    W = np.zeros((X.shape[1], Y.shape[1]))
    losses = []
    for _ in range(steps):
        logits = X64 @ W
        e = np.exp(logits - logits.max(axis=1, keepdims=True))
        P = e / e.sum(axis=1, keepdims=True)
        losses.append(float(-(np.log(P) * Y64).sum() / n))
        W = W - (lr / n) * (X64.T @ (P - Y64))
    return W, losses


def main():
    samples = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    X, Y = make_problem(samples)
    train_gpu(X[:4096], Y[:4096], 2, 0.5)                    # warm the kernels up
    t0 = time.perf_counter()
    placed = place(X, Y)
    t_place = time.perf_counter() - t0
    t0 = time.perf_counter()
    W, losses = train_gpu(X, Y, steps, 0.5, placed)          # ends with W->cpu(): the stream is drained
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    W_ref, ref_losses = train_numpy(X, Y, min(steps, 3), 0.5)
    t_cpu = (time.perf_counter() - t0) / min(steps, 3)
    print("samples %d, features %d, classes %d" % (samples, X.shape[1], Y.shape[1]))
    print("loss: " + " ".join("%.4f" % v for v in losses))
    print("numpy fp64 first steps: " + " ".join("%.4f" % v for v in ref_losses))
    print("placement (H2D of %.0f MB + one transpose): %.0f ms   GPU %.2f ms / step   numpy fp64 %.0f ms / step"
          % ((X.nbytes + Y.nbytes) / 1e6, 1e3 * t_place, 1e3 * t_gpu / steps, 1e3 * t_cpu))


if __name__ == "__main__":
    main()
