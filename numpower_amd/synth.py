"""Deterministic synthetic fp32 inputs: a counter-based generator so that the
build container, the GPU box and every rank produce identical data without shipping fixtures.

    value(seed, i) = lo + (hi - lo) * (splitmix64(seed * 2^32 + i) >> 40) / 2^24

i.e. 24 random mantissa bits per element, uniform on [lo, hi).  numpy only (host-side data
generation is not part of the measured path).
"""
from __future__ import annotations

import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GAMMA = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def uniform(shape, seed: int, lo: float = 0.0, hi: float = 1.0, chunk: int = 1 << 24) -> np.ndarray:
    """fp32 array of `shape`, element i (C order) depends only on (seed, i)."""
    n = int(np.prod(shape, dtype=np.int64))
    out = np.empty(n, dtype=np.float32)
    base = np.uint64(seed) << np.uint64(32)
    scale = np.float32((hi - lo) / float(1 << 24))
    for start in range(0, n, chunk):
        stop = min(start + chunk, n)
        idx = np.arange(start, stop, dtype=np.uint64) + base
        bits = (splitmix64(idx) >> np.uint64(40)).astype(np.float32)
        out[start:stop] = np.float32(lo) + bits * scale
    return out.reshape(shape)


def uniform_many(shape, seeds, lo: float = 0.0, hi: float = 1.0, threads: int | None = None):
    """uniform(shape, seed, lo, hi) for every seed, generated on a thread pool (numpy's ufuncs release the GIL) and yielded
    in order — the 2 x 512 matrices of BASELINE config 5 are 10^9 elements, minutes on one core."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    seeds = list(seeds)
    workers = max(1, min(threads or (os.cpu_count() or 1), 64, len(seeds)))
    with ThreadPoolExecutor(workers) as pool:
        window = []
        for sd in seeds:
            window.append(pool.submit(uniform, shape, sd, lo, hi))
            if len(window) >= 2 * workers:
                yield window.pop(0).result()
        for f in window:
            yield f.result()
