"""CPU: the oracle's restatement of calculate_median (arithmetics.c:111-138) and calculate_quantile
(statistics.c:14-50) against an independent numpy formulation.  The reference holds no PHPT for
median / quantile (tests/ has none): this path is pinned by the formulas only."""
import numpy as np

from numpower_amd import synth


def test_median_matches_sorted_middle(oracle):
    for n in (1, 2, 3, 4, 5, 6, 99, 100, 1001, 4096):
        x = synth.uniform((n,), 40 + n, -50.0, 50.0)
        s = np.sort(x)
        want = s[n // 2] if n % 2 else np.float32((s[n // 2 - 1] + s[n // 2]) / np.float32(2.0))
        got, stats = oracle.median(x, with_stats=True)
        assert got.view(np.uint32) == np.float32(want).view(np.uint32), n
        assert stats[1] == s[n // 2]


def test_quantile_matches_the_interpolation_formula(oracle):
    for n in (1, 2, 3, 10, 1000, 1001):
        x = synth.uniform((n,), 60 + n, -5.0, 5.0)
        s = np.sort(x)
        for q in (0.0, 1.0, 0.5, 0.25, 1.0 / 3.0, 0.99):
            index = np.float32(n - 1) * np.float32(q)
            lo = int(index)
            hi = min(lo + 1, n - 1)
            w = np.float32(index - np.float32(lo))
            # one fma: (1 - w) * lower + round(w * upper), as gcc -mfma compiles the reference's expression
            exact = np.float64(np.float32(1) - w) * np.float64(s[lo]) + np.float64(np.float32(w * s[hi]))
            got, stats = oracle.quantile(x, q, with_stats=True)
            assert stats.tolist() == [s[lo], s[hi]]
            assert got.view(np.uint32) == np.float32(exact).view(np.uint32), (n, q)
