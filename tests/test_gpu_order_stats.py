"""Order statistics on the device (np_order_stat: three-pass radix select) and the two reference
functions built on them: NDArray::median (calculate_median, arithmetics.c:111-138) and
NDArray::quantile (calculate_quantile, statistics.c:14-50).  The selected elements are compared bit
for bit against a sort of the order-preserving integer keys; median / quantile bit for bit against
the oracle's qsort restatement."""
import ctypes as C

import numpy as np
import pytest

from numpower_amd import synth

pytestmark = pytest.mark.gpu


def _keys(x):
    u = x.view(np.uint32)
    return np.where(u & 0x80000000, ~u, u | 0x80000000).astype(np.uint32)


def _from_keys(k):
    k = np.asarray(k, np.uint32)
    return np.where(k & 0x80000000, k & 0x7fffffff, ~k).astype(np.uint32).view(np.float32)


def _order_stat(x, k):
    from numpower_amd import _lib
    lib = _lib.load()
    buf = _lib.DeviceBuffer(4 * x.size)
    _lib.check(lib.np_memcpy_h2d(buf.ptr, x.ctypes.data, 4 * x.size))
    out = (C.c_float * 2)()
    _lib.check(lib.np_order_stat(buf.ptr, x.size, k, out))
    buf.free()
    return np.float32([out[0], out[1]])


def _cases():
    yield "uniform01", synth.uniform((1_000_003,), 71, 0.0, 1.0)
    yield "wide", (synth.uniform((300_001,), 72, -1.0, 1.0) * np.exp(synth.uniform((300_001,), 73, -40.0, 40.0))).astype(np.float32)
    yield "few_values", np.rint(synth.uniform((200_000,), 74, 0.0, 9.0)).astype(np.float32)
    yield "constant", np.full((70_001,), 3.25, np.float32)
    yield "signed_zeros", np.where(synth.uniform((50_000,), 75, 0.0, 1.0) < 0.5, np.float32(-0.0), np.float32(0.0)).astype(np.float32)
    yield "sorted", np.arange(100_000, dtype=np.float32)
    yield "reversed", np.arange(100_000, dtype=np.float32)[::-1].copy()
    yield "with_inf", np.concatenate([synth.uniform((9_999,), 76, -5.0, 5.0), np.float32([np.inf, -np.inf, np.inf])])
    yield "denormals", (synth.uniform((40_000,), 77, -1.0, 1.0) * np.float32(1e-41)).astype(np.float32)
    for n in (1, 2, 3, 4, 5, 7, 8, 9, 255, 256, 257, 1023, 1025):
        yield "n%d" % n, synth.uniform((n,), 300 + n, -2.0, 2.0)


@pytest.mark.parametrize("name,x", list(_cases()), ids=[c[0] for c in _cases()])
def test_order_stat_matches_a_sort(name, x, hip):
    x = np.ascontiguousarray(x, np.float32)
    n = x.size
    want = np.sort(_keys(x))
    ranks = sorted({0, n - 1, n // 2, max(n // 2 - 1, 0), n // 3, (7 * n) // 8, min(n - 1, 1), max(n - 2, 0)})
    for k in ranks:
        got = _order_stat(x, k)
        exp = _from_keys([want[k], want[min(k + 1, n - 1)]])
        assert got.view(np.uint32).tolist() == exp.view(np.uint32).tolist(), (name, k)


def test_order_stat_large_and_every_level(hip):
    """10^8 elements (three full passes), and hand-made arrays where ranks k and k+1 part at the first,
    second and third digit of the key."""
    x = synth.uniform((100_000_000,), 81, -3.0, 5.0)
    want = np.sort(_keys(x))
    for k in (0, 12_345_678, 49_999_999, 50_000_000, 99_999_999):
        got = _order_stat(x, k)
        exp = _from_keys([want[k], want[min(k + 1, x.size - 1)]])
        assert got.view(np.uint32).tolist() == exp.view(np.uint32).tolist(), k
    base = np.float32(1.5).view(np.uint32)
    for delta in (1, 1 << 9, 1 << 10, 1 << 15, 1 << 21, 1 << 22):   # successor differs in a low / middle / high digit
        hi = np.uint32(base + delta).view(np.float32)
        lo_vals = np.full(1000, np.float32(1.5))
        x = np.concatenate([lo_vals, np.full(500, hi), synth.uniform((3000,), 82, 2.0, 3.0)]).astype(np.float32)
        x = x[np.argsort(synth.uniform((x.size,), 83, 0.0, 1.0))]
        got = _order_stat(x, 999)
        assert got.view(np.uint32).tolist() == [base, base + delta], delta
        got = _order_stat(x, 998)
        assert got.view(np.uint32).tolist() == [base, base], delta


def test_order_stat_errors(hip):
    from numpower_amd import _lib
    lib = _lib.load()
    buf = _lib.DeviceBuffer(16)
    out = (C.c_float * 2)()
    assert lib.np_order_stat(buf.ptr, 4, 4, out) != 0
    assert b"out of range" in lib.np_last_error()
    assert lib.np_order_stat(buf.ptr, 0, 0, out) != 0
    buf.free()


@pytest.mark.parametrize("n", [1, 2, 5, 6, 1000, 1001, 65_536, 1_000_003])
def test_median_and_quantile_vs_oracle(n, hip, oracle):
    from numpower_amd.ndarray import NDArray
    x = (synth.uniform((n,), 90 + n % 7, -1.0, 1.0) * np.float32(100.0)).astype(np.float32)
    g = NDArray.array(x).gpu()
    got = np.float32(NDArray.median(g))
    want = oracle.median(x)
    assert got.view(np.uint32) == want.view(np.uint32), (n, got, want)
    for q in (0.0, 1.0, 0.5, 0.25, 0.75, 0.999, 1.0 / 3.0, 0.1):
        got = np.float32(NDArray.quantile(g, q))
        want = oracle.quantile(x, q)
        assert got.view(np.uint32) == want.view(np.uint32), (n, q, got, want)
    # 2-D input: flattened, like NDArray_NUMELEMENTS over NDArray_FDATA
    if n % 4 == 0:
        g2 = NDArray.array(x.reshape(4, -1)).gpu()
        assert np.float32(NDArray.median(g2)).view(np.uint32) == oracle.median(x).view(np.uint32)


def test_quantile_errors(hip):
    from numpower_amd.ndarray import Error, NDArray
    g = NDArray.array(np.float32([3, 1, 2])).gpu()
    with pytest.raises(Error, match="Q must be between 0 and 1"):
        NDArray.quantile(g, 1.5)
    with pytest.raises(Error, match="Q must be between 0 and 1"):
        NDArray.quantile(g, -0.1)
    with pytest.raises(Error, match="Q must be a scalar"):
        NDArray.quantile(g, [0.5, 0.6])
    assert NDArray.quantile(g, 0.5) == 2.0 and NDArray.median(g) == 2.0


@pytest.fixture
def bracket_everywhere():
    """Send every array of >= 2048 elements down the bracket path (sample -> bracket -> filter -> verdict),
    which by default only arrays of >= 2^26 elements take."""
    from numpower_amd import _lib
    lib = _lib.load()
    _lib.check(lib.np_select_set_variant(2048))
    yield lib
    _lib.check(lib.np_select_set_variant(1))


def _path():
    from numpower_amd import _lib
    path = C.c_int(-1)
    _lib.check(_lib.load().np_select_last_path(C.byref(path)))
    return path.value


def test_bracket_is_refused_or_dropped_when_it_cannot_help(hip, bracket_everywhere):
    n = 1_500_000
    relu = np.maximum(synth.uniform((n,), 95, -1.0, 1.0), np.float32(0.0))
    want = np.sort(_keys(relu))
    got = _order_stat(relu, n // 4)                     # inside the 50 % of zeros: no bracket can be tighter
    assert got.view(np.uint32).tolist() == _from_keys([want[n // 4], want[n // 4 + 1]]).view(np.uint32).tolist()
    assert _path() == 0
    got = _order_stat(relu, n * 9 // 10)                # among the positive half: bracketed
    assert got.view(np.uint32).tolist() == _from_keys([want[n * 9 // 10], want[n * 9 // 10 + 1]]).view(np.uint32).tolist()
    assert _path() == 1


def _bracket_cases():
    for name, x in _cases():
        if x.size >= 2048:
            yield name, x
    n = 1_500_000
    u = synth.uniform((n,), 91, 0.0, 1.0)
    yield "normalish", (synth.uniform((n,), 92, -1.0, 1.0) + synth.uniform((n,), 93, -1.0, 1.0) + u).astype(np.float32)
    # the sample lies: 1024 evenly spaced runs of 1024 floats see only the planted value
    lying = synth.uniform((n,), 94, 0.0, 1.0)
    nvec = n // 4
    for b in range(1024):
        start = (b * (nvec - 256) // 1023) * 4
        lying[start:start + 1024] = np.float32(1000.0)
    yield "sample_lies", lying
    yield "half_zero", np.maximum(synth.uniform((n,), 95, -1.0, 1.0), np.float32(0.0))       # a ReLU output: the bracket is refused
    yield "two_clusters", np.where(u < 0.5, np.float32(1e-3), np.float32(1e3)) * synth.uniform((n,), 96, 1.0, 1.001)
    with_nan = synth.uniform((n,), 97, -4.0, 4.0)
    with_nan[::1001] = np.nan
    with_nan[5::2003] = -np.nan
    yield "with_nan", with_nan


@pytest.mark.parametrize("name,x", list(_bracket_cases()), ids=[c[0] for c in _bracket_cases()])
def test_order_stat_bracket_path_matches_a_sort(name, x, hip, bracket_everywhere):
    x = np.ascontiguousarray(x, np.float32)
    n = x.size
    want = np.sort(_keys(x))
    ranks = sorted({0, 1, n - 1, n - 2, n // 2, n // 2 - 1, n // 3, (7 * n) // 8, n // 1000, n - n // 1000})
    # ranks whose successor sits in another top-level bin (where the bracket ends, if it ends there)
    top = want >> 21
    edges = np.flatnonzero(top[1:] != top[:-1])
    ranks = sorted(set(ranks) | {int(e) for e in edges[:3]} | {int(e) + 1 for e in edges[:3]} | {int(e) for e in edges[-2:]})
    for k in ranks:
        got = _order_stat(x, k)
        exp = _from_keys([want[k], want[min(k + 1, n - 1)]])
        assert got.view(np.uint32).tolist() == exp.view(np.uint32).tolist(), (name, k)


def test_bracket_path_is_taken_and_agrees_with_the_plain_passes(hip):
    """7 * 10^7 elements (default threshold 2^26): same answers with the path on and off, for every decile."""
    from numpower_amd import _lib
    lib = _lib.load()
    x = (synth.uniform((70_000_000,), 98, -1.0, 1.0) * np.exp(synth.uniform((70_000_000,), 99, -3.0, 3.0))).astype(np.float32)
    want = np.sort(_keys(x))
    for k in [0, x.size - 1] + [x.size * d // 10 for d in range(1, 10)]:
        exp = _from_keys([want[k], want[min(k + 1, x.size - 1)]]).view(np.uint32).tolist()
        _lib.check(lib.np_select_set_variant(0))
        plain = _order_stat(x, k)
        _lib.check(lib.np_select_set_variant(1))
        fast = _order_stat(x, k)
        path = C.c_int(-1)
        _lib.check(lib.np_select_last_path(C.byref(path)))
        assert plain.view(np.uint32).tolist() == exp, k
        assert fast.view(np.uint32).tolist() == exp, k
        assert path.value == 1, "rank %d fell back to the plain passes" % k


def test_small_arrays_take_the_one_workgroup_kernel_every_rank(hip):
    """n <= 32768: select_small_kernel (one launch).  Every rank of small arrays, incl. duplicates, -0 / +0, +-inf and
    NaNs of both signs (the all-ones NaN is the largest key there is: the 'no successor' sentinel must not eat it)."""
    rng = np.random.default_rng(5)
    specials = np.array([0.0, -0.0, np.inf, -np.inf], np.float32)
    allones_nan = np.array([0x7fffffff, 0xffffffff, 0x7fc00000], np.uint32).view(np.float32)
    for n in (1, 2, 3, 5, 64, 65, 255, 256, 257, 300):
        for style in range(4):
            x = synth.uniform((n,), 700 + n + style, -2.0, 2.0)
            if style == 1:
                x = np.rint(x).astype(np.float32)
            elif style == 2:
                x[rng.integers(0, n, max(1, n // 4))] = specials[rng.integers(0, 4, max(1, n // 4))]
            elif style == 3:
                x[rng.integers(0, n, max(1, n // 3))] = allones_nan[rng.integers(0, 3, max(1, n // 3))]
            want = np.sort(_keys(x))
            for k in range(n):
                got = _order_stat(x, k)
                exp = _from_keys([want[k], want[min(k + 1, n - 1)]])
                assert got.view(np.uint32).tolist() == exp.view(np.uint32).tolist(), (n, style, k)
            assert _path() == 2
    for n in (1023, 1024, 1025, 4097, 32767, 32768):
        x = synth.uniform((n,), 800 + n, -1.0, 1.0) * np.float32(1e-3) + np.float32(1.0)   # one binade: low bytes decide
        want = np.sort(_keys(x))
        for k in sorted({0, 1, n // 2, n - 2, n - 1, int(rng.integers(0, n)), int(rng.integers(0, n))}):
            got = _order_stat(x, k)
            exp = _from_keys([want[k], want[min(k + 1, n - 1)]])
            assert got.view(np.uint32).tolist() == exp.view(np.uint32).tolist(), (n, k)
        assert _path() == 2
    x = synth.uniform((32769,), 9, -1.0, 1.0)
    _order_stat(x, 5)
    assert _path() == 0


@pytest.mark.parametrize("name,x", [c for c in _cases() if c[1].size <= 32768], ids=[c[0] for c in _cases() if c[1].size <= 32768])
def test_plain_passes_still_serve_small_arrays(name, x, hip):
    """np_select_set_variant(0): the three-pass kernels on the sizes the one-workgroup kernel normally takes."""
    from numpower_amd import _lib
    lib = _lib.load()
    x = np.ascontiguousarray(x, np.float32)
    n = x.size
    want = np.sort(_keys(x))
    _lib.check(lib.np_select_set_variant(0))
    try:
        for k in sorted({0, n - 1, n // 2, max(n // 2 - 1, 0), n // 3, min(n - 1, 1), max(n - 2, 0)}):
            got = _order_stat(x, k)
            exp = _from_keys([want[k], want[min(k + 1, n - 1)]])
            assert got.view(np.uint32).tolist() == exp.view(np.uint32).tolist(), (name, k)
        assert _path() == 0
    finally:
        _lib.check(lib.np_select_set_variant(1))
