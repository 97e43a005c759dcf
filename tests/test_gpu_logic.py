"""SURVEY.md §8(f) row 1 on the GPU: comparison elementwise ops and NDArray::all against the oracle
(src/logic.c:25-670), bit-exact, including what the reference's CPU code does differently in its
AVX2 body and its scalar tail (equal / not_equal: exact vs |a-b| <= 1e-7; all: the 0x0F mask)."""
import numpy as np
import pytest

from numpower_amd import synth

pytestmark = pytest.mark.gpu

OPS = ["equal", "not_equal", "greater", "greater_equal", "less", "less_equal"]


def _pair(shape, seed):
    a = synth.uniform(shape, seed, -2.0, 2.0)
    b = a.copy()
    fa, fb = a.reshape(-1), b.reshape(-1)
    fb[::3] = synth.uniform((fb[::3].size,), seed + 1, -2.0, 2.0)      # different values
    fb[1::7] = fa[1::7] + np.float32(5e-8)                              # inside the 1e-7 tolerance
    fb[2::11] = np.nextafter(fa[2::11], np.float32(10))                 # 1 ulp apart
    fa[5::13] = np.nan                                                  # unordered
    fb[6::17] = np.nan
    fa[7::19] = np.inf
    fb[7::19] = np.inf
    return a, b


def _bits(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("shape", [(1000, 1000), (257, 1001), (3, 5), (1, 7), (64, 4096)])
@pytest.mark.parametrize("op", OPS)
def test_compare_same_shape(op, shape, hip, oracle):
    from numpower_amd.ndarray import NDArray
    a, b = _pair(shape, 50 + shape[0])
    got = NDArray._binary(op, NDArray.array(a).gpu(), NDArray.array(b).gpu()).cpu().numpy()
    ref = oracle.binary(op, a, b)
    assert (_bits(got) == _bits(ref)).all(), "%s %s: %d differ" % (op, shape, int((_bits(got) != _bits(ref)).sum()))
    assert set(np.unique(got)) <= {0.0, 1.0}


@pytest.mark.parametrize("op", OPS)
def test_compare_scalar_and_broadcast(op, hip, oracle):
    from numpower_amd.ndarray import NDArray
    a, b = _pair((300, 40), 9)
    ga = NDArray.array(a).gpu()
    got = NDArray._binary(op, ga, 0.5).cpu().numpy()
    assert (_bits(got) == _bits(oracle.binary(op, a, np.float32(0.5)))).all()
    got = NDArray._binary(op, 0.5, ga).cpu().numpy()
    assert (_bits(got) == _bits(oracle.binary(op, np.float32(0.5), a))).all()
    # array (op) row / column: the reference's Less / Equal index the un-broadcast operand there
    # (undefined); everything else is defined and must match
    if op not in ("less", "equal"):
        row, col = b[0].copy(), b[:, :1].copy()
        for small in (row, col):
            got = NDArray._binary(op, ga, NDArray.array(small).gpu()).cpu().numpy()
            assert (_bits(got) == _bits(oracle.binary(op, a, small))).all(), op


def test_compare_errors(hip):
    from numpower_amd.ndarray import Error, NDArray
    a = NDArray.array(np.ones((4, 6), np.float32)).gpu()
    with pytest.raises(Error, match="Devices mismatch in `equal` function"):
        NDArray.equal(a, NDArray.array(np.ones((4, 6), np.float32)))
    with pytest.raises(Error, match="Can't broadcast arrays."):
        NDArray.greater(a, NDArray.array(np.ones((5,), np.float32)).gpu())


@pytest.mark.parametrize("n", [1, 4, 7, 8, 9, 16, 23, 1000, 1001, 100003])
def test_all_matches_reference_cpu_semantics(n, hip, oracle):
    """NDArray_All as the reference's CPU code computes it (logic.c:25-58): full 8-blocks pass only
    with lanes 0-3 non-zero and lanes 4-7 zero/NaN; the scalar tail means 'all non-zero'."""
    from numpower_amd.ndarray import NDArray
    rng = np.random.default_rng(n)
    cases = [np.ones(n, np.float32), np.zeros(n, np.float32), synth.uniform((n,), 3, 0.5, 1.5)]
    pat = np.tile(np.array([1, 1, 1, 1, 0, 0, 0, 0], np.float32), n // 8 + 1)[:n].copy()
    cases.append(pat.copy())                     # passes the body's 0x0F test block by block
    if n % 8:
        pat[n - 1] = 1.0                         # tail elements must be non-zero
        pat[(n // 8) * 8:] = 1.0
        cases.append(pat.copy())
        bad = pat.copy()
        bad[n - 1] = 0.0
        cases.append(bad)
    nanpat = pat.copy()
    if n >= 8:
        nanpat[5] = np.nan                        # NaN in lanes 4-7 counts as "zero" for the mask
        cases.append(nanpat)
        bad2 = pat.copy()
        bad2[2] = 0.0
        cases.append(bad2)
    cases.append(rng.choice(np.array([0.0, 1.0, -2.0], np.float32), size=n))
    for x in cases:
        got = NDArray.all(NDArray.array(x).gpu())
        assert got == int(oracle.reduce_all("all", x)), (n, x[:16])


def test_all_plain_flag(hip):
    """flags = 0 gives the intended meaning (every element non-zero) through the C ABI."""
    import ctypes as C
    from numpower_amd._lib import check, load
    D = hip
    for x, want in [(np.ones(1000, np.float32), 1), (np.r_[np.ones(999, np.float32), 0].astype(np.float32), 0),
                    (np.tile(np.array([1, 1, 1, 1, 0, 0, 0, 0], np.float32), 100), 0)]:
        d = D.DeviceArray.from_host(x)
        v = C.c_int()
        check(load().np_all(d.ptr, x.size, 0, C.byref(v)))
        assert v.value == want


@pytest.mark.parametrize("shape", [(1000, 1000), (257, 1001), (7,), (3, 5)])
def test_maximum_minimum(shape, hip, oracle):
    """NDArray_Maximum / NDArray_Minimum (ndarray.c:853-931; the reference refuses GPU arrays): glibc
    fmaxf / fminf semantics — a NaN loses to a number — with scalar / row / column broadcast."""
    from numpower_amd.ndarray import NDArray
    a = synth.uniform(shape, 81, -2.0, 2.0)
    b = synth.uniform(shape, 82, -2.0, 2.0)
    a.reshape(-1)[::9] = np.nan
    b.reshape(-1)[::13] = np.nan
    b.reshape(-1)[4::17] = a.reshape(-1)[4::17]          # exact ties
    ga, gb = NDArray.array(a).gpu(), NDArray.array(b).gpu()
    for name, ref in (("maximum", np.fmax), ("minimum", np.fmin)):
        got = getattr(NDArray, name)(ga, gb).cpu().numpy()
        want = oracle.binary(name, a, b)
        same = (got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), name
        with np.errstate(invalid="ignore"):
            assert np.array_equal(got, ref(a, b), equal_nan=True)
        # scalar operand, either side
        got = getattr(NDArray, name)(ga, 0.5).cpu().numpy()
        with np.errstate(invalid="ignore"):
            assert np.array_equal(got, ref(a, np.float32(0.5)), equal_nan=True)
        got = getattr(NDArray, name)(-0.25, gb).cpu().numpy()
        with np.errstate(invalid="ignore"):
            assert np.array_equal(got, ref(np.float32(-0.25), b), equal_nan=True)
    if len(shape) == 2:
        row = synth.uniform((shape[1],), 83, -1.0, 1.0)
        col = synth.uniform((shape[0], 1), 84, -1.0, 1.0)
        with np.errstate(invalid="ignore"):
            assert np.array_equal(NDArray.maximum(ga, NDArray.array(row).gpu()).cpu().numpy(), np.fmax(a, row[None, :]), equal_nan=True)
            assert np.array_equal(NDArray.minimum(NDArray.array(col).gpu(), ga).cpu().numpy(), np.fmin(col, a), equal_nan=True)
    # inside a fused chain: relu(x * w) = maximum(x * w, 0)
    from numpower_amd.lazy import Lazy   # noqa: F401
    c = np.nan_to_num(a, nan=0.5)
    gc = NDArray.array(c).gpu()
    fused = (gc.lazy() * 1.5).maximum(0.0).eval().cpu().numpy()
    assert np.array_equal(fused, np.fmax(c * np.float32(1.5), np.float32(0.0)))


def test_inner_copy_negative(hip, oracle):
    from numpower_amd.ndarray import Error, NDArray
    a = synth.uniform((300, 70), 85, -1.0, 1.0)
    b = synth.uniform((300, 70), 86, -1.0, 1.0)
    ga, gb = NDArray.array(a).gpu(), NDArray.array(b).gpu()
    want = float((a.astype(np.float64) * b).sum())
    got = NDArray.inner(ga, gb)                       # N-D: one number shaped (1, 1)
    assert got.shape() == [1, 1] and abs(got.toArray()[0][0] - want) <= 1e-5 * np.abs(a * b).sum()
    v = NDArray.inner(NDArray.array(a[0]).gpu(), NDArray.array(b[0]).gpu())
    assert isinstance(v, float) and abs(v - float((a[0].astype(np.float64) * b[0]).sum())) <= 1e-5 * np.abs(a[0] * b[0]).sum()
    row = NDArray.inner(ga, NDArray.array(b[0]).gpu())   # (300, 70) . (70,): Multiply broadcasts, then sums everything
    assert abs(row.toArray()[0][0] - float((a.astype(np.float64) * b[0][None, :]).sum())) <= 1e-5 * np.abs(a * b[0]).sum()
    with pytest.raises(Error, match="Shape is not aligned"):
        NDArray.inner(ga, NDArray.array(np.ones((300, 69), np.float32)).gpu())
    c = ga.copy()
    assert c.isGPU() and np.array_equal(c.cpu().numpy(), a)
    ga.fill(0.0)
    assert np.array_equal(c.cpu().numpy(), a)          # a real copy
    assert np.array_equal(NDArray.negative(gb).cpu().numpy(), -b)
