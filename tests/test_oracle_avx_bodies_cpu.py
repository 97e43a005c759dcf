"""The oracle's 8-wide (AVX2) loop bodies against a SECOND, independent restatement of the same reference text.

The reference's KATs (tests/golden/phpt_vectors.json) have at most four elements: they pin the scalar tails of
`arithmetics.c` / `logic.c`, never the `for (; i < n - 7; i += 8)` bodies — where the reference's quirks live.
No reference output exists for those bodies (the C files need <php.h>; stand-ins are not permitted), so this is not a pin
against the reference.  It is the next best thing: oracle/np_oracle.c (C, intrinsics-shaped loops) and the numpy code
below (written from the reference source alone, element rules instead of loops) are two restatements by different means;
a slip in either shows up as a difference.  Every rule cites the line it restates.
  multiply   body: product, every zero product becomes -0.0 (fix_negative_zero, arithmetics.c:280-284,397-403);
             tail: product, a -0.0 product becomes +0.0 (:406-412)
  mod        body: a - floor(a / b) * b with the multiply-subtract contracted to one FMA, as `gcc -mavx2 -march=native`
             builds it (:788-795; oracle/np_oracle.c says why); tail: fmodf (:798-800)
  equal      body: exact compare, NaN != NaN (logic.c:535-547); tail: |a - b| <= 1e-7 (:550-552)
  not_equal  body: ordered not-equal — a NaN on either side gives 0 (logic.c:636-648); tail: |a - b| <= 1e-7 -> 0 (:651-653)
  add / subtract / divide and the four ordered compares: one IEEE operation per element, body and tail alike
The body ends at (n // 8) * 8 elements of the FIRST operand after the scalar expand (arithmetics.c:251); operands here
have equal shapes, so that is n."""
from fractions import Fraction

import numpy as np
import pytest

from numpower_amd import synth

SIZES = [1, 7, 8, 9, 15, 16, 17, 64, 100, 257]


def _inputs(n, seed):
    a = synth.uniform((n,), seed, -4.0, 4.0)
    b = synth.uniform((n,), seed + 1, -4.0, 4.0)
    # zeros of both signs, exact ties, infinities and NaNs sprinkled over body and tail alike
    specials = np.float32([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 2.5, 1e-7, 3e-8])
    rng = np.random.default_rng(seed)
    for arr in (a, b):
        idx = rng.integers(0, n, size=max(1, n // 5))
        arr[idx] = rng.choice(specials, size=idx.size)
    k = rng.integers(0, n, size=max(1, n // 4))
    b[k] = a[k]                                             # exact ties for the compares
    return a, b


def _bits(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


def _same(got, want, what):
    both_nan = np.isnan(got) & np.isnan(want)
    bad = (_bits(got) != _bits(want)) & ~both_nan
    assert not bad.any(), (what, np.flatnonzero(bad)[:5], got[bad][:5], want[bad][:5])


def _fma_sub(a, fl, b):
    """float32(a - fl * b) with ONE rounding (what vfnmadd computes), exactly: rational arithmetic, then the correctly
    rounded double -> float32 (a double rounding could only matter at an exact float32 midpoint of the double)."""
    out = np.empty(a.shape, np.float32)
    for i in range(a.size):
        x, f, y = float(a[i]), float(fl[i]), float(b[i])
        if not (np.isfinite(x) and np.isfinite(f) and np.isfinite(y)):
            with np.errstate(invalid="ignore"):
                out[i] = np.float32(np.float64(x) - np.float64(f) * np.float64(y))   # inf / NaN rules need no exactness
            continue
        exact = Fraction(x) - Fraction(f) * Fraction(y)
        if exact == 0:
            # an exact zero from an FMA takes the sign IEEE gives x + (-(f * y)): +0 unless both addends are -0
            prod_neg_zero = (f == 0.0 or y == 0.0) and (np.signbit(np.float32(f)) == np.signbit(np.float32(y)))   # -(f * y) is -0
            out[i] = np.float32(-0.0) if (x == 0.0 and np.signbit(np.float32(x)) and prod_neg_zero) else np.float32(0.0)
        else:
            out[i] = np.float32(float(exact))
    return out


def second_restatement(op, a, b):
    n = a.size
    body = (n // 8) * 8
    with np.errstate(all="ignore"):
        if op == "add":
            return a + b
        if op == "subtract":
            return a - b
        if op == "divide":
            return a / b
        if op == "multiply":
            p = a * b
            out = p.copy()
            zero = p == 0
            out[:body][zero[:body]] = np.float32(-0.0)
            out[body:][zero[body:]] = np.float32(0.0)
            return out
        if op == "mod":
            out = np.empty(n, np.float32)
            fl = np.floor(a[:body] / b[:body]).astype(np.float32)
            out[:body] = _fma_sub(a[:body], fl, b[:body])
            out[body:] = np.fmod(a[body:], b[body:])
            return out
        if op == "equal":
            out = np.empty(n, np.float32)
            out[:body] = (a[:body] == b[:body]).astype(np.float32)
            out[body:] = (np.abs(a[body:] - b[body:]) <= np.float32(0.0000001)).astype(np.float32)
            return out
        if op == "not_equal":
            out = np.empty(n, np.float32)
            ordered = ~(np.isnan(a[:body]) | np.isnan(b[:body]))
            out[:body] = ((a[:body] != b[:body]) & ordered).astype(np.float32)
            out[body:] = (~(np.abs(a[body:] - b[body:]) <= np.float32(0.0000001))).astype(np.float32)
            return out
        cmp = {"greater": np.greater, "greater_equal": np.greater_equal, "less": np.less, "less_equal": np.less_equal}[op]
        return cmp(a, b).astype(np.float32)


@pytest.mark.parametrize("op", ["add", "subtract", "multiply", "divide", "mod", "equal", "not_equal", "greater", "greater_equal",
                                "less", "less_equal"])
def test_oracle_bodies_against_a_second_restatement(op, oracle):
    for n in SIZES:
        for seed in (11, 23, 37):
            a, b = _inputs(n, seed * 1000 + n)
            got = oracle.binary(op, a, b)
            want = second_restatement(op, a, b)
            if op == "mod":
                # x mod 0, inf mod y: the body's a - floor(a / b) * b and fmodf agree on NaN-ness but a NaN's sign / payload is
                # not part of either text
                assert (np.isnan(got) == np.isnan(want)).all(), (n, seed)
            _same(got, want, (op, n, seed))


def test_multiply_zero_signs_follow_the_loop_position(oracle):
    """The one quirk a user can see: the same product is -0.0 in the 8-wide body and +0.0 in the tail."""
    a = np.zeros(19, np.float32)
    b = np.float32([1, -1] * 9 + [1])
    got = oracle.binary("multiply", a, b)
    assert np.signbit(got[:16]).all() and not np.signbit(got[16:]).any()
    assert (_bits(got) == _bits(second_restatement("multiply", a, b))).all()


def test_equal_rules_differ_between_body_and_tail(oracle):
    """inf == inf is true in the body (exact compare) and false in the tail (|inf - inf| = NaN fails the tolerance test); two
    different numbers within 1e-7 of each other are unequal in the body and equal in the tail; NaN != NaN is 1 in the
    tail's not_equal and 0 in the body's (ordered compare).  The same values are placed in both regions of a 19-element pair."""
    n = 19
    a = np.arange(n, dtype=np.float32)
    b = a.copy()
    for at in (0, 16):                      # body position, tail position
        a[at], b[at] = np.inf, np.inf
        a[at + 1], b[at + 1] = np.float32(1e-7), np.float32(3e-8)
        a[at + 2], b[at + 2] = np.nan, np.nan
    eq, ne = oracle.binary("equal", a, b), oracle.binary("not_equal", a, b)
    assert eq[[0, 1, 2]].tolist() == [1.0, 0.0, 0.0] and eq[[16, 17, 18]].tolist() == [0.0, 1.0, 0.0]
    assert ne[[0, 1, 2]].tolist() == [0.0, 1.0, 0.0] and ne[[16, 17, 18]].tolist() == [1.0, 0.0, 1.0]
    assert (_bits(eq) == _bits(second_restatement("equal", a, b))).all()
    assert (_bits(ne) == _bits(second_restatement("not_equal", a, b))).all()


@pytest.mark.parametrize("shape", [(5,), (64,), (1000,), (7, 9), (33, 8), (4, 5, 17), (3, 16, 8)])
def test_oracle_reductions_against_a_second_restatement(shape, oracle):
    """Reductions, restated a second time from the reference text:
      NDArray_Sum_Float / Float_Prod (arithmetics.c:36-71): one fp32 accumulator walked over the elements in memory order —
        numpy's accumulate (ufunc.accumulate is sequential, unlike np.sum's pairwise tree);
      NDArray_Min / NDArray_Max (ndarray.c:752-772,939-959): `if (x < min) min = x` — a NaN never replaces;
      reduce(a, &axis, NDArray_Add_Float | NDArray_Multiply_Float) (ndarray.c:394-429,523-578): the result starts as slice 0 and
        every further slice is folded in by ONE call of the elementwise function — so per element a sequential sum / product
        over the axis, the products with Multiply_Float's zero-sign rule applied per fold (body / tail by the SLICE's size)."""
    x = synth.uniform(shape, 91, -2.0, 2.0)
    flat = x.reshape(-1)
    with np.errstate(all="ignore"):
        assert _bits(oracle.reduce_all("sum", x)) == _bits(np.add.accumulate(flat, dtype=np.float32)[-1])
        assert _bits(oracle.reduce_all("prod", x)) == _bits(np.multiply.accumulate(flat, dtype=np.float32)[-1])
        assert oracle.reduce_all("min", x) == flat.min() and oracle.reduce_all("max", x) == flat.max()
        for axis in range(len(shape)):
            if len(shape) == 1:
                continue
            xm = np.moveaxis(x, axis, 0)                      # slices along the axis, each C-contiguous in the reference's walk
            want_sum = xm[0].copy()
            want_prod = xm[0].copy()
            for k in range(1, xm.shape[0]):
                want_sum = want_sum + xm[k]
                if want_prod.ndim > 0:      # slices with at least one axis go through Multiply_Float's loops (0-d ones take its short cut)
                    want_prod = second_restatement("multiply", np.ascontiguousarray(want_prod).reshape(-1),
                                                   np.ascontiguousarray(xm[k]).reshape(-1)).reshape(want_prod.shape)
                else:
                    want_prod = want_prod * xm[k]
            got_sum, got_prod = oracle.reduce_axis("sum", x, axis), oracle.reduce_axis("prod", x, axis)
            _same(np.asarray(got_sum).reshape(-1), np.asarray(want_sum, np.float32).reshape(-1), ("sum", shape, axis))
            _same(np.asarray(got_prod).reshape(-1), np.asarray(want_prod, np.float32).reshape(-1), ("prod", shape, axis))
    y = x.copy().reshape(-1)
    y[1::3] = np.nan                                          # NaNs that are not first: min / max ignore them
    assert oracle.reduce_all("min", y) == np.nanmin(y) and oracle.reduce_all("max", y) == np.nanmax(y)


def test_oracle_argreduce_transpose_matmul_against_numpy(oracle):
    """The rest of the checker's surface against numpy where numpy has the same definition: argmax / argmin without NaNs
    (first occurrence on ties, calculation.c:9-72) and the two NaN rules spelled out; transpose = numpy's strided copy
    (manipulation.c:68-130, :381-421); matmul (cblas_sgemm through the bundled OpenBLAS, linalg.c:75-79) within 1e-6 |A|.|B| of
    the fp64 product."""
    x = synth.uniform((37, 11, 5), 92, -1.0, 1.0)
    x.reshape(-1)[::7] = np.float32(0.5)                       # ties
    for axis in (None, 0, 1, 2):
        assert (oracle.argreduce(x, axis, True) == np.argmax(x, axis)).all()
        assert (oracle.argreduce(x, axis, False) == np.argmin(x, axis)).all()
    nan = np.nan
    for row, amax, amin in (([nan, 1, 5, 2], 0, 0), ([1, nan, 5, 2], 2, 1), ([1, 5, nan, nan, -7], 1, 2), ([np.inf, 2, np.inf], 0, 1)):
        r = np.float32(row)
        assert (float(oracle.argreduce(r, None, True)), float(oracle.argreduce(r, None, False))) == (amax, amin), row
    for axes in (None, (0, 2, 1), (2, 0, 1), (1, 0, 2)):
        assert (_bits(oracle.transpose(x, axes)) == _bits(np.ascontiguousarray(np.transpose(x, axes)))).all()
    a, b = synth.uniform((130, 70), 93, -1.0, 1.0), synth.uniform((70, 96), 94, -1.0, 1.0)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    assert (np.abs(oracle.matmul(a, b) - ref) <= 1e-6 * scale).all()
