"""GPU parity tests: the HIP path (through the C ABI / host library) against the oracle.

Bar (BASELINE.json north_star): bit-exact where the arithmetic is exact (add, subtract, multiply,
divide, mod incl. the reference's AVX-body quirks, rounding ops, sqrt, rsqrt bit hack, clip ...),
<= 1e-5 relative for transcendentals / reductions / GEMM.  Reductions and GEMM are additionally
checked against an fp64 accumulation, because the reference's sequential fp32 sums themselves
drift from the exact result at large N.
"""
import numpy as np
import pytest

from numpower_amd import synth
from tests.phpt_replay import GpuBackend, load_vectors, replay

pytestmark = pytest.mark.gpu

# ops whose result must equal the oracle bit for bit
EXACT_UNARY = ["abs", "sqrt", "rint", "fix", "floor", "ceil", "trunc", "negate", "sign", "rsqrt",
               "positive", "reciprocal", "degrees", "radians", "logb"]
# libm-class functions: glibc (oracle) and the device libm (ocml) are both ~1 ulp accurate but
# not bit-identical; tolerance = 1e-5 relative (north_star) with an absolute floor for results
# that cross zero.
TRANSCENDENTAL_UNARY = ["exp", "exp2", "expm1", "log", "log2", "log10", "log1p", "sin", "cos", "tan",
                        "arcsin", "arccos", "arctan", "sinh", "cosh", "tanh", "arcsinh", "arccosh",
                        "arctanh", "sinc"]
REL_TOL = 1e-5

DOMAIN = {  # input ranges inside each function's domain
    "log": (1e-3, 1e3), "log2": (1e-3, 1e3), "log10": (1e-3, 1e3), "log1p": (-0.99, 1e3),
    "logb": (1e-3, 1e3), "sqrt": (0.0, 1e3), "rsqrt": (1e-3, 1e3), "arcsin": (-1.0, 1.0),
    "arccos": (-1.0, 1.0), "arccosh": (1.0, 1e3), "arctanh": (-0.999, 0.999), "exp": (-10, 10),
    "exp2": (-10, 10), "expm1": (-10, 10), "sinh": (-10, 10), "cosh": (-10, 10),
    "reciprocal": (0.1, 100.0),
}


def bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(got, ref, what):
    got = np.asarray(got, dtype=np.float32)
    ref = np.asarray(ref, dtype=np.float32)
    assert got.shape == ref.shape, what
    both_nan = np.isnan(got) & np.isnan(ref)
    diff = (bits(got) != bits(ref)) & ~both_nan
    assert not diff.any(), "%s: %d of %d elements differ, first at %s: got %r ref %r" % (
        what, int(diff.sum()), diff.size, np.argwhere(diff)[0].tolist(),
        got[tuple(np.argwhere(diff)[0])], ref[tuple(np.argwhere(diff)[0])])


def assert_close(got, ref, what, rel=REL_TOL, abs_floor=1e-6):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, what
    finite = np.isfinite(ref)
    assert (np.isfinite(got) == finite).all(), what + ": non-finite pattern differs"
    err = np.abs(got[finite] - ref[finite])
    bound = rel * np.abs(ref[finite]) + abs_floor * rel
    bad = err > bound
    assert not bad.any(), "%s: max rel err %.3g" % (what, float((err / np.maximum(np.abs(ref[finite]), 1e-30)).max()))


# ---------------------------------------------------------------------------------------------
# 1. the reference's own KATs through the GPU path
# ---------------------------------------------------------------------------------------------

TESTS = load_vectors()
_TEXT_EXACT = {"add", "subtract", "multiply", "divide", "mod", "abs", "clip", "sign", "sqrt", "square",
               "ceil", "fix", "floor", "rint", "round", "trunc", "prod", "sum", "max", "min", "matmul",
               "degrees", "radians", "logb"}


def _numbers(text):
    import re
    toks = re.findall(r"=> (-?[\w.+-]+)$", text, flags=re.M)
    return toks


@pytest.mark.parametrize("test", TESTS, ids=[t["source"].split("/")[-1] for t in TESTS])
def test_phpt_kat_on_gpu(test, hip):
    got = replay(GpuBackend(), test).rstrip()
    want = test["expect"].rstrip()
    op = test["title"].split("::")[-1].strip().lower()
    if got == want:
        return
    assert op not in _TEXT_EXACT, "exact op %s: GPU text differs from the reference's EXPECT" % op
    # transcendental: same structure, numbers within 1e-5 relative
    g, w = _numbers(got), _numbers(want)
    assert len(g) == len(w) and len(g) > 0
    for a, b in zip(g, w):
        if a == b:
            continue
        fa, fb = float(a.replace("NAN", "nan")), float(b.replace("NAN", "nan"))
        assert abs(fa - fb) <= REL_TOL * abs(fb) + 1e-12, (op, a, b)


# ---------------------------------------------------------------------------------------------
# 2. binary ops vs the oracle, all operand shapes, incl. the AVX-body quirks
# ---------------------------------------------------------------------------------------------

SHAPES = [(1000, 1000), (257, 1001), (3, 5), (64, 4096), (1, 7), (9, 1)]


def _pair(shape, seed, lo=-4.0, hi=4.0):
    a = synth.uniform(shape, seed, lo, hi)
    b = synth.uniform(shape, seed + 1, lo, hi)
    # plant exact zeros, negative zeros and negative operands so the multiply/mod quirks fire
    flat_a, flat_b = a.reshape(-1), b.reshape(-1)
    flat_a[::7] = 0.0
    flat_a[3::11] = -0.0
    flat_b[5::13] = np.float32(-2.5)
    flat_b[flat_b == 0] = np.float32(1.0)
    return a, b


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("op", ["add", "subtract", "multiply", "divide", "mod"])
def test_binary_exact_same_shape(op, shape, hip, oracle):
    from numpower_amd.ndarray import NDArray
    a, b = _pair(shape, 100 + len(shape) + shape[0])
    got = NDArray._binary(op, NDArray.array(a).gpu(), NDArray.array(b).gpu()).cpu().numpy()
    assert_bit_equal(got, oracle.binary(op, a, b), "%s %s" % (op, shape))


@pytest.mark.parametrize("shape", [(1000, 1000), (257, 1001), (12, 4), (5, 3)])
@pytest.mark.parametrize("op", ["add", "subtract", "multiply", "divide", "mod"])
def test_binary_exact_broadcast(op, shape, hip, oracle):
    """scalar / row vector / column / 1xC operands on either side (ndarray.c:1196-1291)."""
    from numpower_amd.ndarray import NDArray
    a, b = _pair(shape, 7)
    r, c = shape
    ga = NDArray.array(a).gpu()
    row = b[0].copy()
    col = b[:, :1].copy()
    cases = {
        "arr,scalar": (ga, 2.5, a, np.float32(2.5)),
        "scalar,arr": (-1.5, ga, np.float32(-1.5), a),
        "arr,row": (ga, NDArray.array(row).gpu(), a, row),
        "row,arr": (NDArray.array(row).gpu(), ga, row, a),
        "arr,col": (ga, NDArray.array(col).gpu(), a, col),
        "col,arr": (NDArray.array(col).gpu(), ga, col, a),
    }
    for name, (x, y, ox, oy) in cases.items():
        got = NDArray._binary(op, x, y).cpu().numpy()
        assert_bit_equal(got, oracle.binary(op, ox, oy), "%s %s %s" % (op, shape, name))
    if r == c:   # the reference's 1xC -> RxC branch only fills the result when C == R
        one_by_c = b[:1].copy()
        got = NDArray._binary(op, ga, NDArray.array(one_by_c).gpu()).cpu().numpy()
        assert_bit_equal(got, oracle.binary(op, a, one_by_c), "%s %s arr,1xC" % (op, shape))


@pytest.mark.parametrize("shape", [(2, 1 << 22), (3, (1 << 22) + 256), (5, 4500004), (9, 1 << 22)])
def test_binary_long_row_operand_column_block_order(shape, hip, oracle):
    """A row operand of >= 16 MB is walked in column blocks (np_elementwise.hip Ragged::rowblock_rows: the workgroups in flight
    share a piece of the vector instead of each result row re-reading all of it): the same bits as the plain order
    (np_elementwise_set_variant(8000)) and as the oracle, with the row vector on either side, quirk flags included."""
    from numpower_amd.ndarray import NDArray
    from numpower_amd._lib import check, load
    lib = load()
    a, b = _pair(shape, 17)
    row = b[0].copy()
    ga, grow = NDArray.array(a).gpu(), NDArray.array(row).gpu()
    for op in ("add", "multiply", "mod"):
        for x, y, ox, oy in ((ga, grow, a, row), (grow, ga, row, a)):
            got = NDArray._binary(op, x, y).cpu().numpy()
            check(lib.np_elementwise_set_variant(8000))
            try:
                plain = NDArray._binary(op, x, y).cpu().numpy()
            finally:
                check(lib.np_elementwise_set_variant(0))
            assert_bit_equal(got, plain, "%s %s vs the plain order" % (op, shape))
            assert_bit_equal(got, oracle.binary(op, ox, oy), "%s %s" % (op, shape))


@pytest.mark.parametrize("shape", [(37, 1024), (2, 2048), (1001, 1056), (3, 65536), (5, 131104), (8, 70016)])
def test_binary_row_operand_2d_form(shape, hip, oracle):
    """X (op) row on whole, 128-byte-aligned float4 columns runs binary_rows2d_kernel (a lane owns one float4 column of two
    rows; np_elementwise.hip) — for the row vector as the LEFT operand from 1024 columns, as the right one from 65536: the same
    bits as the flat walk (np_elementwise_set_variant(8100)) and as the oracle, odd row counts and quirk flags included."""
    from numpower_amd.ndarray import NDArray
    from numpower_amd._lib import check, load
    lib = load()
    a, b = _pair(shape, 23)
    row = b[0].copy()
    ga, grow = NDArray.array(a).gpu(), NDArray.array(row).gpu()
    for op in ("add", "divide", "multiply", "mod"):
        for x, y, ox, oy in ((ga, grow, a, row), (grow, ga, row, a)):
            got = NDArray._binary(op, x, y).cpu().numpy()
            check(lib.np_elementwise_set_variant(8100))
            try:
                flat = NDArray._binary(op, x, y).cpu().numpy()
            finally:
                check(lib.np_elementwise_set_variant(8101))
            assert_bit_equal(got, flat, "%s %s vs the flat walk" % (op, shape))
            assert_bit_equal(got, oracle.binary(op, ox, oy), "%s %s" % (op, shape))


def test_binary_view_operand(hip, oracle):
    """$a + $a[1]: the row operand is a view into the same buffer (unaligned for odd widths)."""
    from numpower_amd.ndarray import NDArray
    for shape in [(6, 1001), (6, 1000)]:
        a, _ = _pair(shape, 3)
        ga = NDArray.array(a).gpu()
        got = (ga * ga[1]).cpu().numpy()
        assert_bit_equal(got, oracle.binary("multiply", a, a[1]), "view %s" % (shape,))


@pytest.mark.parametrize("shape", [(1000, 1000), (33, 65)])
def test_pow_and_arctan2(shape, hip, oracle):
    from numpower_amd.ndarray import NDArray
    base = synth.uniform(shape, 11, 0.1, 4.0)
    ex = synth.uniform(shape, 12, -3.0, 3.0)
    got = NDArray.pow(NDArray.array(base).gpu(), NDArray.array(ex).gpu()).cpu().numpy()
    assert_close(got, oracle.binary("pow", base, ex), "pow")
    y = synth.uniform(shape, 13, -3.0, 3.0)
    got = NDArray.arctan2(NDArray.array(ex).gpu(), NDArray.array(y).gpu()).cpu().numpy()
    assert_close(got, oracle.binary("arctan2", ex, y), "arctan2")


def test_pow_special_cases_and_range(hip, oracle):
    """pow: the fp64 fast path (finite normal base, finite exponent, negative bases with integer exponents)
    and the library fallback (zeros, denormals, inf, NaN) against glibc's powf (arithmetics.c:912-914):
    same NaN pattern, same zeros / infinities with sign, finite results within 1 ulp; over / underflow
    into inf, denormals and zero."""
    from numpower_amd.ndarray import NDArray
    f = np.float32
    xs = f([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 2.0, -2.0, 3.0, -3.0, np.inf, -np.inf, np.nan, 1e-40, -1e-40, 3.4028235e38,
            -3.4028235e38, 1.1754944e-38, -1.1754944e-38, 1e-30, 1e30, -1e30, 0.99999994, 1.0000001, -7.5, 10.0])
    ys = f([0.0, -0.0, 1.0, -1.0, 2.0, -2.0, 3.0, -3.0, 0.5, -0.5, 1.5, np.inf, -np.inf, np.nan, 1e10, -1e10, 16777216.0,
            16777215.0, -16777215.0, 33554432.0, 127.0, -149.0, 1e-40, 4.0, 5.0, 1e-3, 38.5, -45.25])
    X, Y = np.meshgrid(xs, ys, indexing="ij")
    X = np.ascontiguousarray(X); Y = np.ascontiguousarray(Y)
    got = NDArray.pow(NDArray.array(X).gpu(), NDArray.array(Y).gpu()).cpu().numpy()
    with np.errstate(all="ignore"):
        want = oracle.binary("pow", X, Y)
    assert (np.isnan(got) == np.isnan(want)).all(), list(zip(X[np.isnan(got) != np.isnan(want)], Y[np.isnan(got) != np.isnan(want)]))
    special = ~np.isnan(want) & ((want == 0) | np.isinf(want))
    bad = special & (got.view(np.uint32) != want.view(np.uint32))
    assert not bad.any(), list(zip(X[bad], Y[bad], got[bad], want[bad]))
    fin = ~np.isnan(want) & ~special
    ulp = np.abs(got[fin].view(np.int32).astype(np.int64) - want[fin].view(np.int32).astype(np.int64))
    assert ulp.max() <= 1, (X[fin][ulp.argmax()], Y[fin][ulp.argmax()])
    # wide random range, both signs, integer and fractional exponents
    n = 1_000_000
    mag = np.exp(synth.uniform((n,), 61, -80.0, 80.0).astype(np.float64)).astype(np.float32)
    sign = np.where(synth.uniform((n,), 62, 0.0, 1.0) < 0.3, f(-1), f(1))
    x = (mag * sign).astype(np.float32)
    y = synth.uniform((n,), 63, -10.0, 10.0)
    y[::3] = np.rint(y[::3])
    got = NDArray.pow(NDArray.array(x).gpu(), NDArray.array(y).gpu()).cpu().numpy()
    with np.errstate(all="ignore"):
        want = oracle.binary("pow", x, y)
    assert (np.isnan(got) == np.isnan(want)).all()
    ok = ~np.isnan(want)
    ulp = np.abs(got[ok].view(np.int32).astype(np.int64) - want[ok].view(np.int32).astype(np.int64))
    assert ulp.max() <= 1, (x[ok][ulp.argmax()], y[ok][ulp.argmax()], got[ok][ulp.argmax()], want[ok][ulp.argmax()])
    assert (ulp == 0).mean() > 0.995


@pytest.mark.parametrize("n", [1, 3, 4, 63, 255, 256, 257, 1000, 4099, 65536 + 5, 1_000_003])
def test_pow_partial_last_waves(n, hip, oracle):
    """The log2 table of the pow kernel lives in the lanes of each wave (ds_bpermute): sizes that leave the last wave
    partly past the end, with the rare cases (negative base, zero, inf, NaN) sprinkled in, against glibc's powf and bit
    for bit against the LDS-table form of the same kernel (variant 9000); row / column / scalar exponents too."""
    from numpower_amd import device as D
    from numpower_amd._lib import check, load
    lib = load()
    x = synth.uniform((n,), 11, 0.01, 4.0)
    y = synth.uniform((n,), 12, -3.0, 3.0)
    if n > 256:
        x[5], x[77], x[200] = -2.0, 0.0, np.inf
        y[5], y[78], y[201] = 3.0, np.nan, -np.inf
    da, db = D.DeviceArray.from_host(x), D.DeviceArray.from_host(y)
    got = D.binary("pow", da, "full", db, "full", 1, n).to_host().reshape(-1)
    check(lib.np_elementwise_set_variant(9000))
    try:
        lds = D.binary("pow", da, "full", db, "full", 1, n).to_host().reshape(-1)
    finally:
        check(lib.np_elementwise_set_variant(0))
    assert_bit_equal(got, lds, "register table vs LDS table, n = %d" % n)
    with np.errstate(all="ignore"):
        want = oracle.binary("pow", x, y)
    assert (np.isnan(got) == np.isnan(want)).all()
    ok = ~np.isnan(want)
    ulp = np.abs(got[ok].view(np.int32).astype(np.int64) - want[ok].view(np.int32).astype(np.int64))
    assert ulp.max() <= 1
    if n >= 1000:   # broadcast kinds of the same kernel: rows x cols with cols % 4 == 0 (vector path)
        rows, cols = n // 8, 8
        X = np.ascontiguousarray(x[:rows * cols].reshape(rows, cols))
        dX = D.DeviceArray.from_host(X)
        for kind, operand in (("row", y[:cols].copy()), ("col", y[:rows].copy()), ("scalar", y[:1].copy())):
            dy = D.DeviceArray.from_host(operand)
            got = D.binary("pow", dX, "full", dy, kind, rows, cols).to_host().reshape(rows, cols)
            full = {"row": operand[None, :], "col": operand[:, None], "scalar": operand[0]}[kind]
            with np.errstate(all="ignore"):
                want = oracle.binary("pow", X, np.ascontiguousarray(np.broadcast_to(full, X.shape)).astype(np.float32))
            assert (np.isnan(got) == np.isnan(want)).all(), kind
            ok = ~np.isnan(want)
            ulp = np.abs(got[ok].view(np.int32).astype(np.int64) - want[ok].view(np.int32).astype(np.int64))
            assert ulp.max() <= 1, kind


def test_pow_scalar_two_is_a_square(hip, oracle):
    """`$a ** 2` takes the multiply kernel: same values as glibc's powf(x, 2) up to its own rounding slack
    (<= 1 ulp), same zeros / infinities / NaNs."""
    from numpower_amd.ndarray import NDArray
    x = synth.uniform((300_007,), 64, -1e3, 1e3)
    x[:8] = [0.0, -0.0, np.inf, -np.inf, np.nan, 1e-30, -3e19, 1e-23]
    got = (NDArray.array(x).gpu() ** 2.0).cpu().numpy()
    with np.errstate(all="ignore"):
        want = oracle.binary("pow", x, np.float32(2.0))
    assert (np.isnan(got) == np.isnan(want)).all()
    ok = ~np.isnan(want)
    ulp = np.abs(got[ok].view(np.int32).astype(np.int64) - want[ok].view(np.int32).astype(np.int64))
    assert ulp.max() <= 1 and (ulp == 0).mean() > 0.995    # measured 0.9989: powf(x, 2) is not always correctly rounded


def test_binary_errors(hip):
    from numpower_amd.ndarray import Error, NDArray
    a = NDArray.array(np.ones((4, 6), np.float32)).gpu()
    b = NDArray.array(np.ones((5,), np.float32)).gpu()
    with pytest.raises(Error, match="Can't broadcast arrays."):
        a + b
    with pytest.raises(Error, match="Device mismatch, both NDArray MUST be in the same device."):
        a + NDArray.array(np.ones((4, 6), np.float32))


# ---------------------------------------------------------------------------------------------
# 3. unary ops vs the oracle
# ---------------------------------------------------------------------------------------------

def _unary_input(op, shape=(1000, 1003)):
    lo, hi = DOMAIN.get(op, (-10.0, 10.0))
    x = synth.uniform(shape, 40 + len(op), lo, hi)
    flat = x.reshape(-1)
    if op not in DOMAIN or op in ("sqrt",):
        flat[::97] = 0.0
    if op in ("rint", "round", "fix", "floor", "ceil", "trunc"):
        flat[1::5] = np.float32(0.5) + np.arange(flat[1::5].size, dtype=np.float32) - 100
    return x


@pytest.mark.parametrize("op", EXACT_UNARY)
def test_unary_exact(op, hip, oracle):
    from numpower_amd.ndarray import NDArray
    x = _unary_input(op)
    got = NDArray._unary(op, NDArray.array(x).gpu()).cpu().numpy()
    assert_bit_equal(got, oracle.unary(op, x), op)


@pytest.mark.parametrize("op", TRANSCENDENTAL_UNARY)
def test_unary_transcendental(op, hip, oracle):
    from numpower_amd.ndarray import NDArray
    x = _unary_input(op)
    got = NDArray._unary(op, NDArray.array(x).gpu()).cpu().numpy()
    assert_close(got, oracle.unary(op, x), op)


def test_clip_and_round_exact(hip, oracle):
    from numpower_amd.ndarray import NDArray
    x = _unary_input("round")
    gx = NDArray.array(x).gpu()
    assert_bit_equal(NDArray.clip(gx, -1.5, 2.25).cpu().numpy(), oracle.unary("clip", x, -1.5, 2.25), "clip")
    for d in (0, 1, 2, 3, -1):
        assert_bit_equal(NDArray.round(gx, d).cpu().numpy(), oracle.unary("round", x, d), "round %d" % d)


def test_unary_special_values(hip, oracle):
    """NaN / inf / -0 / out-of-domain inputs (sqrt(-x) -> NAN as tests/math/028 expects)."""
    from numpower_amd.ndarray import NDArray
    x = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-40, -1e-40, 3.4e38, 0.5, -0.5, 2.5, -2.5],
                 dtype=np.float32)
    gx = NDArray.array(x).gpu()
    for op in ["abs", "sqrt", "negate", "sign", "positive", "floor", "ceil", "trunc", "fix", "rint", "reciprocal"]:
        assert_bit_equal(NDArray._unary(op, gx).cpu().numpy(), oracle.unary(op, x), op + " specials")


# ---------------------------------------------------------------------------------------------
# 4. reductions
# ---------------------------------------------------------------------------------------------

@pytest.mark.parametrize("shape", [(1000, 1000), (257, 1001), (7,), (3, 5, 64), (1, 1)])
def test_full_reductions(shape, hip, oracle):
    from numpower_amd.ndarray import NDArray
    x = synth.uniform(shape, 21, 0.0, 1.0)
    gx = NDArray.array(x).gpu()
    exact = x.astype(np.float64)
    for op, ref64 in [("sum", exact.sum()), ("min", exact.min()), ("max", exact.max())]:
        got = NDArray._reduce(op, gx, None)
        ref = float(oracle.reduce_all(op, x))
        if op in ("min", "max"):
            assert got == ref
        else:
            # both within tolerance of the fp64 sum; the oracle's sequential order is the looser one
            assert abs(got - ref64) <= REL_TOL * abs(ref64)
            assert abs(got - ref) <= 2e-5 * abs(ref64) + abs(ref - ref64)
    w = min(0.1, 1.0 / np.sqrt(x.size))   # keeps the product of x.size factors inside fp32 range
    p = synth.uniform(shape, 22, 1.0 - w, 1.0 + w)
    got = NDArray.prod(NDArray.array(p).gpu())
    ref64 = float(np.prod(p.astype(np.float64)))
    # When two fp32 numbers 1+a, 1+b that sit exactly on the fp32 lattice near 1 are multiplied, the
    # second-order term a*b is below half an ulp and is dropped; a pairwise (tree) order does that
    # for all n/2 first-level pairs, a sequential order does not.  The dropped terms add up to a
    # random ~sqrt(n) * w^2 relative difference (7.8e-4 at n = 1e6, reproduced with a numpy
    # pairwise product on the CPU): the bar scales with it.
    tol = max(1e-4, 4.0 * np.sqrt(x.size) * w * w)
    assert abs(got - ref64) <= tol * abs(ref64)
    assert abs(float(oracle.reduce_all("prod", p)) - ref64) <= tol * abs(ref64)
    m = NDArray.mean(gx)
    assert abs(m - exact.mean()) <= REL_TOL * abs(exact.mean())


@pytest.mark.parametrize("shape,axis", [((1000, 1000), 0), ((1000, 1000), 1), ((257, 1001), 0), ((257, 1001), 1),
                                        ((7, 300, 64), 0), ((7, 300, 64), 1), ((7, 300, 64), 2), ((5, 8), 0),
                                        ((4096, 5000), 1), ((40, 33), 1), ((16, 8, 4, 12), 2)])
def test_axis_sum_and_mean(shape, axis, hip, oracle):
    from numpower_amd.ndarray import NDArray
    x = synth.uniform(shape, 31, 0.0, 1.0)
    gx = NDArray.array(x).gpu()
    ref64 = x.astype(np.float64).sum(axis)
    got = NDArray.sum(gx, axis)
    got = got.cpu().numpy() if not isinstance(got, float) else np.float32(got)
    assert_close(got, ref64, "sum axis %d of %s vs fp64" % (axis, shape))
    ref = oracle.reduce_axis("sum", x, axis)
    # oracle (sequential fp32, like the reference) within its own drift of the same fp64 truth
    assert np.abs(ref - ref64).max() <= 1e-4 * np.abs(ref64).max()
    gm = NDArray.mean(gx, axis)
    gm = gm.cpu().numpy() if not isinstance(gm, float) else np.float32(gm)
    assert_close(gm, x.astype(np.float64).mean(axis), "mean axis %d" % axis)


def test_axis_sum_small_is_bit_exact(hip, oracle):
    """For KAT-sized inputs (<= 2 slices) the order of additions is forced, so the GPU result
    equals the reference's sequential sum bit for bit."""
    from numpower_amd.ndarray import NDArray
    x = synth.uniform((2, 1000), 5, -100, 100)
    assert_bit_equal(NDArray.sum(NDArray.array(x).gpu(), 0).cpu().numpy(), oracle.reduce_axis("sum", x, 0), "2-row sum")


@pytest.mark.parametrize("shape,axis", [((6, 1000), 0), ((6, 1001), 0), ((300, 20), 1), ((4, 5, 16), 1)])
def test_axis_prod_zero_sign_quirk(shape, axis, hip, oracle):
    """reduce(Multiply_Float): zero results carry the sign the reference's AVX body / scalar tail
    leave behind (arithmetics.c:403,410-412); checked bit-exact on inputs whose products are
    exactly representable."""
    from numpower_amd.ndarray import NDArray
    rng = np.random.default_rng(3)
    x = rng.choice(np.array([0.0, -0.0, 1.0, -1.0, 2.0, -2.0, 0.5], dtype=np.float32), size=shape)
    got = NDArray.prod(NDArray.array(x).gpu(), axis).cpu().numpy()
    assert_bit_equal(got, oracle.reduce_axis("prod", x, axis), "prod axis %d of %s" % (axis, shape))


def test_axis_min_max(hip):
    from numpower_amd.ndarray import NDArray
    x = synth.uniform((300, 257), 8, -5, 5)
    gx = NDArray.array(x).gpu()
    for axis in (0, 1):
        assert_bit_equal(NDArray.max(gx, axis).cpu().numpy(), x.max(axis), "max axis")
        assert_bit_equal(NDArray.min(gx, axis).cpu().numpy(), x.min(axis), "min axis")


def test_axis_errors(hip):
    from numpower_amd.ndarray import Error, NDArray
    gx = NDArray.array(np.ones((3, 4), np.float32)).gpu()
    with pytest.raises(Error, match="axis 2 is out of bounds for array of dimension 2"):
        NDArray.sum(gx, 2)


# ---------------------------------------------------------------------------------------------
# 5. matmul
# ---------------------------------------------------------------------------------------------

def _gemm_check(got, A, B, what):
    ref64 = A.astype(np.float64) @ B.astype(np.float64)
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    err = np.abs(got.astype(np.float64) - ref64) / np.maximum(scale, 1e-30)
    assert err.max() <= 1e-6, "%s: max normalised err %.3g" % (what, err.max())
    return ref64


@pytest.mark.parametrize("mnk", [(2, 2, 2), (2, 1, 2), (64, 64, 64), (128, 128, 128), (100, 90, 70),
                                 (257, 129, 65), (1000, 1000, 1000), (1024, 512, 2048), (1, 1000, 1),
                                 # LDS-DMA kernel with M / N edges (clamped source rows / columns)
                                 (2000, 2004, 2000), (1000, 1000, 1008), (300, 516, 2048), (257, 4, 4096),
                                 (3, 132, 64)])
def test_matmul_vs_oracle_and_fp64(mnk, hip, oracle):
    from numpower_amd.ndarray import NDArray
    m, n, k = mnk
    A = synth.uniform((m, k), 3, -1, 1)
    B = synth.uniform((k, n), 4, -1, 1)
    got = NDArray.matmul(NDArray.array(A).gpu(), NDArray.array(B).gpu()).cpu().numpy()
    ref64 = _gemm_check(got, A, B, "matmul %s" % (mnk,))
    ref = oracle.matmul(A, B)   # OpenBLAS, the reference's CPU back end
    # 1e-5 relative to the reference, measured against the magnitude of the row.column products
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    assert (np.abs(got - ref) / np.maximum(scale, 1e-30)).max() <= REL_TOL
    assert (np.abs(ref - ref64) / np.maximum(scale, 1e-30)).max() <= REL_TOL


def test_matmul_transpose_detecting(hip):
    """A = I with an asymmetric B catches swapped operands / transposed C writes."""
    from numpower_amd.ndarray import NDArray
    n = 160
    B = (np.arange(n * n, dtype=np.float32).reshape(n, n) % 251) - 100
    got = NDArray.matmul(NDArray.array(np.eye(n, dtype=np.float32)).gpu(), NDArray.array(B).gpu()).cpu().numpy()
    assert_bit_equal(got, B, "I.B")
    got = NDArray.matmul(NDArray.array(B).gpu(), NDArray.array(np.eye(n, dtype=np.float32)).gpu()).cpu().numpy()
    assert_bit_equal(got, B, "B.I")


def test_matmul_errors_and_dot(hip, oracle):
    from numpower_amd.ndarray import Error, NDArray
    a = NDArray.array(np.ones((3, 4), np.float32)).gpu()
    with pytest.raises(Error, match=r"Shape mismatch for matmul. cols\(a\) != rows\(b\)"):
        NDArray.matmul(a, NDArray.array(np.ones((3, 4), np.float32)).gpu())
    with pytest.raises(Error, match="Stack of matrices not allowed"):
        t = NDArray.array(np.ones((2, 3, 3), np.float32)).gpu()
        NDArray.matmul(t, t)
    with pytest.raises(Error, match="Arrays must have the same shape. Broadcasting not implemented."):
        NDArray.matmul(a, NDArray.array(np.ones((4,), np.float32)).gpu())
    A = synth.uniform((300, 1000), 1, -1, 1)
    x = synth.uniform((1000,), 2, -1, 1)
    got = NDArray.dot(NDArray.array(A).gpu(), NDArray.array(x).gpu()).cpu().numpy()
    ref = oracle.matvec(A, x)
    scale = np.abs(A).astype(np.float64) @ np.abs(x).astype(np.float64)
    assert (np.abs(got - ref) / scale).max() <= REL_TOL


def test_batched_matmul(hip):
    from numpower_amd.ndarray import NDArray
    A = synth.uniform((5, 130, 70), 12, -1, 1)
    B = synth.uniform((5, 70, 200), 13, -1, 1)
    got = NDArray.batched_matmul(NDArray.array(A).gpu(), NDArray.array(B).gpu()).cpu().numpy()
    for i in range(5):
        _gemm_check(got[i], A[i], B[i], "batch %d" % i)


# ---------------------------------------------------------------------------------------------
# 6. device-buffer layer
# ---------------------------------------------------------------------------------------------

def test_placement_roundtrip_and_leak_counter(hip):
    from numpower_amd.ndarray import Error, NDArray
    before = NDArray.live_device_allocations()
    x = synth.uniform((123, 457), 9, -1, 1)
    g = NDArray.array(x).gpu()
    assert g.isGPU() and g.shape() == [123, 457]
    assert_bit_equal(g.cpu().numpy(), x, "gpu()/cpu() round trip")
    g2 = g.gpu()   # already on the GPU: a copy (ndarray.c:1043-1045)
    assert_bit_equal(g2.cpu().numpy(), x, "gpu() of a GPU array copies")
    with pytest.raises(Error, match="NDArray must be on CPU RAM before it can be converted to a PHP array."):
        g.toArray()
    row = g[5]
    assert_bit_equal(row.cpu().numpy(), x[5], "row view")
    del g, g2
    assert_bit_equal(row.cpu().numpy(), x[5], "view keeps its base alive")
    del row
    assert NDArray.live_device_allocations() == before   # vmemcheck parity (gpu_alloc.c:36-40)


def test_fill_and_zeros(hip):
    from numpower_amd.ndarray import GPU, NDArray
    z = NDArray.zeros([37, 19], GPU)
    assert not z.cpu().numpy().any()
    z.fill(2.5)
    assert (z.cpu().numpy() == np.float32(2.5)).all()
    z[3].fill(-1.0)   # unaligned view
    h = z.cpu().numpy()
    assert (h[3] == -1.0).all() and (h[2] == 2.5).all() and (h[4] == 2.5).all()


@pytest.mark.parametrize("mnk", [(100, 100, 100000), (64, 64, 4096), (37, 5, 30011), (128, 96, 2050), (1, 1, 500000),
                                 # planner cases (plan_sgemm): everything split on the 256x128 LDS-DMA tiles,
                                 # with a remainder chunk riding as the last batch entry (K_last) ...
                                 (1280, 1280, 8192), (1536, 1536, 1536),
                                 # ... and a split-K TAIL: 9 x 29 = 261 tiles on 256 CUs
                                 (2304, 3712, 1040), (520, 3000, 4100)])
def test_matmul_splitk_small_result_long_k(mnk, hip, oracle):
    """Small M x N with a long K runs as split-K (chunks of K as the batch dimension + one
    deterministic reduce, np_sgemm.hip try_splitk): same 1e-5 bar vs fp64 as every other product,
    bit-identical from run to run, and within accumulation-order noise of the plain kernel."""
    from numpower_amd import _lib
    from numpower_amd import device as D
    m, n, k = mnk
    a = synth.uniform((m, k), 31, -1.0, 1.0)
    b = synth.uniform((k, n), 32, -1.0, 1.0)
    da, db = D.DeviceArray.from_host(a), D.DeviceArray.from_host(b)
    got = D.sgemm(da, db).to_host()
    again = D.sgemm(da, db).to_host()
    assert (got.view(np.uint32) == again.view(np.uint32)).all()
    ref64 = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    assert (np.abs(got - ref64) / scale).max() <= 1e-5
    _lib.check(_lib.load().np_sgemm_set_variant(-1))
    try:
        plain = D.sgemm(da, db).to_host()
    finally:
        _lib.check(_lib.load().np_sgemm_set_variant(-2))
    assert (np.abs(plain - ref64) / scale).max() <= 1e-5
    assert (np.abs(plain.astype(np.float64) - got) / scale).max() <= 2e-6


@pytest.mark.parametrize("mnk", [(2000, 2004, 2000), (300, 516, 2048), (257, 4, 4096), (3, 132, 64), (255, 127, 16),
                                 (513, 260, 1024), (4000, 4000, 496),
                                 # K tails (K % 16 = 4, 8, 12): clamped fetch + zeroed LDS slots, incl. the
                                 # cases where the ragged tile is loaded by the prologue (K < 32)
                                 (1000, 1000, 1000), (512, 256, 4), (512, 256, 8), (512, 256, 12), (512, 256, 20),
                                 (512, 256, 28), (256, 128, 36), (300, 260, 44), (1500, 1500, 1500), (64, 64, 3000)])
def test_matmul_dma_edge_kernel(mnk, hip):
    """sgemm_dma_kernel<EDGE> forced (variant 7): M / N not multiples of the 256 x 128 tile — source
    rows / columns are clamped, the garbage only reaches rows >= M / columns >= N that the guarded
    epilogue never stores.  Checked against fp64 AND bit-for-bit against the 128 x 128 kernel's
    neighbours: nothing outside C may be written (a canary frame around the output)."""
    from numpower_amd import _lib
    from numpower_amd import device as D
    lib = _lib.load()
    m, n, k = mnk
    a = synth.uniform((m, k), 33, -1.0, 1.0)
    b = synth.uniform((k, n), 34, -1.0, 1.0)
    # A and B sit at the very end of poisoned allocations: a K tail that read past its row would pick up
    # NaNs (and the clamped fetches must stay inside the buffers)
    da, db = D.DeviceArray.from_host(a), D.DeviceArray.from_host(b)
    pad = 4096                                             # floats of canary before and after C
    frame = D.DeviceArray((m * n + 2 * pad,))
    D.fill(frame, -777.0)
    _lib.check(lib.np_sgemm_set_variant(7))
    try:
        _lib.check(lib.np_sgemm(m, n, k, da.ptr, db.ptr, frame.ptr + 4 * pad))
    finally:
        _lib.check(lib.np_sgemm_set_variant(0))
    host = frame.to_host().reshape(-1)
    assert (host[:pad] == -777.0).all() and (host[pad + m * n:] == -777.0).all()
    got = host[pad:pad + m * n].reshape(m, n)
    ref64 = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    assert (np.abs(got - ref64) / scale).max() <= 1e-6


@pytest.mark.parametrize("mnk", [(2049, 2051, 2053), (1000, 1000, 3001), (1537, 1538, 1540), (4097, 130, 4097),
                                 (2100, 2102, 1001), (5, 3, 7), (300, 1, 100)])
def test_matmul_padded_operands(mnk, hip):
    """K % 16 != 0 / rows that are not 16-byte aligned: large products copy A and B once into
    zero-padded aligned workspaces and run the LDS-DMA kernel (launch_padded, np_sgemm.hip); C is
    written in place with its real row length.  fp64 bar + canary frame around C, and the same
    product with the path off (variant -1: whole-K plans on the original operands) agrees."""
    from numpower_amd import _lib
    from numpower_amd import device as D
    lib = _lib.load()
    m, n, k = mnk
    a = synth.uniform((m, k), 35, -1.0, 1.0)
    b = synth.uniform((k, n), 36, -1.0, 1.0)
    da, db = D.DeviceArray.from_host(a), D.DeviceArray.from_host(b)
    pad = 4096
    frame = D.DeviceArray((m * n + 2 * pad,))
    D.fill(frame, -777.0)
    _lib.check(lib.np_sgemm_set_variant(-3))          # take the padded path whatever the model says
    try:
        _lib.check(lib.np_sgemm(m, n, k, da.ptr, db.ptr, frame.ptr + 4 * pad))
    finally:
        _lib.check(lib.np_sgemm_set_variant(-2))
    host = frame.to_host().reshape(-1)
    assert (host[:pad] == -777.0).all() and (host[pad + m * n:] == -777.0).all()
    got = host[pad:pad + m * n].reshape(m, n)
    ref64 = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    assert (np.abs(got - ref64) / scale).max() <= 1e-6
    _lib.check(lib.np_sgemm_set_variant(-1))
    try:
        plain = D.sgemm(da, db).to_host()
    finally:
        _lib.check(lib.np_sgemm_set_variant(-2))
    assert (np.abs(plain.astype(np.float64) - got) / scale).max() <= 2e-6


UNALIGNED_SHAPES = [(300, 5, 5), (300, 6, 7), (257, 7, 6), (1, 4, 4), (255, 129, 17), (256, 131, 18), (513, 258, 19),
                    (300, 1001, 21), (64, 7, 33), (700, 133, 47), (1000, 1001, 1001), (1001, 1003, 1005), (130, 4097, 31),
                    (2049, 2051, 2053), (3001, 2999, 1003), (512, 256, 1001), (512, 257, 1024), (4097, 4097, 4097)]


@pytest.mark.parametrize("form", ["tile", "streamk", "default"])
@pytest.mark.parametrize("mnk", UNALIGNED_SHAPES, ids=["%dx%dx%d" % s for s in UNALIGNED_SHAPES])
def test_matmul_dma_unaligned_operands(mnk, form, hip):
    """K % 4, N % 4 != 0, odd base addresses: the LDS-DMA kernel takes the operands as they are (4-byte-aligned
    global_load_lds; a chunk cut by the end of a row is fetched so that it ends WITH the row and moved into place in LDS
    in the last K-tile).  A and B sit at odd float offsets inside NaN-filled allocations — a K tail that was not zeroed,
    or anything read from outside the operands and used, turns up as NaN — C inside a canary frame; the tile form
    (variant 7), stream-K (-4) and the default planner; fp64 bar, and agreement with the pad-copy path (-6)."""
    from numpower_amd import _lib
    from numpower_amd import device as D
    lib = _lib.load()
    m, n, k = mnk
    a = synth.uniform((m, k), 37, -1.0, 1.0)
    b = synth.uniform((k, n), 38, -1.0, 1.0)
    pad = 1024
    frames = []
    ptrs = []
    for off, mat in ((1, a), (3, b)):
        frame = D.DeviceArray((mat.size + 2 * pad + 4,))
        D.fill(frame, float("nan"))
        where = frame.ptr + 4 * (pad + off)
        _lib.check(lib.np_memcpy_h2d(where, mat.ctypes.data, mat.nbytes))
        frames.append(frame)
        ptrs.append(where)
    out = D.DeviceArray((m * n + 2 * pad,))
    D.fill(out, -777.0)
    variant = {"tile": 7, "streamk": -4, "default": None}[form]
    if variant is not None:
        _lib.check(lib.np_sgemm_set_variant(variant))
    try:
        _lib.check(lib.np_sgemm(m, n, k, ptrs[0], ptrs[1], out.ptr + 4 * pad))
    finally:
        _lib.check(lib.np_sgemm_set_variant(0))
        _lib.check(lib.np_sgemm_set_variant(-2))
    host = out.to_host().reshape(-1)
    assert (host[:pad] == -777.0).all() and (host[pad + m * n:] == -777.0).all()
    got = host[pad:pad + m * n].reshape(m, n)
    assert not np.isnan(got).any()
    ref64 = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    assert (np.abs(got - ref64) / scale).max() <= 1e-6
    if form == "default":
        _lib.check(lib.np_sgemm_set_variant(-6))      # as before round 3: such operands go through padded copies
        try:
            _lib.check(lib.np_sgemm(m, n, k, ptrs[0], ptrs[1], out.ptr + 4 * pad))
        finally:
            _lib.check(lib.np_sgemm_set_variant(-7))
        padded = out.to_host().reshape(-1)[pad:pad + m * n].reshape(m, n)
        assert (np.abs(padded.astype(np.float64) - got) / scale).max() <= 2e-6
    for f in frames + [out]:
        f.free()


@pytest.mark.parametrize("mnk", [(1, 4097, 4097), (2, 1000, 3000), (3, 257, 5000), (5, 4099, 300), (8, 8192, 1024), (7, 300, 100000),
                                 (4, 1024, 1024), (1, 256, 4000), (6, 70001, 256), (8, 1031, 1031),
                                 # 9 ... 64 rows: sgemm_skinny_kernel (the rows on the matrix cores, B read once)
                                 (9, 4097, 4097), (16, 8192, 2048), (31, 4099, 4001), (32, 5000, 3333), (33, 4096, 4096), (64, 4097, 4100),
                                 (50, 70001, 513), (64, 512, 40000), (17, 100003, 600)])
def test_matmul_few_rows(mnk, hip, oracle):
    """M <= 64 rows against a large B (vector . matrix, the row edge of a peeled product, a small batch against a weight
    matrix): sgemm_fewrows_kernel (M <= 8) / sgemm_skinny_kernel (MFMA, M <= 64) — B read once, K cut into chunks whose
    partial sums are folded in chunk order.  fp64 bar, the oracle's OpenBLAS product at
    1e-5 |A|.|B|, a canary frame around C, bit-identical when repeated, and the tiled kernels (variant -12) agree."""
    from numpower_amd import _lib
    from numpower_amd import device as D
    lib = _lib.load()
    m, n, k = mnk
    a = synth.uniform((m, k), 43, -1.0, 1.0)
    b = synth.uniform((k, n), 44, -1.0, 1.0)
    da, db = D.DeviceArray.from_host(a), D.DeviceArray.from_host(b)
    pad = 1024
    frame = D.DeviceArray((m * n + 2 * pad,))
    outs = []
    for _ in range(2):
        D.fill(frame, -777.0)
        _lib.check(lib.np_sgemm(m, n, k, da.ptr, db.ptr, frame.ptr + 4 * pad))
        host = frame.to_host().reshape(-1)
        assert (host[:pad] == -777.0).all() and (host[pad + m * n:] == -777.0).all()
        outs.append(host[pad:pad + m * n].reshape(m, n).copy())
    assert (outs[0].view(np.uint32) == outs[1].view(np.uint32)).all()
    got = outs[0]
    ref64 = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    assert (np.abs(got - ref64) / scale).max() <= 1e-6
    assert (np.abs(got - oracle.matmul(a, b)) <= 1e-5 * scale).all()
    _lib.check(lib.np_sgemm_set_variant(-12))
    try:
        tiled = D.sgemm(da, db).to_host()
    finally:
        _lib.check(lib.np_sgemm_set_variant(-13))
    assert (np.abs(tiled.astype(np.float64) - got) / scale).max() <= 2e-6


@pytest.mark.parametrize("mnk", [(2305, 2178, 77), (2049, 2304, 100), (2304, 2049, 64), (2312, 2050, 33), (4097, 4097, 600),
                                 (2320, 4100, 4100), (2336, 4096, 4097)])
@pytest.mark.parametrize("mode", ["forced", "default"])
def test_matmul_peeled_edges(mnk, mode, hip):
    """A thin ragged edge (M % 256 <= 32 rows, N % 128 <= 2 columns) of a large product is peeled off: whole tiles for
    the main block — C and B addressed as windows of the full matrices — and thin products for the edges (try_peeled,
    np_sgemm.hip).  Forced (variant -11) on shapes the model would leave alone, and the default on one it takes;
    fp64 bar, a canary frame around C, agreement with the unpeeled product (-9)."""
    from numpower_amd import _lib
    from numpower_amd import device as D
    lib = _lib.load()
    m, n, k = mnk
    a = synth.uniform((m, k), 41, -1.0, 1.0)
    b = synth.uniform((k, n), 42, -1.0, 1.0)
    da, db = D.DeviceArray.from_host(a), D.DeviceArray.from_host(b)
    pad = 4096
    frame = D.DeviceArray((m * n + 2 * pad,))
    D.fill(frame, -777.0)
    _lib.check(lib.np_sgemm_set_variant(-11 if mode == "forced" else -10))
    try:
        _lib.check(lib.np_sgemm(m, n, k, da.ptr, db.ptr, frame.ptr + 4 * pad))
    finally:
        _lib.check(lib.np_sgemm_set_variant(-10))
    host = frame.to_host().reshape(-1)
    assert (host[:pad] == -777.0).all() and (host[pad + m * n:] == -777.0).all()
    got = host[pad:pad + m * n].reshape(m, n)
    ref64 = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    assert (np.abs(got - ref64) / scale).max() <= 1e-6
    _lib.check(lib.np_sgemm_set_variant(-9))
    try:
        whole = D.sgemm(da, db).to_host()
    finally:
        _lib.check(lib.np_sgemm_set_variant(-10))
    assert (np.abs(whole.astype(np.float64) - got) / scale).max() <= 2e-6
    for x in (da, db, frame):
        x.free()


def test_batched_matmul_unaligned_strides(hip):
    """The batched entry with odd matrix sizes (every matrix starts at an odd float offset) — one launch of the LDS-DMA
    kernel over blockIdx.z — and the progress-reporting form np_comm's pipeline uses."""
    from numpower_amd import _lib
    from numpower_amd import device as D
    lib = _lib.load()
    batch, m, n, k = 5, 259, 131, 77
    a = synth.uniform((batch, m, k), 39, -1.0, 1.0)
    b = synth.uniform((batch, k, n), 40, -1.0, 1.0)
    da, db = D.DeviceArray.from_host(a), D.DeviceArray.from_host(b)
    out = D.DeviceArray((batch, m, n))
    _lib.check(lib.np_sgemm_strided_batched(batch, m, n, k, da.ptr, m * k, db.ptr, k * n, out.ptr, m * n))
    got = out.to_host().reshape(batch, m, n)
    ref64 = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    assert (np.abs(got - ref64) / scale).max() <= 1e-6


def _log_sweep(lo, hi, n, seed):
    """n values log-uniform in [lo, hi]."""
    u = synth.uniform((n,), seed, 0.0, 1.0).astype(np.float64)
    return np.exp(np.log(lo) + u * (np.log(hi) - np.log(lo))).astype(np.float32)


@pytest.mark.parametrize("op", ["log1p", "sinh", "cosh", "arcsinh", "arccosh", "arctanh"])
def test_fast_hyperbolic_accuracy(op, hip, oracle):
    """log1p and the hyperbolic family are hand-written around expf / logf / sqrtf (the device
    library's versions are VALU-bound at a third of the HBM roofline): whole-range accuracy against
    fp64 (a few ulp: 2e-6 relative) and against the oracle's glibc (the 1e-5 bar), and glibc's
    results at the special values."""
    from numpower_amd.ndarray import NDArray
    f64 = {"log1p": np.log1p, "sinh": np.sinh, "cosh": np.cosh, "arcsinh": np.arcsinh, "arccosh": np.arccosh,
           "arctanh": np.arctanh}[op]
    n = 200_000
    if op == "log1p":
        parts = [_log_sweep(1e-38, 1e-6, n, 1), _log_sweep(1e-6, 1.0, n, 2), _log_sweep(1.0, 3e38, n, 3),
                 -_log_sweep(1e-38, 1e-3, n, 4), -_log_sweep(1e-3, 0.999999, n, 5)]
    elif op in ("sinh", "cosh"):
        pos = [_log_sweep(1e-38, 1e-3, n, 1), _log_sweep(1e-3, 0.5, n, 2), _log_sweep(0.5, 88.0, n, 3),
               synth.uniform((n,), 4, 88.0, 89.4), synth.uniform((n,), 5, 0.49, 0.51)]
        parts = pos + [-p for p in pos]
    elif op == "arcsinh":
        pos = [_log_sweep(1e-38, 1e-3, n, 1), _log_sweep(1e-3, 10.0, n, 2), _log_sweep(10.0, 3e8, n, 3),
               _log_sweep(2e8, 3e38, n, 4)]
        parts = pos + [-p for p in pos]
    elif op == "arccosh":
        parts = [np.float32(1.0) + _log_sweep(1e-7, 1.0, n, 1), _log_sweep(2.0, 3e8, n, 2), _log_sweep(2e8, 3e38, n, 3)]
    else:
        pos = [_log_sweep(1e-38, 1e-3, n, 1), _log_sweep(1e-3, 0.9, n, 2), np.float32(1.0) - _log_sweep(6e-8, 0.1, n, 3)]
        parts = pos + [-p for p in pos]
    x = np.concatenate(parts).astype(np.float32)
    got = NDArray._unary(op, NDArray.array(x).gpu()).cpu().numpy().astype(np.float64)
    with np.errstate(all="ignore"):
        want = f64(x.astype(np.float64))
    finite = np.isfinite(want) & (np.abs(want) < 3.4e38)
    rel = np.abs(got[finite] - want[finite]) / np.maximum(np.abs(want[finite]), 1e-300)
    assert rel.max() <= 2e-6, "%s: max rel err vs fp64 %.3g at x = %r" % (op, rel.max(), x[finite][rel.argmax()])
    ref = oracle.unary(op, x).astype(np.float64)
    ok = np.isfinite(ref)
    rel = np.abs(got[ok] - ref[ok]) / np.maximum(np.abs(ref[ok]), 1e-300)
    assert rel.max() <= 1e-5
    assert (np.isinf(ref) == np.isinf(got)).all()          # overflow to inf happens at the same inputs
    # special values: exactly glibc's answers (sign of zero included; NaN where glibc says NaN)
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 0.5, -0.5, 2.0, -2.0, 89.5, -89.5,
                   3.4e38, -3.4e38, 0.99999994, -0.99999994, 1.0000001], dtype=np.float32)
    g = NDArray._unary(op, NDArray.array(sp).gpu()).cpu().numpy()
    r = oracle.unary(op, sp)
    assert (np.isnan(g) == np.isnan(r)).all(), (op, sp[np.isnan(g) != np.isnan(r)])
    m = ~np.isnan(r)
    assert (np.isinf(g[m]) == np.isinf(r[m])).all() and (np.signbit(g[m]) == np.signbit(r[m])).all()
    fin = m & np.isfinite(r)
    assert np.allclose(g[fin], r[fin], rtol=1e-5, atol=0.0)


@pytest.mark.parametrize("shape,axis", [((3, 300_000), 1), ((300_000, 3), 0), ((1_000_003, 1), 0), ((1, 1_000_003), 1),
                                        ((5, 70_000, 3), 1), ((2, 40_000, 17), 1), ((1001, 1003), 0), ((4001, 250), 0),
                                        ((9, 20_011), 1), ((200_000, 2), 0), ((64, 5000, 64), 1), ((7, 600, 1), 1),
                                        # very many short rows: slabs staged through LDS (reduce_rows_staged), incl. a ragged
                                        # last slab and rows * len % 4 != 0; ragged wide rows through the float4 column kernel
                                        ((900_001, 10), 1), ((1_700_000, 5), 1), ((530_003, 16), 1), ((180_001, 47), 1),
                                        ((3000, 4001), 0), ((2100, 10_007), 0)])
def test_axis_reduce_few_outputs_long_axis(shape, axis, hip, oracle):
    """Shapes where the outputs alone cannot fill the machine — a few long rows, column sums of an
    N x 3 array, ragged inner sizes: the axis is cut into chunks / flat slabs (reduce_rows_block
    chunks, reduce_small_inner, chunked reduce_axis_generic in np_reduce.hip) and the partials are
    folded by a second pass.  sum / mean vs fp64, min / max exact, prod's zero-sign quirk vs the
    oracle."""
    from numpower_amd.ndarray import NDArray
    x = synth.uniform(shape, 41, -1.0, 1.0)
    gx = NDArray.array(x).gpu()
    ref64 = x.astype(np.float64).sum(axis)
    scale = np.abs(x).astype(np.float64).sum(axis)

    def host(v):
        return v.cpu().numpy() if not isinstance(v, float) else np.float32(v)
    got = host(NDArray.sum(gx, axis))
    assert got.shape == ref64.shape
    assert (np.abs(got - ref64) <= 1e-5 * scale).all()
    gm = host(NDArray.mean(gx, axis))
    assert (np.abs(gm - ref64 / shape[axis]) <= 1e-5 * scale / shape[axis]).all()
    assert_bit_equal(host(NDArray.max(gx, axis)), x.max(axis), "max")
    assert_bit_equal(host(NDArray.min(gx, axis)), x.min(axis), "min")
    # prod: values whose products are exact, zeros of both signs
    rng = np.random.default_rng(5)
    p = rng.choice(np.array([1.0, -1.0, 1.0, 1.0], dtype=np.float32), size=shape)
    p.reshape(-1)[::997] = 0.0
    p.reshape(-1)[5::1009] = -0.0
    small = tuple(min(s, 3000) if i == axis else s for i, s in enumerate(shape))   # oracle's slice-by-slice reduce() is slow
    p = np.ascontiguousarray(p[tuple(slice(0, s) for s in small)])
    got = host(NDArray.prod(NDArray.array(p).gpu(), axis))
    assert_bit_equal(got, oracle.reduce_axis("prod", p, axis), "prod")


@pytest.mark.parametrize("mn", [(1, 3_000_001), (1, 65536), (10, 500_000), (7, 16385), (200_000, 10), (5000, 1), (4096, 33),
                                # 2 .. 16 long rows: a workgroup per chunk of x takes all rows (sgemv_fewrows_chunks_kernel, round 5)
                                (2, 1_000_003), (16, 300_001), (3, 65536), (13, 70_002), (17, 200_000),
                                (300, 70_001), (50_000, 100), (20_000, 256), (20_000, 257), (100_000, 8), (100_000, 9),
                                # LDS-staged slabs of short rows (M >= 262144, 4 <= N <= 63), incl. a ragged last slab and M * N % 4 != 0
                                (300_000, 10), (262_144, 4), (262_145, 63), (400_003, 7), (1_000_000, 16), (270_001, 33)])
def test_sgemv_shapes(mn, hip, oracle):
    """np_sgemv / NDArray_Dot outside the square case: inner products and few long rows (chunked:
    sgemv_chunks_kernel + fold), many short rows (thread per row), against fp64 and the oracle's
    OpenBLAS sgemv."""
    from numpower_amd.ndarray import NDArray
    m, n = mn
    a = synth.uniform((m, n), 37, -1.0, 1.0)
    x = synth.uniform((n,), 38, -1.0, 1.0)
    ga, gx = NDArray.array(a).gpu(), NDArray.array(x).gpu()
    got = NDArray.dot(ga, gx).cpu().numpy()
    ref64 = a.astype(np.float64) @ x.astype(np.float64)
    scale = np.abs(a).astype(np.float64) @ np.abs(x).astype(np.float64)
    assert got.shape == (m,)
    assert (np.abs(got - ref64) <= 1e-5 * scale).all()
    assert (np.abs(oracle.matvec(a, x) - ref64) <= 1e-5 * scale).all()
    if m == 1:   # 1-D . 1-D -> NDArray_Inner
        v = NDArray.dot(NDArray.array(a[0]).gpu(), gx)
        assert abs(float(v) - ref64[0]) <= 1e-5 * scale[0]


def test_batched_matmul_more_than_65535_matrices(hip):
    """blockIdx.z carries the batch index; 70 000 small matrices go in slabs (np_sgemm_strided_batched)."""
    from numpower_amd import _lib
    from numpower_amd import device as D
    batch, m, n, k = 70_000, 4, 5, 6
    a = synth.uniform((batch, m, k), 39, -1.0, 1.0)
    b = synth.uniform((batch, k, n), 40, -1.0, 1.0)
    da, db, dc = D.DeviceArray.from_host(a), D.DeviceArray.from_host(b), D.DeviceArray((batch, m, n))
    _lib.check(_lib.load().np_sgemm_strided_batched(batch, m, n, k, da.ptr, m * k, db.ptr, k * n, dc.ptr, m * n))
    got = dc.to_host()
    ref = np.einsum("bmk,bkn->bmn", a.astype(np.float64), b.astype(np.float64))
    assert np.abs(got - ref).max() <= 1e-5


@pytest.mark.parametrize("cols", [5, 7, 8, 9, 16, 24, 33, 64, 100, 129, 255, 256, 257, 511])
def test_row_reductions_short_rows(cols, hip, oracle):
    """Rows of 5 .. ~500 elements: L lanes per row (reduce_rows_group), 257+ a wave per row."""
    from numpower_amd.ndarray import NDArray
    rows = 3001
    x = synth.uniform((rows, cols), 43, -1.0, 1.0)
    gx = NDArray.array(x).gpu()
    ref64 = x.astype(np.float64).sum(1)
    scale = np.abs(x).astype(np.float64).sum(1)
    got = NDArray.sum(gx, 1).cpu().numpy()
    assert (np.abs(got - ref64) <= 1e-5 * scale).all()
    assert (np.abs(NDArray.mean(gx, 1).cpu().numpy() - ref64 / cols) <= 1e-5 * scale / cols).all()
    assert_bit_equal(NDArray.max(gx, 1).cpu().numpy(), x.max(1), "max")
    assert_bit_equal(NDArray.min(gx, 1).cpu().numpy(), x.min(1), "min")
    p = np.where(synth.uniform((rows, cols), 44, 0.0, 1.0) < 0.02, np.float32(-0.0), np.float32(1.0)).astype(np.float32)
    p[::3] = np.abs(p[::3])
    assert_bit_equal(NDArray.prod(NDArray.array(p).gpu(), 1).cpu().numpy(), oracle.reduce_axis("prod", p, 1), "prod")


@pytest.mark.parametrize("mnk", [(100_000, 3, 3), (50_000, 10, 784), (20_000, 16, 16), (30_000, 8, 64), (5000, 1, 64), (4097, 32, 100),
                                 (2048, 5, 7), (3, 3, 1_000_000), (10, 7, 300_001), (64, 32, 70_000), (1, 1, 200_000), (2047, 3, 3),
                                 # tiny M, wide N: sgemm_thin_left_kernel
                                 (3, 1_000_003, 3), (4, 400_000, 4), (8, 70_001, 64), (1, 65536, 8), (5, 100_002, 17), (16, 66_000, 33),
                                 (10, 100_000, 16),
                                 # N = 5..32, K % 4 == 0: sgemm_thin_mfma_kernel (16x16x4 for N <= 16, 32x32x2 above); K tails
                                 # inside a 32-step and a 64-chunk, M not a multiple of the wave's rows
                                 (2049, 31, 36), (5001, 17, 8), (3001, 12, 132), (2500, 9, 64), (2100, 16, 96), (70_001, 24, 1024),
                                 (2048, 5, 12), (9999, 32, 260), (100_003, 6, 4),
                                 (50_000, 10, 785), (3000, 20, 7), (4099, 16, 33), (2050, 30, 66), (2500, 8, 5),
                                 # 17 <= M < 2048 against a very long K: the same kernel over K-chunks + a fold (X^T . G)
                                 (784, 10, 200_000), (500, 5, 30_001), (100, 17, 16_384), (2047, 3, 20_000), (17, 32, 100_000), (300, 16, 65_537),
                                 # ... and N = 33..64 there (two 32-column blocks): cluster sums H^T . X, Gram matrices of <= 64 features
                                 (32, 64, 300_000), (64, 64, 100_001), (100, 33, 20_000), (2047, 50, 16_384), (40, 63, 70_000),
                                 # fewer row tiles than four waves: 128- and 64-thread workgroups
                                 (20, 8, 50_000), (30, 16, 20_000), (32, 32, 65_536), (33, 20, 40_000), (48, 12, 30_000),
                                 # 2048 <= M, too few rows to fill the machine, a long K: the MFMA thin kernel over K-chunks + a fold;
                                 # N = 1 there: the matrix . vector kernel (rows of any alignment)
                                 (4096, 8, 4097), (4096, 2, 4097), (8192, 16, 8192), (3000, 31, 5001), (2048, 5, 1024), (20_000, 24, 3000),
                                 (4096, 1, 4097), (2048, 1, 2049), (5000, 1, 1001), (3000, 1, 256), (2500, 1, 4098)])
def test_matmul_thin(mnk, hip, oracle):
    """N <= 32: GEMV-with-several-right-hand-sides kernels (sgemm_thin_kernel: lane groups per row of A;
    sgemm_thin_mfma_kernel: one wave = 16 / 32 rows on the MFMA with B through LDS; sgemm_thin_chunks_kernel:
    few rows, long K) instead of mostly empty 64-wide tiles."""
    from numpower_amd.ndarray import NDArray
    m, n, k = mnk
    a = synth.uniform((m, k), 45, -1.0, 1.0)
    b = synth.uniform((k, n), 46, -1.0, 1.0)
    got = NDArray.matmul(NDArray.array(a).gpu(), NDArray.array(b).gpu()).cpu().numpy()
    ref64 = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    assert got.shape == (m, n)
    assert (np.abs(got - ref64) <= 1e-6 * np.maximum(scale, 1e-30)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [70_000, 300_001, 1_000_000, 1_048_576, 1_100_003])
def test_full_reduction_one_launch_is_bit_identical_to_two(n, hip):
    """nd::sum() of up to 2^20 elements (BASELINE config 1's 1000 x 1000) runs on at most 128 fat workgroups whose last one folds
    the partials behind one ticket (np_internal.h) instead of a second launch: the same partials in the same order, so the value
    is bit-identical to the two-launch form of the same grid (np_reduce_set_variant(2000000 + 0) brings that back), call
    after call (the ticket must come back clean), and within 1e-6 of the sum the streaming grid of rounds 2-4 produces
    (np_reduce_set_variant(3000000 + 0): a different grouping of the same fp32 additions)."""
    import ctypes as C
    from numpower_amd._lib import check, load, REDUCE_OPS
    lib = load()
    x = synth.uniform((n,), 77, 0.5, 1.5)
    d = hip.DeviceArray.from_host(x)
    v = C.c_float()
    try:
        for op in ("sum", "prod", "min", "max", "mean"):
            vals = []
            for variant in (2000000 + 0, 2000000 + 256, 2000000 + 0, 2000000 + 256):
                check(lib.np_reduce_set_variant(variant))
                for _ in range(3):
                    check(lib.np_reduce_all(REDUCE_OPS[op], d.ptr, n, C.byref(v)))
                    vals.append(np.float32(v.value).view(np.uint32))
            assert len(set(int(b) for b in vals)) == 1, (op, n, vals)
            fat = v.value
            check(lib.np_reduce_set_variant(3000000 + 0))
            check(lib.np_reduce_all(REDUCE_OPS[op], d.ptr, n, C.byref(v)))
            check(lib.np_reduce_set_variant(3000000 + 128))
            if op in ("min", "max"):
                assert v.value == fat
            elif op != "prod":
                assert abs(v.value - fat) <= 1e-6 * abs(fat), (op, v.value, fat)
        flag = C.c_int(-1)
        for variant in (2000000 + 0, 2000000 + 256):
            check(lib.np_reduce_set_variant(variant))
            check(lib.np_count_mismatch(0, d.ptr, d.ptr, n, 0.0, 0.0, C.byref(flag)))
            assert flag.value == 0
    finally:
        check(lib.np_reduce_set_variant(2000000 + 256))
        check(lib.np_reduce_set_variant(3000000 + 128))
        d.free()
