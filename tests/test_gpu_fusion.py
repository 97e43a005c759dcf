"""SURVEY.md §8(f) row 4 on the GPU: fused elementwise chains

  1. against the ORACLE's composition of the same ops (oracle.binary(oracle.unary(...)): the reference's CPU
     functions applied one after the other, arithmetics.c / double_math.c) — bit for bit where every op of the chain
     is exact arithmetic (incl. the multiply / mod AVX-body quirks), within 1e-5 relative where a libm-class
     function is involved;
  2. and, second, against the same ops issued one by one through the stand-alone GPU entry points (bit-identical:
     fusion must not change a single bit),
for every op kind incl. the quirk-carrying ones, at aligned, ragged and view shapes."""
import numpy as np
import pytest

from numpower_amd import synth

pytestmark = pytest.mark.gpu


def _bits(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and ((_bits(a) == _bits(b)) | (np.isnan(a) & np.isnan(b))).all()


def _close(got, want, scale=None, rel=1e-5):
    """|got - want| <= rel * scale (scale defaults to |want|, floor 1e-30); NaN / inf patterns must agree.  `scale` is
    the magnitude of the operands of the chain's last additive step where that step can cancel."""
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    if got.shape != want.shape:
        return False
    fin = np.isfinite(want)
    if not (np.isfinite(got) == fin).all() or not (got[~fin] == want[~fin])[~np.isnan(want[~fin])].all():
        return False
    s = np.abs(want) if scale is None else np.asarray(scale, dtype=np.float64)
    s = np.broadcast_to(s, want.shape)
    return bool((np.abs(got[fin] - want[fin]) <= rel * np.maximum(s[fin], 1e-30)).all())


@pytest.mark.parametrize("shape", [(1000, 1000), (257, 1001), (3, 5), (64, 4096)])
def test_fused_equals_unfused(shape, hip, oracle):
    from numpower_amd.lazy import Lazy   # noqa: F401  (installs NDArray.lazy)
    from numpower_amd.ndarray import NDArray
    a = synth.uniform(shape, 91, -2.0, 2.0)
    b = synth.uniform(shape, 92, 0.5, 2.0)
    c = synth.uniform(shape, 93, -1.0, 1.0)
    a.reshape(-1)[::7] = 0.0          # zero products: multiply's -0.0 / +0.0 quirk must survive fusion
    ga, gb, gc = NDArray.array(a).gpu(), NDArray.array(b).gpu(), NDArray.array(c).gpu()

    O, f32 = oracle, np.float32
    fused = (ga.lazy().exp() * gb + 2.0).eval().cpu().numpy()
    want = O.binary("add", O.binary("multiply", O.unary("exp", a), b), f32(2.0))       # libm inside: 1e-5
    assert _close(fused, want), "exp(a)*b+2 vs the oracle's composition"
    plain = (NDArray.exp(ga) * gb) + 2.0
    assert _same(fused, plain.cpu().numpy())

    fused = ((ga.lazy() * gc) % gb).abs().sqrt().eval().cpu().numpy()
    want = O.unary("sqrt", O.unary("abs", O.binary("mod", O.binary("multiply", a, c), b)))   # exact ops + both quirks
    assert _same(fused, want), "sqrt(abs((a*c) % b)) vs the oracle's composition"
    plain = NDArray.sqrt(NDArray.abs((ga * gc) % gb))
    assert _same(fused, plain.cpu().numpy())

    fused = (1.5 - ga.lazy()).clip(-1.0, 2.0).round(2).eval().cpu().numpy()
    want = O.unary("round", O.unary("clip", O.binary("subtract", f32(1.5), a), -1.0, 2.0), 2.0)
    assert _same(fused, want), "round(clip(1.5-a)) vs the oracle's composition"
    plain = NDArray.round(NDArray.clip(1.5 - ga, -1.0, 2.0), 2)
    assert _same(fused, plain.cpu().numpy())

    fused = (ga.lazy().equal(gc) + ga.lazy().greater(gb).eval()).eval().cpu().numpy()
    want = O.binary("add", O.binary("equal", a, c), O.binary("greater", a, b))
    assert _same(fused, want), "equal + greater vs the oracle's composition"
    plain = NDArray.equal(ga, gc) + NDArray.greater(ga, gb)
    assert _same(fused, plain.cpu().numpy())

    # the same array bound twice is loaded once
    lz = ga.lazy() * ga + ga
    assert len(lz.inputs) == 1
    got = lz.eval().cpu().numpy()
    assert _same(got, O.binary("add", O.binary("multiply", a, a), a)), "a*a+a vs the oracle's composition"
    assert _same(got, ((ga * ga) + ga).cpu().numpy())


@pytest.mark.parametrize("shape", [(250, 400), (33, 1001), (7, 3)])
def test_fused_broadcast_operands(shape, hip, oracle):
    """Row / column / 0-d device operands inside a chain (ndarray.c:1196-1291 resolved in-kernel):
    bit-identical to the op-by-op path, incl. the AVX-body bound that depends on WHICH side the
    small operand is on (arithmetics.c:251)."""
    from numpower_amd.lazy import Lazy   # noqa: F401
    from numpower_amd.ndarray import NDArray
    R, Cc = shape
    x = synth.uniform(shape, 21, -2.0, 2.0)
    x.reshape(-1)[::5] = 0.0
    row = synth.uniform((Cc,), 22, -1.0, 1.0)
    col = synth.uniform((R, 1), 23, 0.5, 1.5)
    row1 = synth.uniform((1, Cc), 24, -1.0, 1.0)
    row[::3] = 0.0
    gx, grow, gcol, grow1 = (NDArray.array(v).gpu() for v in (x, row, col, row1))

    O = oracle
    ex = O.unary("exp", x)
    fused = (gx.lazy().exp() + grow).eval().cpu().numpy()
    assert _close(fused, O.binary("add", ex, row), np.abs(ex) + np.abs(row)[None, :]), "exp(x)+row vs the oracle"
    assert _same(fused, (NDArray.exp(gx) + grow).cpu().numpy())
    fused = (gx.lazy().exp() + gcol).eval().cpu().numpy()
    assert _close(fused, O.binary("add", ex, col), np.abs(ex) + np.abs(col)), "exp(x)+col vs the oracle"
    assert _same(fused, (NDArray.exp(gx) + gcol).cpu().numpy())

    # quirk-carrying ops with the small operand on either side: exact arithmetic, so the oracle's composition
    # (NDArray_Multiply_Float / NDArray_Mod_Float with their broadcast + AVX-body bound, arithmetics.c:251) bit for bit
    fused = ((gx.lazy() * grow) % gcol).eval().cpu().numpy()
    assert _same(fused, O.binary("mod", O.binary("multiply", x, row), col)), "(x*row) % col vs the oracle"
    assert _same(fused, ((gx * grow) % gcol).cpu().numpy())
    fused = (grow * gx.lazy()).eval().cpu().numpy()
    assert _same(fused, O.binary("multiply", row, x)), "row*x vs the oracle"
    assert _same(fused, (grow * gx).cpu().numpy())
    fused = (gcol % gx.lazy().abs() + grow1).eval().cpu().numpy()
    # (the reference's own 1 x C -> R x C broadcast is only defined for C == R, ndarray.c:1273-1291 tests the wrong
    # dimension; the oracle is given the row as a 1-D array, the form the reference does define: same meaning)
    assert _same(fused, O.binary("add", O.binary("mod", col, O.unary("abs", x)), row1.reshape(-1))), "col % |x| + row1 vs the oracle"
    assert _same(fused, ((gcol % NDArray.abs(gx)) + grow1).cpu().numpy())
    fused = gx.lazy().equal(grow1).eval().cpu().numpy()
    assert _same(fused, O.binary("equal", x, row1.reshape(-1))), "equal(x, row1) vs the oracle"
    assert _same(fused, NDArray.equal(gx, grow1).cpu().numpy())
    fused = gx.lazy().not_equal(gcol).eval().cpu().numpy()
    assert _same(fused, O.binary("not_equal", x, col)), "not_equal(x, col) vs the oracle"
    assert _same(fused, NDArray.not_equal(gx, gcol).cpu().numpy())

    # numpy meaning as an independent check
    got = (gx.lazy().exp() + grow).eval().cpu().numpy()
    want = np.exp(x.astype(np.float64)) + row[None, :]
    assert np.allclose(got, want, rtol=1e-5, atol=1e-6)


def test_long_chain_splits_and_views(hip, oracle):
    from numpower_amd.lazy import Lazy   # noqa: F401
    from numpower_amd.ndarray import NDArray
    x = synth.uniform((6, 1001), 3, 0.5, 1.5)
    g = NDArray.array(x).gpu()
    row, other = g[1], g[2]                      # 4-byte aligned views: scalar path of the kernel
    lz = row.lazy()
    plain = row
    want = x[1]
    for k in range(20):                          # longer than one chain: flushes in between
        lz = (lz * other + 0.25).sqrt()
        plain = NDArray.sqrt(plain * other + 0.25)
        want = oracle.unary("sqrt", oracle.binary("add", oracle.binary("multiply", want, x[2]), np.float32(0.25)))
    got = lz.eval().cpu().numpy()
    assert _same(got, want), "60 exact steps vs the oracle's composition"
    assert _same(got, plain.cpu().numpy())


def test_fused_errors(hip):
    from numpower_amd.lazy import Lazy   # noqa: F401
    from numpower_amd.ndarray import Error, NDArray
    a = NDArray.array(np.ones((4, 6), np.float32)).gpu()
    with pytest.raises(Error, match="Can't broadcast arrays."):
        (a.lazy() + NDArray.array(np.ones((5,), np.float32)).gpu()).eval()
    with pytest.raises(Error, match="Can't broadcast arrays."):   # the accumulator would have to grow
        (NDArray.array(np.ones((6,), np.float32)).gpu().lazy() + a).eval()
    with pytest.raises(Error, match="Device mismatch"):
        (a.lazy() + NDArray.array(np.ones((4, 6), np.float32))).eval()
    with pytest.raises(Error, match="only computes on the GPU"):
        NDArray.array(np.ones((4, 6), np.float32)).lazy().exp().eval()


@pytest.mark.parametrize("shape", [(1000, 1000), (257, 1001), (3, 5), (1,)])
def test_chain_ending_in_a_reduction(shape, hip, oracle):
    """sum / prod / min / max / mean of an expression without materialising it
    (np_fused_chain_reduce): min / max are exact; sum / mean within 1e-5 of an fp64 accumulation of
    the SAME fp32 chain values (the reference's own order is sequential fp32); deterministic."""
    from numpower_amd.lazy import Lazy   # noqa: F401
    from numpower_amd.ndarray import NDArray
    a = synth.uniform(shape, 94, -2.0, 2.0)
    b = synth.uniform(shape, 95, 0.5, 2.0)
    ga, gb = NDArray.array(a).gpu(), NDArray.array(b).gpu()
    vals = (NDArray.exp(ga) * gb + 2.0).cpu().numpy()          # the chain's fp32 values, op by op
    ovals = oracle.binary("add", oracle.binary("multiply", oracle.unary("exp", a), b), np.float32(2.0))   # ... and the oracle's
    chain = lambda: ga.lazy().exp() * gb + 2.0                 # noqa: E731
    assert chain().max() == float(vals.max()) and chain().min() == float(vals.min())
    assert abs(chain().max() - float(ovals.max())) <= 1e-5 * abs(float(ovals.max()))
    assert abs(chain().min() - float(ovals.min())) <= 1e-5 * abs(float(ovals.min()))
    s = chain().sum()
    assert s == chain().sum()
    want = float(vals.astype(np.float64).sum())
    assert abs(s - want) <= 1e-5 * abs(want)
    owant = float(ovals.astype(np.float64).sum())              # fp64 sum of the ORACLE's chain values
    assert abs(s - owant) <= 1e-5 * abs(owant)
    assert abs(chain().mean() - want / vals.size) <= 1e-5 * abs(want / vals.size)
    assert abs(chain().mean() - owant / vals.size) <= 1e-5 * abs(owant / vals.size)
    # a chain of length 0 is a plain reduction
    assert abs(ga.lazy().sum() - float(a.astype(np.float64).sum())) <= 1e-5 * np.abs(a).sum()
    # prod over a short vector; broadcast operand in a reduced chain
    if vals.size <= 15:
        p = (ga.lazy() * 1.0).prod()
        assert abs(p - float(np.prod(a.astype(np.float64)))) <= 1e-5 * abs(float(np.prod(a.astype(np.float64))))
    if len(shape) == 2:
        row = synth.uniform((shape[1],), 96, -1.0, 1.0)
        got = (ga.lazy() * NDArray.array(row).gpu()).sum()
        want = float((a * row[None, :]).astype(np.float64).sum())
        assert abs(got - want) <= 1e-5 * np.abs(a * row[None, :]).sum()


@pytest.mark.parametrize("cols", [1, 2, 3, 4, 7, 8, 12, 100, 1000, 4000, 4096, 65536, 99991])
def test_fused_broadcast_index_arithmetic(cols, hip):
    """The fused kernel derives (row, col) of an element with a multiply-high division (fast_div):
    every row length class — 1, powers of two, odd, prime, > 2^16 — against the op-by-op path."""
    from numpower_amd.lazy import Lazy   # noqa: F401
    from numpower_amd.ndarray import NDArray
    rows = max(2, min(2000, 3_000_000 // cols))
    x = synth.uniform((rows, cols), 97, -1.0, 1.0)
    r = synth.uniform((cols,), 98, -1.0, 1.0)
    c = synth.uniform((rows, 1), 99, -1.0, 1.0)
    gx, gr, gc = NDArray.array(x).gpu(), NDArray.array(r).gpu(), NDArray.array(c).gpu()
    got = (gx.lazy() + gr - gc).eval().cpu().numpy()
    assert np.array_equal(got, (x + r[None, :]) - c)


@pytest.mark.parametrize("shape", [(25000, 4000), (4096, 1000), (300, 70_000), (5000, 257), (1000, 66), (129, 64), (2000, 48), (50, 1000),
                                   (100_000, 12), (7, 5), (3, 1_000_000), (64, 128, 96), (1000,),
                                   # very many short rows: fused_chain_rows_staged_kernel (slabs through LDS), ragged last slab
                                   (900_001, 10), (530_003, 16), (180_001, 47)])
def test_chain_ending_in_an_axis_reduction(shape, hip, oracle):
    """sum / max / min / mean / prod over the last axis (any rank) and the first axis (2-d) as the chain's
    last step (np_fused_chain_reduce_axis: row-sink and column-sink kernels, or the temporary + reduce
    fallback for shapes they would idle on) against the op-by-op path: min / max bit-identical, sums within
    1e-5 of fp64 of the same chain values."""
    from numpower_amd.lazy import Lazy   # noqa: F401
    from numpower_amd.ndarray import NDArray
    a = synth.uniform(shape, 301, -1.0, 1.0)
    ga = NDArray.array(a).gpu()
    nd = len(shape)
    last = shape[-1]
    row = NDArray.array(synth.uniform((last,), 302, 0.5, 1.5)).gpu()
    col = NDArray.array(synth.uniform((shape[0], 1), 303, 0.5, 1.5)).gpu() if nd == 2 else None
    axes = [nd - 1] + ([0] if nd == 2 else []) + ([1] if nd == 3 else [])

    O, f32 = oracle, np.float32
    h_row = row.cpu().numpy()
    h_col = col.cpu().numpy() if col is not None else None

    # (label, lazy chain, op-by-op GPU chain, the ORACLE's composition on the host, every op exact?)
    def chains():
        yield "exp", lambda x: x.exp(), lambda x: NDArray.exp(x), lambda: O.unary("exp", a), False
        # (a 1-D row against an N-D array, N > 2, is left uninitialised by the reference's NDArray_Broadcast,
        # ndarray.c:1202-1223: the oracle sees the array as rows x cols, which is how the library treats it)
        yield ("x*row+0.5", lambda x: x * row + 0.5, lambda x: (x * row) + 0.5,
               lambda: O.binary("add", O.binary("multiply", a.reshape(-1, last), h_row), f32(0.5)).reshape(shape), True)
        if col is not None:
            yield ("|x-col|*2", lambda x: (x - col).abs() * 2.0, lambda x: NDArray.abs(x - col) * 2.0,
                   lambda: O.binary("multiply", O.unary("abs", O.binary("subtract", a, h_col)), f32(2.0)), True)
        yield ("tanh(x)+x", lambda x: x.tanh() + ga, lambda x: NDArray.tanh(x) + ga,      # full interpreter + input 0 reused
               lambda: O.binary("add", O.unary("tanh", a), a), False)

    for label, build, eager, by_oracle, exact in chains():
        value = eager(ga).cpu().numpy()
        v64 = value.astype(np.float64)
        ovalue = by_oracle()                      # what the reference's CPU functions give for the same chain
        o64 = ovalue.astype(np.float64)
        if exact:
            assert _same(value, ovalue), (label, shape, "op-by-op GPU chain vs the oracle's composition")
        for axis in axes:
            n_axis = shape[axis]
            for mm in ("max", "min"):
                got = getattr(build(ga.lazy()), mm)(axis=axis)
                got = got.cpu().numpy() if hasattr(got, "cpu") else np.float32(got)
                oref = getattr(ovalue, mm)(axis=axis)
                if exact:
                    assert _same(got, oref), (label, shape, axis, mm, "vs the oracle")
                else:
                    assert _close(got, oref, np.abs(oref) + 1e-3), (label, shape, axis, mm, "vs the oracle")
                assert _same(got, getattr(value, mm)(axis=axis)), (label, shape, axis, mm)
            for op, ref, scale, oref, oscale in (
                    ("sum", v64.sum(axis=axis), np.abs(v64).sum(axis=axis), o64.sum(axis=axis), np.abs(o64).sum(axis=axis)),
                    ("mean", v64.sum(axis=axis) / n_axis, np.abs(v64).sum(axis=axis) / n_axis,
                     o64.sum(axis=axis) / n_axis, np.abs(o64).sum(axis=axis) / n_axis)):
                got = getattr(build(ga.lazy()), op)(axis=axis)
                got = got.cpu().numpy() if hasattr(got, "cpu") else np.float32(got)
                assert got.shape == np.asarray(ref).shape, (label, shape, axis, op)
                assert (np.abs(got - oref) <= 1e-5 * np.maximum(oscale, 1e-30)).all(), (label, shape, axis, op, "vs the oracle")
                assert (np.abs(got - ref) <= 1e-5 * np.maximum(scale, 1e-30)).all(), (label, shape, axis, op)
                if nd == 2 and axis == 1 and 4 < last <= 48 and a.size >= (8 << 20):
                    # the staged kernel repeats reduce_rows_staged's fold order: bit-identical to the op-by-op sums
                    stepwise = getattr(NDArray, op)(eager(ga), axis)
                    assert _same(got, stepwise.cpu().numpy()), (label, shape, axis, op, "fused vs op-by-op")
    # prod: values near 1 so that long rows neither overflow nor vanish.  Factors this close to 1 are the
    # worst case for any fp32 product TREE: a pairwise fp32 product of 10^6 of them is 9e-4 off fp64 on the
    # CPU as well (the rounding of products straddling 1.0 does not average out), ~1e-9 per factor
    p = (1.0 + synth.uniform(shape, 304, -1.0, 1.0) * np.float32(1e-3)).astype(np.float32)
    gp = NDArray.array(p).gpu()
    for axis in axes:
        got = (gp.lazy() * 1.0).prod(axis=axis)
        got = got.cpu().numpy() if hasattr(got, "cpu") else np.float32(got)
        ref = p.astype(np.float64).prod(axis=axis)
        assert (np.abs(got - ref) <= (1e-5 + 2e-9 * shape[axis]) * np.abs(ref)).all(), (shape, axis, "prod")


def test_chain_axis_reduction_errors(hip):
    from numpower_amd.lazy import Lazy   # noqa: F401
    from numpower_amd.ndarray import Error, NDArray
    g = NDArray.array(synth.uniform((10, 20), 305, 0.0, 1.0)).gpu()
    with pytest.raises(Error, match="axis 2 is out of bounds for array of dimension 2"):
        g.lazy().exp().sum(axis=2)
    with pytest.raises(Error, match="axis -1 is out of bounds for array of dimension 2"):
        g.lazy().exp().sum(axis=-1)


def test_chains_are_persistent_values(hip):
    """A chain continued in two directions gives two different expressions (`_unary` / `_binary`
    used to append to the one shared object, so `y1 = base + 1; y2 = base * 2` both became (exp(a)+1)*2)."""
    from numpower_amd.lazy import Lazy   # noqa: F401
    from numpower_amd.ndarray import NDArray
    a = synth.uniform((513, 129), 97, -2.0, 2.0)
    ga = NDArray.array(a).gpu()
    base = ga.lazy().exp()
    y1 = base + 1.0
    y2 = base * 2.0
    y3 = (base - 0.5).abs()
    assert len(base.ops) == 1 and len(y1.ops) == 2 and len(y2.ops) == 2 and len(y3.ops) == 3
    e = NDArray.exp(ga)
    assert _same(base.eval().cpu().numpy(), e.cpu().numpy())
    assert _same(y1.eval().cpu().numpy(), (e + 1.0).cpu().numpy())
    assert _same(y2.eval().cpu().numpy(), (e * 2.0).cpu().numpy())
    assert _same(y3.eval().cpu().numpy(), NDArray.abs(e - 0.5).cpu().numpy())
    assert _same(base.eval().cpu().numpy(), e.cpu().numpy())          # still exp(a) after its continuations ran
    # a full chain (12 ops; the one operand array is bound once) continued twice: each continuation evaluates
    # the full prefix and starts afresh, the 12-op chain itself stays what it was
    b = synth.uniform((513, 129), 98, 0.0, 0.25)
    gb = NDArray.array(b).gpu()
    long = ga.lazy()
    for _ in range(12):
        long = long + gb
    assert len(long.ops) == 12 and len(long.inputs) == 2
    p, q = long * 3.0, long - 1.0
    ref = a.copy()
    for _ in range(12):
        ref = ref + b
    assert _same(p.eval().cpu().numpy(), ref * np.float32(3.0)) and _same(q.eval().cpu().numpy(), ref - np.float32(1.0))
    assert len(long.ops) == 12 and len(p.ops) == 1 and len(q.ops) == 1


@pytest.mark.parametrize("shape", [(1000, 1000), (257, 1001), (3, 5), (5000, 4000)])
def test_pow_steps_are_the_stand_alone_pow_bit_for_bit(shape, hip, oracle):
    """`value ** 2` with a PHP number is x * x in np_binary (the correctly rounded square); a chain runs the same product for that
    step — and the general pow (array exponent, another number, the number as the BASE, a 0-d device 2.0) runs np_binary's pow.
    Bit-identical to the op-by-op sequence at every size class of the stand-alone kernel (its log2 table lives in LDS for small
    arrays, in registers for large ones), incl. as the operand of a full and of an axis reduction."""
    from numpower_amd.lazy import Lazy   # noqa: F401
    from numpower_amd.ndarray import NDArray
    x = synth.uniform(shape, 191, -3.0, 3.0)
    y = synth.uniform(shape, 192, 0.5, 4.0)
    p = synth.uniform(shape, 193, 0.25, 4.0)
    x.reshape(-1)[::11] = y.reshape(-1)[::11]                      # zero differences
    gx, gy, gp = NDArray.array(x).gpu(), NDArray.array(y).gpu(), NDArray.array(p).gpu()
    two = NDArray.array(np.float32(2.0)).gpu()

    d = oracle.binary("subtract", x, y)
    fused = ((gx.lazy() - gy) ** 2).eval().cpu().numpy()
    assert _same(fused, d * d), "(x - y) ** 2 is the fp32 product"
    assert _same(fused, ((gx - gy) ** 2).cpu().numpy())
    fused = (((gx.lazy() - gy) ** 2) * 0.5 + gp).eval().cpu().numpy()
    assert _same(fused, (((gx - gy) ** 2) * 0.5 + gp).cpu().numpy())
    fused = ((gx.lazy() ** 2) ** 2).eval().cpu().numpy()            # first step of a chain, and twice
    assert _same(fused, ((gx ** 2) ** 2).cpu().numpy())

    for name, lz, eager in (("p ** y", gp.lazy().sqrt() ** gy, NDArray.sqrt(gp) ** gy),
                            ("p ** 1.5", (gp.lazy() * 2.0) ** 1.5, (gp * 2.0) ** 1.5),
                            ("2 ** x", 2.0 ** (gx.lazy() + 1.0), NDArray._binary("pow", 2.0, gx + 1.0)),
                            ("p ** 0-d 2.0", (gp.lazy() + 1.0) ** two, (gp + 1.0) ** two)):
        assert _same(lz.eval().cpu().numpy(), eager.cpu().numpy()), name
    want = oracle.binary("pow", oracle.unary("sqrt", p), y)
    assert _close((gp.lazy().sqrt() ** gy).eval().cpu().numpy(), want), "sqrt(p) ** y vs the oracle's composition"

    # a mean squared error in one launch: the squares never go to memory
    d64 = (x.astype(np.float64) - y.astype(np.float64)) ** 2
    mse = ((gx.lazy() - gy) ** 2).mean()
    assert abs(mse - d64.mean()) <= 1e-5 * d64.mean()
    for axis in (0, 1):
        got = ((gx.lazy() - gy) ** 2).sum(axis=axis).cpu().numpy().astype(np.float64)
        assert (np.abs(got - d64.sum(axis=axis)) <= 1e-5 * d64.sum(axis=axis)).all(), axis


@pytest.mark.parametrize("shape", [(2000, 3000), (257, 1001), (3, 5)])
def test_chains_that_name_an_array_twice(shape, hip, oracle):
    """f(x) * y + x and (x - y) * y name a full array twice: such chains run on the interpreter's KEEP variant (plain loads, so the
    second read of a line is an L2 hit instead of a second stream from HBM) — the arithmetic is the same code: bit-identical to the
    op-by-op sequence and to the same chain on the common kernel (np_elementwise_set_variant(7002))."""
    from numpower_amd._lib import check, load
    from numpower_amd.lazy import Lazy   # noqa: F401
    from numpower_amd.ndarray import NDArray
    lib = load()
    x = synth.uniform(shape, 291, -2.0, 2.0)
    y = synth.uniform(shape, 292, 0.5, 2.0)
    x.reshape(-1)[::9] = 0.0
    gx, gy = NDArray.array(x).gpu(), NDArray.array(y).gpu()
    cases = {"tanh(x)*y+x": (lambda: (gx.lazy().tanh() * gy + gx), lambda: NDArray.tanh(gx) * gy + gx),
             "(x-y)*y": (lambda: (gx.lazy() - gy) * gy, lambda: (gx - gy) * gy),
             "x*y+x-y": (lambda: gx.lazy() * gy + gx - gy, lambda: gx * gy + gx - gy),
             "x*x*x": (lambda: gx.lazy() * gx * gx, lambda: gx * gx * gx)}
    for name, (lz, eager) in cases.items():
        want = eager().cpu().numpy()
        got = lz().eval().cpu().numpy()
        assert _same(got, want), name
        check(lib.np_elementwise_set_variant(7002))
        try:
            assert _same(lz().eval().cpu().numpy(), want), name + " (common kernel)"
        finally:
            check(lib.np_elementwise_set_variant(0))
    want = oracle.binary("multiply", oracle.binary("subtract", x, y), y)      # exact ops, incl. multiply's zero-sign quirk
    assert _same(((gx.lazy() - gy) * gy).eval().cpu().numpy(), want)
    s64 = ((x.astype(np.float64) - y) * y).sum()
    assert abs(((gx.lazy() - gy) * gy).sum() - s64) <= 1e-5 * np.abs((x.astype(np.float64) - y) * y).sum()
