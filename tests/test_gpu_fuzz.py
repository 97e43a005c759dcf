"""Seeded random-shape parity sweep: the specialised kernels (thin / split-K / padded GEMMs, lane-group
and chunked reductions, tiled / skinny / gathered permutes, ragged broadcasts, pitched copies, radix
select) are picked by shape, so shapes are drawn from a pool rich in the dispatch boundaries (1, 3, 4, 5,
16, 17, 31..33, 63..65, 255..257, 1000, 2048, 4099, 65536, ...).  NP_FUZZ_CASES scales the sweep
(tools/fuzz_parity.py runs it long)."""
import math
import os

import numpy as np
import pytest

from numpower_amd import synth

pytestmark = pytest.mark.gpu

POOL = [1, 2, 3, 4, 5, 7, 8, 15, 16, 17, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 255, 256, 257, 511, 1000, 1024, 2047, 2048,
        2049, 4099, 10_000, 65_535, 65_536, 100_003, 1_000_000]
CASES = int(os.environ.get("NP_FUZZ_CASES", "60"))
SEED = int(os.environ.get("NP_FUZZ_SEED", "0"))     # added to every family's generator seed and data seeds


def _nd():
    from numpower_amd.ndarray import NDArray
    return NDArray


def _shape(rng, ndim, budget):
    """ndim extents from POOL whose product stays within budget (drawn one by one from what still fits)."""
    dims, left = [], budget
    for _ in range(ndim):
        fits = [p for p in POOL if p <= left]
        d = int(rng.choice(fits))
        dims.append(d)
        left //= d
    order = rng.permutation(ndim)
    return tuple(dims[i] for i in order)


def _bits(x):
    return np.ascontiguousarray(x, np.float32).view(np.uint32)


def test_fuzz_axis_reductions(hip, oracle):
    nd = _nd()
    rng = np.random.default_rng(1234 + SEED)
    for case in range(CASES):
        ndim = int(rng.integers(1, 5))
        shape = _shape(rng, ndim, 3_000_000)
        axis = int(rng.integers(0, ndim))
        x = synth.uniform(shape, 1000 + case + 100_000 * SEED, -1.0, 1.0)
        g = nd.array(x).gpu()
        n_axis = shape[axis]
        for op in ("sum", "max", "min", "mean"):
            got = getattr(nd, op)(g, axis)
            got = got.cpu().numpy() if hasattr(got, "cpu") else np.float32(got)
            if op in ("max", "min"):
                want = getattr(x, op)(axis=axis)
                assert (_bits(got) == _bits(want)).all(), (op, shape, axis)
            else:
                ref = x.astype(np.float64).sum(axis=axis) / (n_axis if op == "mean" else 1)
                scale = np.abs(x).astype(np.float64).sum(axis=axis) / (n_axis if op == "mean" else 1)
                assert (np.abs(got - ref) <= 1e-5 * np.maximum(scale, 1e-30)).all(), (op, shape, axis)
        for name, fn in (("argmax", np.argmax), ("argmin", np.argmin)):
            got = getattr(nd, name)(g, axis)
            got = got.cpu().numpy() if hasattr(got, "cpu") else np.float32(got)
            assert (np.asarray(got, np.float32) == fn(x, axis=axis).astype(np.float32)).all(), (name, shape, axis)


def test_fuzz_broadcast_binary(hip, oracle):
    nd = _nd()
    rng = np.random.default_rng(99 + SEED)
    ops = ["add", "subtract", "multiply", "divide", "greater", "maximum", "mod", "equal"]
    for case in range(CASES):
        rows, cols = _shape(rng, 2, 4_000_000)
        kind = rng.choice(["full", "row", "col", "scalar"])
        a = synth.uniform((rows, cols), 2000 + case + 100_000 * SEED, -2.0, 2.0)
        a.reshape(-1)[::7] = 0.0
        b = {"full": lambda: synth.uniform((rows, cols), 3000 + case + 100_000 * SEED, 0.5, 2.0),
             "row": lambda: synth.uniform((cols,), 3000 + case + 100_000 * SEED, 0.5, 2.0),
             "col": lambda: synth.uniform((rows, 1), 3000 + case + 100_000 * SEED, 0.5, 2.0),
             "scalar": lambda: np.float32(1.5)}[kind]()
        op = ops[case % len(ops)]
        ga, gb = nd.array(a).gpu(), (nd.array(b).gpu() if kind != "scalar" else float(b))
        for left, right, hl, hr in ((ga, gb, a, b), (gb, ga, b, a)):
            if op in ("maximum", "minimum") and kind == "scalar":
                continue
            try:
                want = oracle.binary(op, hl, hr)
            except Exception:
                continue      # a combination the reference rejects (e.g. (R,1) on the left)
            got = nd._binary(op, left, right).cpu().numpy()
            # x % 0 is NaN on both sides, but x86 and gfx950 produce different default-NaN sign bits
            same = (_bits(got) == _bits(want)) | (np.isnan(got) & np.isnan(want))
            assert same.all(), (op, kind, rows, cols)


def test_fuzz_matmul(hip):
    nd = _nd()
    rng = np.random.default_rng(7 + SEED)
    pool = [p for p in POOL if p <= 100_003]
    done = 0
    while done < CASES:
        m, n, k = (int(rng.choice(pool)) for _ in range(3))
        if m * k > 40_000_000 or k * n > 40_000_000 or m * n > 40_000_000 or m * n * k > 60_000_000_000:
            continue
        done += 1
        a = synth.uniform((m, k), 4000 + done + 100_000 * SEED, -1.0, 1.0)
        b = synth.uniform((k, n), 5000 + done + 100_000 * SEED, -1.0, 1.0)
        got = nd.matmul(nd.array(a).gpu(), nd.array(b).gpu()).cpu().numpy()
        # spot-check up to 64 rows x 64 columns against fp64 (the full product would dominate the run time)
        ri = np.unique(rng.integers(0, m, size=min(m, 64)))
        ci = np.unique(rng.integers(0, n, size=min(n, 64)))
        ref = a[ri].astype(np.float64) @ b[:, ci].astype(np.float64)
        scale = np.abs(a[ri]).astype(np.float64) @ np.abs(b[:, ci]).astype(np.float64)
        assert got.shape == (m, n)
        assert (np.abs(got[np.ix_(ri, ci)] - ref) <= 2e-6 * np.maximum(scale, 1e-30)).all(), (m, n, k)
        if m * n <= 4_000_000:      # and nothing written outside / left unwritten
            assert np.isfinite(got).all()


def test_fuzz_permute_and_concatenate(hip):
    nd = _nd()
    rng = np.random.default_rng(5 + SEED)
    for case in range(CASES):
        ndim = int(rng.integers(2, 6))
        shape = _shape(rng, ndim, 3_000_000)
        perm = [int(p) for p in rng.permutation(ndim)]
        x = synth.uniform(shape, 6000 + case + 100_000 * SEED, -1.0, 1.0)
        g = nd.array(x).gpu()
        got = nd.transpose(g, perm).cpu().numpy()
        assert (_bits(got) == _bits(np.transpose(x, perm))).all(), (shape, perm)
        axis = int(rng.integers(0, ndim))
        other = list(shape)
        other[axis] = int(rng.choice([1, 2, 3, 17, 64]))
        if math.prod(other) > 3_000_000:
            continue
        y = synth.uniform(tuple(other), 7000 + case + 100_000 * SEED, -1.0, 1.0)
        got = nd.concatenate([g, nd.array(y).gpu(), g], axis).cpu().numpy()
        assert (_bits(got) == _bits(np.concatenate([x, y, x], axis))).all(), (shape, axis)


def test_fuzz_argreduce(hip, oracle):
    """np_argreduce over random (outer, axis, inner) views — every streaming form of round 5 is picked by shape (one wave per row,
    grid-stride rows, flat float4 walks for a few columns / whole float4 column groups, column tiles with the coalesced fold, the
    lane-group and generic kernels) — with ties, +-inf and NaNs thrown in, position 0 included: exact indices against the oracle."""
    import ctypes as C
    from numpower_amd._lib import check, load
    lib = load()
    rng = np.random.default_rng(61 + SEED)
    inners = [1, 1, 1, 2, 3, 4, 5, 8, 12, 16, 63, 64, 100, 128, 191, 192, 193, 256, 257, 1000, 1024, 4099]
    for case in range(CASES):
        inner = int(rng.choice(inners))
        outer, length = _shape(rng, 2, max(2, 4_000_000 // inner))
        if rng.integers(0, 3) == 0:
            outer, length = 1, outer * length              # one long axis
        length = max(length, 1)
        shape = (outer, length, inner)
        x = synth.uniform(shape, 15000 + case + 100_000 * SEED, -1.0, 1.0)
        flat = x.reshape(-1)
        if rng.integers(0, 2):
            flat[::int(rng.choice([3, 7, 64]))] = np.float32(rng.choice([0.75, -0.75, 0.0]))      # runs of exact ties
        for v, step in ((np.inf, 1013), (-np.inf, 1511), (np.nan, 4099)):
            if rng.integers(0, 2):
                flat[int(rng.integers(0, flat.size))::step] = v
        if rng.integers(0, 2):
            x[:, 0, :].reshape(-1)[::int(rng.choice([1, 2, 5]))] = np.nan                         # NaNs in position 0 of the axis
        d = hip.DeviceArray.from_host(x)
        out = hip.DeviceArray((outer * inner,))
        for is_max in (True, False):
            check(lib.np_argreduce(1 if is_max else 0, d.ptr, outer, length, inner, out.ptr))
            got = out.to_host()
            want = np.asarray(oracle.argreduce(x, 1, is_max), np.float32).reshape(-1)
            bad = np.flatnonzero(got != want)
            assert bad.size == 0, (shape, is_max, case, bad[:5].tolist(), got[bad[:5]].tolist(), want[bad[:5]].tolist())
        d.free()
        out.free()
    del C


def test_fuzz_order_statistics(hip, oracle):
    nd = _nd()
    rng = np.random.default_rng(11 + SEED)
    for case in range(CASES):
        n = int(rng.choice([p for p in POOL if p >= 2]))
        style = case % 3
        x = synth.uniform((n,), 8000 + case + 100_000 * SEED, -1.0, 1.0)
        if style == 1:
            x = np.rint(x * 3).astype(np.float32)            # heavy duplicates
        elif style == 2:
            x = (x * np.float32(1e-3) + np.float32(1.0)).astype(np.float32)   # one binade: ranks part in the low digits
        x[x == 0] = 0.0                                      # the reference's comparator cannot order -0 / +0
        g = nd.array(x).gpu()
        assert np.float32(nd.median(g)).view(np.uint32) == oracle.median(x).view(np.uint32), (n, style)
        q = float(rng.uniform(0.0, 1.0))
        assert np.float32(nd.quantile(g, q)).view(np.uint32) == oracle.quantile(x, q).view(np.uint32), (n, style, q)


def test_fuzz_order_statistics_bracket_path(hip):
    """np_order_stat with the bracket path forced on every array >= 2048 elements: random length, rank and
    distribution (incl. ones whose bracket is refused or missed), against a sort of the keys."""
    import ctypes as C
    from numpower_amd import _lib
    from tests.test_gpu_order_stats import _from_keys, _keys
    lib = _lib.load()
    rng = np.random.default_rng(23 + SEED)
    _lib.check(lib.np_select_set_variant(2048))
    try:
        taken = 0
        for case in range(CASES):
            n = int(rng.integers(2048, 3_000_000))
            u = synth.uniform((n,), 9000 + case + 100_000 * SEED, 0.0, 1.0)
            style = case % 6
            if style == 0:
                x = u
            elif style == 1:      # many binades, both signs
                x = ((u - np.float32(0.5)) * np.exp(synth.uniform((n,), 9500 + case, -30.0, 30.0))).astype(np.float32)
            elif style == 2:      # a big lump of one value among random ones
                x = np.where(u < rng.uniform(0.05, 0.6), np.float32(rng.uniform(-1, 1)), u - np.float32(0.5)).astype(np.float32)
            elif style == 3:      # sorted / reversed
                x = np.sort(u)[::(1 if case % 12 < 6 else -1)].copy()
            elif style == 4:      # few distinct values
                x = np.rint(u * np.float32(rng.integers(2, 200))).astype(np.float32)
            else:                 # one binade, dense keys
                x = (u * np.float32(1e-3) + np.float32(rng.uniform(0.5, 4.0))).astype(np.float32)
            x = np.ascontiguousarray(x, np.float32)
            want = np.sort(_keys(x))
            buf = _lib.DeviceBuffer(4 * n)
            _lib.check(lib.np_memcpy_h2d(buf.ptr, x.ctypes.data, 4 * n))
            out = (C.c_float * 2)()
            ranks = {0, n - 1, int(rng.integers(0, n)), int(rng.integers(0, n)), int(rng.integers(0, min(n, 5000))),
                     n - 1 - int(rng.integers(0, min(n, 5000)))}
            for k in sorted(ranks):
                _lib.check(lib.np_order_stat(buf.ptr, n, k, out))
                exp = _from_keys([want[k], want[min(k + 1, n - 1)]]).view(np.uint32).tolist()
                assert np.float32([out[0], out[1]]).view(np.uint32).tolist() == exp, (case, style, n, k)
                path = C.c_int(-1)
                _lib.check(lib.np_select_last_path(C.byref(path)))
                taken += path.value
            buf.free()
        assert taken > 0          # the sweep did exercise the copied-keys passes, not only the fallback
    finally:
        _lib.check(lib.np_select_set_variant(1))


# ops whose GPU result is the oracle's bit for bit (tests/test_gpu_parity.py); everything else is libm-class: 1e-5
_EXACT_CHAIN_UNARY = ["abs", "sqrt", "negate", "floor", "ceil", "sign", "rint", "trunc", "reciprocal"]
_EXACT_CHAIN_BINARY = ["add", "subtract", "multiply", "divide", "maximum", "minimum", "greater", "less_equal", "mod", "equal"]


def _step_agrees(name, got, want):
    """One op applied to IDENTICAL inputs by the GPU (got) and by the oracle (want)."""
    both_nan = np.isnan(got) & np.isnan(want)
    if name in _EXACT_CHAIN_UNARY or name in _EXACT_CHAIN_BINARY:
        return got.shape == want.shape and ((_bits(got) == _bits(want)) | both_nan).all()
    g, w = got.astype(np.float64), want.astype(np.float64)
    fin = np.isfinite(w)
    if got.shape != want.shape or not (np.isfinite(g) == fin).all() or not ((g == w) | both_nan)[~fin].all():
        return False
    return bool((np.abs(g[fin] - w[fin]) <= 1e-5 * np.abs(w[fin]) + 1e-11).all())


def test_fuzz_fused_chains(hip, oracle):
    """Random linear chains (<= 10 steps, row / column / 0-d / python-scalar operands on either side) through the
    one-kernel interpreter, checked three ways:
      * against the same ops issued one by one on the GPU: bit-identical (fusion changes nothing);
      * every step of that op-by-op run against the ORACLE applied to the very same step input ("teacher forced"):
        bit-exact for exact ops, 1e-5 for libm-class ones.  Random chains put discontinuous ops (floor, sign,
        greater, mod ...) and unbounded amplifiers (divide, reciprocal, sinh) behind libm-class ones, so the
        end-to-end composition of two 1-ulp-different libms is not comparable with any fixed tolerance; step by step
        on identical inputs it is, and together with the first check it pins the fused result to the oracle;
      * chains drawn from exact ops only, end to end against the oracle's composition
        (oracle.binary(oracle.unary(...))): bit-exact."""
    from numpower_amd.lazy import Lazy   # noqa: F401  (installs NDArray.lazy)
    nd = _nd()
    rng = np.random.default_rng(21 + SEED)
    unary_all = ["abs", "exp", "sqrt", "sin", "cos", "tanh", "negate", "floor", "ceil", "sign", "log1p", "arctan", "rint",
                 "trunc", "sinh", "reciprocal", "log", "expm1"]
    binary_all = ["add", "subtract", "multiply", "divide", "maximum", "minimum", "greater", "less_equal", "mod", "pow", "equal"]
    n_cases = max(CASES // 2, 10)
    for case in range(2 * n_cases):
        exact_only = case >= n_cases            # second half: exact ops only, end to end against the oracle
        unary = _EXACT_CHAIN_UNARY if exact_only else unary_all
        binary = _EXACT_CHAIN_BINARY if exact_only else binary_all
        rows, cols = _shape(rng, 2, 2_000_000)
        # 1 x C and R x 1 arrays are left out: there a vector operand has as many elements as the chain, the
        # reference treats the pair as a flat elementwise op and the eager result takes the LEFT operand's
        # shape ((C,) instead of (1, C)), which changes what later steps may broadcast with; a chain
        # keeps the shape of its first array throughout
        rows, cols = max(rows, 2), max(cols, 2)
        a = synth.uniform((rows, cols), 9000 + case + 100_000 * SEED, -1.5, 1.5)
        a.reshape(-1)[::5] = 0.0
        ga = nd.array(a).gpu()
        host = {"full": synth.uniform((rows, cols), 9500 + case + 100_000 * SEED, 0.25, 2.0),
                "row": synth.uniform((cols,), 9600 + case + 100_000 * SEED, 0.25, 2.0),
                "col": synth.uniform((rows, 1), 9700 + case + 100_000 * SEED, 0.25, 2.0),
                "zero_d": np.float32(1.25), "py": np.float32(0.75)}
        operands = {k: nd.array(v).gpu() for k, v in host.items() if k != "py"}
        operands["py"] = 0.75
        lz, eager, desc = ga.lazy(), ga, []
        o_end = a                                # the oracle's own composition from the start (exact chains)
        for _ in range(int(rng.integers(1, 11))):
            step_in = eager.cpu().numpy()        # what this step's GPU op reads
            if rng.random() < 0.45:
                name = str(rng.choice(unary))
                lz = getattr(lz, name)()
                eager = nd._unary(name, eager) if hasattr(nd, "_unary") else getattr(nd, name)(eager)
                desc.append(name)
                o_step = oracle.unary(name, step_in)
                o_end = oracle.unary(name, o_end) if exact_only else None
            else:
                name, kind, swap = str(rng.choice(binary)), str(rng.choice(list(operands))), bool(rng.random() < 0.3)
                if kind == "col" and swap:
                    swap = False          # (R,1) on the left of an (R,C) array: rejected by the reference's broadcast
                if kind == "py" and name in ("maximum", "minimum", "greater", "less_equal", "equal"):
                    kind = "zero_d"
                other = operands[kind]
                if swap:
                    lz = lz._binary(name, other, True)
                    eager = nd._binary(name, other, eager)
                    o_step = oracle.binary(name, host[kind], step_in)
                    o_end = oracle.binary(name, host[kind], o_end) if exact_only else None
                else:
                    lz = lz._binary(name, other, False)
                    eager = nd._binary(name, eager, other)
                    o_step = oracle.binary(name, step_in, host[kind])
                    o_end = oracle.binary(name, o_end, host[kind]) if exact_only else None
                desc.append("%s(%s%s)" % (name, kind, ",swapped" if swap else ""))
            assert _step_agrees(name, eager.cpu().numpy(), o_step.reshape(step_in.shape)), \
                ("step vs the oracle on the same input", (rows, cols), desc)
        got, want = lz.eval().cpu().numpy(), eager.cpu().numpy()
        same = (_bits(got) == _bits(want)) | (np.isnan(got) & np.isnan(want))
        assert got.shape == want.shape and same.all(), ((rows, cols), desc)
        if exact_only:
            o_end = o_end.reshape(got.shape)
            same = (_bits(got) == _bits(o_end)) | (np.isnan(got) & np.isnan(o_end))
            assert same.all(), ("fused chain vs the oracle's composition", (rows, cols), desc, int((~same).sum()))


def test_fuzz_vectors_statistics_slices(hip, oracle):
    """dot (matrix . vector, vector . vector), outer, variance / std, array_equal / allclose and strided
    slices at random shapes."""
    nd = _nd()
    rng = np.random.default_rng(31 + SEED)
    for case in range(CASES):
        m, n = _shape(rng, 2, 6_000_000)
        a = synth.uniform((m, n), 11000 + case + 100_000 * SEED, -1.0, 1.0)
        x = synth.uniform((n,), 12000 + case + 100_000 * SEED, -1.0, 1.0)
        ga, gx = nd.array(a).gpu(), nd.array(x).gpu()
        got = nd.dot(ga, gx)
        got = got.cpu().numpy() if hasattr(got, "cpu") else np.float32(got)
        ref = a.astype(np.float64) @ x.astype(np.float64)
        scale = np.abs(a).astype(np.float64) @ np.abs(x).astype(np.float64)
        assert (np.abs(got - ref) <= 2e-6 * np.maximum(scale, 1e-30)).all(), ("matvec", m, n)
        got = np.float32(nd.dot(gx, gx))
        ref = float((x.astype(np.float64) ** 2).sum())
        assert abs(got - ref) <= 2e-6 * max(ref, 1e-30), ("inner", n)
        if m * n <= 2_000_000:
            y = synth.uniform((m,), 13000 + case + 100_000 * SEED, -1.0, 1.0)
            got = nd.outer(nd.array(y).gpu(), gx).cpu().numpy()
            want = oracle.outer(y, x)     # sger onto a zeroed matrix: a zero product is +0.0 whatever its factors' signs
            bad = np.argwhere(_bits(got) != _bits(want))
            assert len(bad) == 0, ("outer", m, n, case, len(bad), bad[:4].tolist(),
                                   [(float(got[tuple(i)]), float(want[tuple(i)])) for i in bad[:4]])
        flat = a.reshape(-1)
        var = float(np.float32(nd.variance(ga)))
        ref = float(flat.astype(np.float64).var())
        assert abs(var - ref) <= 1e-5 * max(ref, 1e-30), ("variance", m, n)
        std = float(np.float32(nd.std(ga)))
        assert abs(std - ref ** 0.5) <= 1e-5 * max(ref ** 0.5, 1e-30), ("std", m, n)
        b = a.copy()
        assert nd.array_equal(ga, nd.array(b).gpu()) is True and nd.allclose(ga, nd.array(b).gpu()) is True
        i = int(rng.integers(0, m)); j = int(rng.integers(0, n))
        b[i, j] += np.float32(0.5)
        gb = nd.array(b).gpu()
        assert nd.array_equal(ga, gb) is False and nd.allclose(ga, gb) is False, (m, n, i, j)
        # strided slice on both axes: [start, stop, step]
        r0 = int(rng.integers(0, m)); r1 = int(rng.integers(r0, m)) + 1; rs = int(rng.integers(1, 4))
        c0 = int(rng.integers(0, n)); c1 = int(rng.integers(c0, n)) + 1; cs = int(rng.integers(1, 4))
        got = ga.slice([r0, r1, rs], [c0, c1, cs]).cpu().numpy()
        assert (_bits(got) == _bits(a[r0:r1:rs, c0:c1:cs])).all(), ("slice", (m, n), (r0, r1, rs), (c0, c1, cs))


def test_fuzz_chain_axis_ends(hip, oracle):
    """Chains ending in an axis reduction at random 2-d / 3-d shapes (every row-length regime of the sink
    kernels and the fallback) vs the ORACLE's composition of the chain reduced in fp64 (max / min: bit-exact for the
    exact chain, 1e-5 for the libm ones), and vs the materialised GPU chain reduced by numpy."""
    from numpower_amd.lazy import Lazy   # noqa: F401
    nd = _nd()
    rng = np.random.default_rng(41 + SEED)
    for case in range(CASES):
        ndim = int(rng.integers(1, 4))
        shape = _shape(rng, ndim, 3_000_000)
        a = synth.uniform(shape, 14000 + case + 100_000 * SEED, -1.0, 1.0)
        ga = nd.array(a).gpu()
        row = nd.array(synth.uniform((shape[-1],), 15000 + case + 100_000 * SEED, 0.5, 1.5)).gpu()
        kind = case % 3
        h_row = row.cpu().numpy()
        if kind == 0:
            lz, value, ov = ga.lazy().exp(), nd.exp(ga), oracle.unary("exp", a)
        elif kind == 1:
            try:
                ov = oracle.unary("abs", oracle.binary("multiply", a, h_row))
            except oracle.OracleError:
                # 1-D -> N-D with N > 2: the reference's NDArray_Broadcast leaves the temporary uninitialised
                # (ndarray.c:1202-1223 handles N = 2 only); the library gives it its NumPy meaning — |a * row| in
                # IEEE fp32, where abs() removes the one thing the reference's multiply quirk touches (the zero's sign)
                ov = np.abs(a * h_row)
            lz, value = (ga.lazy() * row).abs(), nd.abs(ga * row)
        else:
            lz, value, ov = ga.lazy().sin() * ga, nd.sin(ga) * ga, oracle.binary("multiply", oracle.unary("sin", a), a)
        v = value.cpu().numpy()
        ov = ov.reshape(v.shape)
        axis = int(rng.integers(0, ndim))
        op = ("sum", "max", "min", "mean")[int(rng.integers(0, 4))]
        got = getattr(lz, op)(axis=axis)
        got = got.cpu().numpy() if hasattr(got, "cpu") else np.float32(got)
        if op in ("max", "min"):
            oref = getattr(ov, op)(axis=axis)
            if kind == 1:
                assert (_bits(got) == _bits(oref)).all(), (shape, axis, op, kind, "vs the oracle")
            else:
                assert (np.abs(got.astype(np.float64) - oref) <= 1e-5 * np.abs(oref) + 1e-9).all(), (shape, axis, op, kind, "vs the oracle")
            assert (_bits(got) == _bits(getattr(v, op)(axis=axis))).all(), (shape, axis, op, kind)
        else:
            div = shape[axis] if op == "mean" else 1
            oref = ov.astype(np.float64).sum(axis=axis) / div
            oscale = np.abs(ov).astype(np.float64).sum(axis=axis) / div
            assert got.shape == oref.shape and (np.abs(got - oref) <= 1e-5 * np.maximum(oscale, 1e-30)).all(), (shape, axis, op, kind, "vs the oracle")
            ref = v.astype(np.float64).sum(axis=axis) / div
            scale = np.abs(v).astype(np.float64).sum(axis=axis) / div
            assert got.shape == ref.shape and (np.abs(got - ref) <= 1e-5 * np.maximum(scale, 1e-30)).all(), (shape, axis, op, kind)


def test_fuzz_views_at_odd_offsets(hip, oracle):
    """Row views of matrices with odd row lengths start at any 4-byte offset: every elementwise / reduction
    kernel behind them must take dword-aligned pointers and ragged lengths (float4 bodies, scalar tails)."""
    nd = _nd()
    rng = np.random.default_rng(51 + SEED)
    exact = ["abs", "sqrt", "floor", "ceil", "rint", "negate", "sign", "reciprocal", "trunc"]
    loose = ["exp", "log", "sin", "tanh", "log1p", "arctan", "sinh"]
    for case in range(CASES):
        cols = int(rng.choice([1, 3, 5, 7, 9, 13, 17, 31, 33, 63, 65, 127, 129, 255, 257, 1001, 4099, 65_537, 1_000_003]))
        rows = int(rng.integers(2, 6))
        m = synth.uniform((rows, cols), 16000 + case + 100_000 * SEED, 0.25, 3.0)
        w = synth.uniform((rows, cols), 17000 + case + 100_000 * SEED, 0.5, 2.0)
        gm, gw = nd.array(m).gpu(), nd.array(w).gpu()
        i, j = int(rng.integers(0, rows)), int(rng.integers(0, rows))
        va, vb = gm.slice([i]), gw.slice([j])            # contiguous views at offsets i*cols, j*cols floats
        ha, hb = m[i], w[j]
        op = str(rng.choice(exact))
        assert (_bits(nd._unary(op, va).cpu().numpy()) == _bits(oracle.unary(op, ha))).all(), (op, cols, i)
        op = str(rng.choice(loose))
        got, want = nd._unary(op, va).cpu().numpy().astype(np.float64), oracle.unary(op, ha).astype(np.float64)
        assert (np.abs(got - want) <= 1e-5 * np.abs(want) + 1e-11).all(), (op, cols, i)
        for bop in ("add", "multiply", "divide", "greater", "maximum"):
            got = nd._binary(bop, va, vb).cpu().numpy()
            assert (_bits(got) == _bits(oracle.binary(bop, ha, hb))).all(), (bop, cols, i, j)
        assert (_bits(nd._binary("subtract", va, 1.5).cpu().numpy()) == _bits(oracle.binary("subtract", ha, np.float32(1.5)))).all()
        s = float(nd.sum(va)); ref = float(ha.astype(np.float64).sum())
        assert abs(s - ref) <= 1e-5 * ref, ("sum", cols, i)
        assert np.float32(nd.max(va)) == ha.max() and np.float32(nd.min(vb)) == hb.min(), ("minmax", cols, i, j)
        assert np.float32(nd.median(va)).view(np.uint32) == oracle.median(ha).view(np.uint32), ("median", cols, i)
        d = float(np.float32(nd.dot(va, vb))); ref = float(ha.astype(np.float64) @ hb.astype(np.float64))
        assert abs(d - ref) <= 2e-6 * float(np.abs(ha).astype(np.float64) @ np.abs(hb).astype(np.float64)), ("dot", cols)
        if cols <= 4099:
            got = nd.matmul(gm, nd.transpose(gw)).cpu().numpy()           # (rows x cols) . (cols x rows)
            ref = m.astype(np.float64) @ w.astype(np.float64).T
            assert (np.abs(got - ref) <= 2e-6 * (np.abs(m).astype(np.float64) @ np.abs(w).astype(np.float64).T)).all(), ("matmul", cols)


@pytest.mark.parametrize("tool,cases", [("fused_static_fuzz.py", "80"), ("gemm_mid_fuzz.py", "60")])
def test_round4_fuzzers_small(tool, cases):
    """The two round-4 fuzzers (tools/: compiled chains against the chain interpreter on random chains / shapes / special
    values; sgemm_dmas_kernel on random shapes, tile shapes, splits and operand offsets inside canary frames), a short run
    each with a seed of its own — the long runs are logged in profiles/r04/fuzz_round4.log."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    p = subprocess.run([sys.executable, str(root / "tools" / tool), cases, "7"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:]
    assert "0 mismatches" in p.stdout, p.stdout[-2000:]
