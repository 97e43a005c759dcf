"""Compiled chains (numpower_amd/csrc/np_fused_static.hip: straight-line kernels for 1-3 step chains from a menu) against
the chain interpreter they replace for those chains (np_elementwise_set_variant(7000) = interpreter only), and both
against the oracle's composition of the reference's CPU functions (arithmetics.c / double_math.c):
  * a chain that stores its value: BIT-identical to the interpreter (same op bodies, np_elementwise_ops.h), which
    tests/test_gpu_fusion.py in turn holds bit-identical to the op-by-op sequence;
  * a chain that ends in a sum (whole array, or over the first axis): within 1e-5 of an fp64 accumulation of the
    oracle's chain values — the compiled kernels fold in a different order (more accumulators per lane);
  * every operand form: full array, row / column broadcast, 0-d device scalar, host scalar, input 0 named again;
  * ragged ends: n % 4 != 0, fewer slots than lanes, one row, a row chunk that is not a multiple of the rows in flight."""
import ctypes as C

import numpy as np
import pytest

from numpower_amd import synth
from numpower_amd._lib import BINARY_OPS, UNARY_OPS, FusedOp, check, load

pytestmark = pytest.mark.gpu

FULL, SCALAR0D, ROW, COL, HOST = 0, 1, 2, 3, 4


def U(op):
    return ("u", op)


def B(op, operand, swap=0):
    return ("b", op, operand, swap)


# (steps, operands beyond input 0 as kinds) — all on the menu unless said otherwise
CHAINS = [
    ([U("exp")], []),
    ([U("log")], []),
    ([B("multiply", 1)], [FULL]),
    ([B("subtract", 1)], [HOST]),
    ([U("exp"), B("add", 1)], [ROW]),
    ([U("exp"), B("add", 1)], [COL]),
    ([U("exp"), B("multiply", 1)], [SCALAR0D]),
    ([U("sqrt"), B("multiply", 1)], [HOST]),
    ([B("subtract", 1), U("abs")], [FULL]),
    ([B("subtract", 1), U("exp")], [COL]),
    ([B("multiply", 1), B("add", 2)], [FULL, ROW]),
    ([B("multiply", 0), B("add", 1)], [FULL]),                       # x * x + b: input 0 named again
    ([U("exp"), B("multiply", 1), B("add", 2)], [FULL, HOST]),      # bench.py's exp(a) * b + 2
    ([B("subtract", 1), U("exp"), B("divide", 2)], [COL, COL]),      # a softmax row: exp(x - m) / s
    ([B("subtract", 1), B("divide", 2), U("exp")], [ROW, HOST]),
    ([B("multiply", 1), B("add", 2), B("add", 3)], [HOST, COL, ROW]),  # examples/kmeans.py: x * -2 + |x|^2 (column) + |c|^2 (row)
    ([U("tanh"), B("add", 1)], [FULL]),                              # NOT on the menu: interpreter both times
    ([B("subtract", 1, 1), U("exp")], [FULL]),                       # swapped operand order: not on the menu either
]
SHAPES = [(1, 5), (3, 8), (1, 1028), (37, 64), (300, 1000), (1, 70001), (515, 516), (2000, 4000), (9000, 36), (3, 1024), (37, 2048)]   # (the last three: cchain_tile2d_kernel)


def _build(steps, kinds, rows, cols, seed):
    n = rows * cols
    arrays = [synth.uniform((n,), seed, 0.25, 2.0)]
    in_kinds = [FULL]
    for i, k in enumerate(kinds):
        size = {FULL: n, ROW: cols, COL: rows, SCALAR0D: 1, HOST: 1}[k]
        arrays.append(synth.uniform((size,), seed + 1 + i, 0.5, 1.5))
        in_kinds.append(k)
    prog = []
    for st in steps:
        if st[0] == "u":
            prog.append(FusedOp(0, UNARY_OPS[st[1]], 0, 0, 0, 0, 0, 0))
        else:
            prog.append(FusedOp(1, BINARY_OPS[st[1]], st[2], st[3], 0, 0, 0, 0))
    return arrays, in_kinds, (FusedOp * len(prog))(*prog)


def _oracle_chain(oracle, steps, arrays, in_kinds, rows, cols):
    def shaped(i):
        k, x = in_kinds[i], arrays[i]
        if k == FULL:
            return x.reshape(rows, cols)
        if k == ROW:
            return np.broadcast_to(x[None, :], (rows, cols)).copy()
        if k == COL:
            return np.broadcast_to(x[:, None], (rows, cols)).copy()
        return np.full((rows, cols), x[0], dtype=np.float32)
    acc = shaped(0)
    for st in steps:
        if st[0] == "u":
            acc = oracle.unary(st[1], acc)
        else:
            other = shaped(st[2])
            acc = oracle.binary(st[1], other, acc) if st[3] else oracle.binary(st[1], acc, other)
    return acc


@pytest.mark.parametrize("ci", range(len(CHAINS)))
def test_compiled_chain_matches_interpreter_and_oracle(ci, hip, oracle):
    steps, kinds = CHAINS[ci]
    lib = load()
    for si, (rows, cols) in enumerate(SHAPES):
        if (ROW in kinds or COL in kinds) and cols % 4 != 0 and cols < 4:
            continue
        n = rows * cols
        arrays, in_kinds, prog = _build(steps, kinds, rows, cols, 900 + 10 * ci + si)
        dev = [None if k == HOST else hip.DeviceArray.from_host(x) for x, k in zip(arrays, in_kinds)]
        ptrs = (C.c_void_p * len(arrays))(*[x.ctypes.data if d is None else d.ptr for x, d in zip(arrays, dev)])
        ckinds = (C.c_int * len(arrays))(*in_kinds)
        want = _oracle_chain(oracle, steps, arrays, in_kinds, rows, cols).astype(np.float64)
        out = hip.DeviceArray((n,))
        red = hip.DeviceArray((cols,))
        red1 = hip.DeviceArray((rows,))
        one = hip.DeviceArray((1,))
        res = C.c_float(0.0)
        stored, sums, colsums, rowsums = [], [], [], []
        for variant in (7000, 0, 7004):
            check(lib.np_elementwise_set_variant(variant))
            try:
                hip.fill(out, float("nan"))
                check(lib.np_fused_chain(ptrs, ckinds, len(arrays), prog, len(prog), out.ptr, rows, cols))
                stored.append(out.to_host().reshape(-1).copy())
                check(lib.np_fused_chain_reduce(ptrs, ckinds, len(arrays), prog, len(prog), 0, rows, cols, C.byref(res)))
                sums.append(float(res.value))
                check(lib.np_fused_chain_reduce_dev(ptrs, ckinds, len(arrays), prog, len(prog), 0, rows, cols, one.ptr))   # device result
                sums.append(float(one.to_host()[0]))
                check(lib.np_fused_chain_reduce_axis(ptrs, ckinds, len(arrays), prog, len(prog), 0, rows, cols, 1, red1.ptr))
                rowsums.append(red1.to_host().astype(np.float64).copy())
                check(lib.np_fused_chain_reduce_axis(ptrs, ckinds, len(arrays), prog, len(prog), 0, rows, cols, 0, red.ptr))
                colsums.append(red.to_host().astype(np.float64).copy())
            finally:
                check(lib.np_elementwise_set_variant(0))
        for got in stored[1:]:
            assert (got.view(np.uint32) == stored[0].view(np.uint32)).all(), (steps, kinds, rows, cols, "store: compiled != interpreter")
        assert (np.abs(stored[0].astype(np.float64) - want.reshape(-1)) <= 1e-5 * np.abs(want.reshape(-1)) + 1e-30).all(), (steps, rows, cols)
        scale = float(np.abs(want).sum())
        for s in sums:
            assert abs(s - float(want.sum())) <= 1e-5 * scale, (steps, kinds, rows, cols, "sum", s, float(want.sum()))
        for cs in colsums:
            assert (np.abs(cs - want.sum(0)) <= 1e-5 * np.abs(want).sum(0)).all(), (steps, kinds, rows, cols, "axis-0 sum")
        for rs in rowsums:
            assert (np.abs(rs - want.sum(1)) <= 1e-5 * np.abs(want).sum(1)).all(), (steps, kinds, rows, cols, "axis-1 sum")


def test_compiled_chain_mean_over_first_axis_and_chunked_rows(hip, oracle):
    """np_fused_chain_reduce_axis(mean) through the compiled column kernel: one chunk (the division happens in the kernel) and
    many chunks (partials folded by np_reduce_axis, then divided), rows not a multiple of the rows in flight."""
    lib = load()
    for rows, cols in ((7, 256), (1001, 512), (40003, 64)):
        a = synth.uniform((rows * cols,), 77, -1.0, 1.0)
        da = hip.DeviceArray.from_host(a)
        prog = (FusedOp * 2)(FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0), FusedOp(1, BINARY_OPS["multiply"], 1, 0, 0, 0, 0, 0))
        half = np.float32([0.5])
        ptrs = (C.c_void_p * 2)(da.ptr, half.ctypes.data)
        kinds = (C.c_int * 2)(FULL, HOST)
        red = hip.DeviceArray((cols,))
        check(lib.np_fused_chain_reduce_axis(ptrs, kinds, 2, prog, 2, 4, rows, cols, 0, red.ptr))      # 4 = NP_MEAN
        vals = oracle.binary("multiply", oracle.unary("exp", a.reshape(rows, cols)), np.float32(0.5)).astype(np.float64)
        assert (np.abs(red.to_host().astype(np.float64) - vals.mean(0)) <= 1e-5 * np.abs(vals).mean(0)).all(), (rows, cols)


QUIRK_CHAINS = [
    ([B("multiply", 1)], [FULL]),
    ([U("exp"), B("multiply", 1), B("add", 2)], [FULL, HOST]),      # nd::exp($x) * $y + 2 as the PHP binding's chain carries it
    ([B("multiply", 1), B("add", 2)], [FULL, ROW]),
    ([B("subtract", 1), B("multiply", 2)], [FULL, HOST]),            # ($x - $y) * 0.0-ish: zero products from a scalar
    ([B("multiply", 1), B("multiply", 2), B("add", 3)], [COL, FULL, FULL]),
    ([U("abs"), B("multiply", 1)], [ROW]),
]


@pytest.mark.parametrize("ci", range(len(QUIRK_CHAINS)))
def test_compiled_chains_carry_the_avx_body_quirk_of_multiply(ci, hip, oracle):
    """What the PHP binding's chains look like: every multiply step carries NP_QUIRK_AVX_BODY + the index where the reference's AVX2
    body ends (zero products are -0.0f inside the body, +0.0f in the scalar tail, arithmetics.c:403,410-412).  Round 6 (second half):
    the compiled kernels know that flag (flat store / flat sum / last-axis sum / the 2-D broadcast form), so such chains no longer fall
    back to the interpreter — bit-identical to the interpreter (7000) and to the oracle's op-by-op composition, zeros and negative zeros
    included; first-axis sums of such chains still run on the interpreter."""
    steps, kinds = QUIRK_CHAINS[ci]
    lib = load()
    for si, (rows, cols) in enumerate([(1, 5), (3, 8), (1, 1031), (37, 64), (300, 1000), (515, 516), (9000, 36), (3, 1024), (37, 2048)]):
        n = rows * cols
        arrays, in_kinds, _ = _build(steps, kinds, rows, cols, 1900 + 10 * ci + si)
        arrays[0] = synth.uniform((n,), 1950 + ci + si, -1.0, 1.0)
        arrays[0][::3] = 0.0                                   # zero products in the body and in the tail
        arrays[0][1::7] = -0.0
        for i, k in enumerate(in_kinds[1:], 1):
            if k == HOST and steps[-1][1] == "multiply":
                arrays[i][:] = 0.0                               # a scalar that makes EVERY product zero
        body_end = n - n % 8                                     # np_avx_body_end(n): what the host layer passes for equal shapes
        prog = []
        for st in steps:
            if st[0] == "u":
                prog.append(FusedOp(0, UNARY_OPS[st[1]], 0, 0, 0, 0, 0, 0))
            else:
                quirk = st[1] == "multiply"
                prog.append(FusedOp(1, BINARY_OPS[st[1]], st[2], st[3], 0, 0, 1 if quirk else 0, body_end if quirk else 0))
        prog = (FusedOp * len(prog))(*prog)
        dev = [None if k == HOST else hip.DeviceArray.from_host(x) for x, k in zip(arrays, in_kinds)]
        ptrs = (C.c_void_p * len(arrays))(*[x.ctypes.data if d is None else d.ptr for x, d in zip(arrays, dev)])
        ckinds = (C.c_int * len(arrays))(*in_kinds)
        out, red1, red0, res = hip.DeviceArray((n,)), hip.DeviceArray((rows,)), hip.DeviceArray((cols,)), C.c_float(0.0)
        stored, sums, rowsums, colsums = [], [], [], []
        for variant in (7000, 0):
            check(lib.np_elementwise_set_variant(variant))
            try:
                hip.fill(out, float("nan"))
                check(lib.np_fused_chain(ptrs, ckinds, len(arrays), prog, len(prog), out.ptr, rows, cols))
                stored.append(out.to_host().reshape(-1).copy())
                check(lib.np_fused_chain_reduce(ptrs, ckinds, len(arrays), prog, len(prog), 0, rows, cols, C.byref(res)))
                sums.append(float(res.value))
                check(lib.np_fused_chain_reduce_axis(ptrs, ckinds, len(arrays), prog, len(prog), 0, rows, cols, 1, red1.ptr))
                rowsums.append(red1.to_host().copy())
                check(lib.np_fused_chain_reduce_axis(ptrs, ckinds, len(arrays), prog, len(prog), 0, rows, cols, 0, red0.ptr))
                colsums.append(red0.to_host().copy())
            finally:
                check(lib.np_elementwise_set_variant(0))
        assert (stored[0].view(np.uint32) == stored[1].view(np.uint32)).all(), (steps, kinds, rows, cols, "store: compiled != interpreter")
        if steps[-1][0] == "b" and steps[-1][1] == "multiply" and n >= 24:
            assert (np.signbit(stored[1]) & (stored[1] == 0)).any(), "the case must produce negative zeros"
        # the oracle's composition with the reference's body / tail split of the flat index
        flat = np.arange(n).reshape(rows, cols)

        def shaped(i):
            k, x = in_kinds[i], arrays[i]
            if k == FULL:
                return x.reshape(rows, cols)
            if k == ROW:
                return np.broadcast_to(x[None, :], (rows, cols)).copy()
            if k == COL:
                return np.broadcast_to(x[:, None], (rows, cols)).copy()
            return np.full((rows, cols), x[0], dtype=np.float32)
        acc = shaped(0)
        for st in steps:
            if st[0] == "u":
                acc = oracle.unary(st[1], acc)
            elif st[1] == "multiply":
                p = (acc * shaped(st[2])).astype(np.float32)
                p[(p == 0) & (flat < body_end)] = np.float32(-0.0)
                p[(p == 0) & (flat >= body_end)] = np.float32(0.0)
                acc = p
            else:
                acc = oracle.binary(st[1], acc, shaped(st[2]))
        if all(st[0] == "b" or st[1] in ("abs",) for st in steps):          # exact ops only: bit for bit
            assert (stored[1].view(np.uint32) == acc.reshape(-1).view(np.uint32)).all(), (steps, kinds, rows, cols, "vs the composition")
        w64 = acc.astype(np.float64)
        assert abs(sums[0] - sums[1]) <= 2e-6 * np.abs(w64).sum() + 1e-30 and abs(sums[1] - w64.sum()) <= 1e-5 * np.abs(w64).sum() + 1e-30
        assert (np.abs(rowsums[1] - w64.sum(1)) <= 1e-5 * np.abs(w64).sum(1) + 1e-30).all()
        assert (np.abs(colsums[1] - w64.sum(0)) <= 1e-5 * np.abs(w64).sum(0) + 1e-30).all()
