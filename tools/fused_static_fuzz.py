"""Random-shape fuzz of the compiled chains against the chain interpreter (np_elementwise_set_variant(7000)): random menu chains,
operand kinds, shapes (ragged, tiny, one row, one column), special values sprinkled in.  Stored values must be bit-identical
(NaN patterns included); sums / axis sums within 1e-5 of the interpreter's (scaled by the sum of magnitudes).
Usage: python tools/fused_static_fuzz.py [cases = 400] [seed = 1]"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from numpower_amd import device as D
from numpower_amd._lib import BINARY_OPS, UNARY_OPS, FusedOp, check, load

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
D.init(0)
lib = load()
UN = ["exp", "log", "sqrt", "abs", "negate"]
BI = ["add", "subtract", "multiply", "divide"]
FULL, S0D, ROW, COL, HOST = 0, 1, 2, 3, 4
bad = 0
for case in range(cases):
    rows = int(rng.choice([1, 2, 3, 7, 64, 257, 1000, 4099]))
    cols = int(rng.choice([1, 4, 5, 8, 60, 64, 132, 1000, 4096, 10007]))
    if rows * cols > 6_000_000:
        cols = 132
    n = rows * cols
    nsteps = int(rng.integers(1, 4))
    arrays = [rng.uniform(0.25, 2.0, n).astype(np.float32)]
    kinds = [FULL]
    prog = []
    for _ in range(nsteps):
        if rng.random() < 0.45:
            prog.append(FusedOp(0, UNARY_OPS[str(rng.choice(UN))], 0, 0, 0, 0, 0, 0))
        else:
            k = int(rng.choice([FULL, S0D, ROW, COL, HOST]))
            if k in (ROW, COL) and cols % 4 != 0 and cols < 4:
                k = HOST
            size = {FULL: n, ROW: cols, COL: rows, S0D: 1, HOST: 1}[k]
            arrays.append(rng.uniform(0.5, 1.5, size).astype(np.float32))
            kinds.append(k)
            prog.append(FusedOp(1, BINARY_OPS[str(rng.choice(BI))], len(arrays) - 1, int(rng.random() < 0.15), 0, 0, 0, 0))
    if rng.random() < 0.3:      # special values in input 0
        idx = rng.integers(0, n, size=min(n, 5))
        arrays[0][idx] = rng.choice(np.float32([0.0, -0.0, np.inf, -np.inf, np.nan, -1.0, 1e-42, 3e38]), size=idx.size)
    dev = [None if k == HOST else D.DeviceArray.from_host(x) for x, k in zip(arrays, kinds)]
    ptrs = (C.c_void_p * len(arrays))(*[x.ctypes.data if d is None else d.ptr for x, d in zip(arrays, dev)])
    ck = (C.c_int * len(arrays))(*kinds)
    pr = (FusedOp * len(prog))(*prog)
    out, red0, red1, one = D.DeviceArray((n,)), D.DeviceArray((cols,)), D.DeviceArray((rows,)), D.DeviceArray((1,))
    got = {}
    for v in (7000, 0):
        check(lib.np_elementwise_set_variant(v))
        D.fill(out, 7.0)
        check(lib.np_fused_chain(ptrs, ck, len(arrays), pr, len(prog), out.ptr, rows, cols))
        check(lib.np_fused_chain_reduce_dev(ptrs, ck, len(arrays), pr, len(prog), 0, rows, cols, one.ptr))
        check(lib.np_fused_chain_reduce_axis(ptrs, ck, len(arrays), pr, len(prog), 0, rows, cols, 0, red0.ptr))
        check(lib.np_fused_chain_reduce_axis(ptrs, ck, len(arrays), pr, len(prog), 0, rows, cols, 1, red1.ptr))
        got[v] = (out.to_host().reshape(-1).copy(), float(one.to_host()[0]), red0.to_host().copy(), red1.to_host().copy())
    check(lib.np_elementwise_set_variant(0))
    a, b = got[7000], got[0]
    same = ((a[0].view(np.uint32) == b[0].view(np.uint32)) | (np.isnan(a[0]) & np.isnan(b[0]))).all()
    vals = a[0].astype(np.float64)
    finite = np.isfinite(vals).all()
    ok = bool(same)
    if finite:
        scale = np.abs(vals).sum() + 1e-30
        # (a sum of finite fp32 values may still overflow — a 3e38 special times 1.2 next to 4098 ordinary ones: both forms then
        # hold the same infinity, and inf - inf is not a difference; equal values are equal)
        close = lambda x, y, tol: bool(np.all((np.asarray(x) == np.asarray(y)) | (np.abs(np.asarray(x, np.float64) - np.asarray(y, np.float64)) <= tol)))
        ok = ok and close(a[1], b[1], 1e-5 * scale)
        m = np.abs(vals.reshape(rows, cols))
        with np.errstate(invalid="ignore"):
            ok = ok and close(a[2], b[2], 1e-5 * (m.sum(0) + 1e-30))
            ok = ok and close(a[3], b[3], 1e-5 * (m.sum(1) + 1e-30))
    if not ok:
        bad += 1
        print("MISMATCH case %d: rows %d cols %d kinds %s steps %s stored-identical %s" % (
            case, rows, cols, kinds, [(p.kind, p.op, p.operand, p.swap) for p in prog], bool(same)), flush=True)
        if finite:      # which of the three reductions disagrees, and by how much
            print("    sum: interpreter %r compiled %r (scale %.6g)" % (a[1], b[1], scale), flush=True)
            d0 = np.abs(a[2].astype(np.float64) - b[2]); d1 = np.abs(a[3].astype(np.float64) - b[3])
            print("    axis 0: max |diff| %.6g at %d (%r vs %r); axis 1: max |diff| %.6g at %d (%r vs %r)" % (
                d0.max(), int(d0.argmax()), a[2][d0.argmax()], b[2][d0.argmax()], d1.max(), int(d1.argmax()), a[3][d1.argmax()], b[3][d1.argmax()]), flush=True)
            print("    reference: sum %.9g, axis-0[0] %.9g" % (vals.sum(), vals.reshape(rows, cols).sum(0)[0]), flush=True)
    for d in dev:
        if d is not None:
            d.free()
    for d in (out, red0, red1, one):
        d.free()
print("%d cases, %d mismatches" % (cases, bad))
sys.exit(1 if bad else 0)
