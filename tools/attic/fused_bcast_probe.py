"""Dev tool: the fused-chain kernel on 25000 x 4000 with a row or a column operand, with and without the exp in front —
where does exp(X) + col lose its 10 % against exp(X) + row?  Usage: python tools/fused_bcast_probe.py"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from numpower_amd import device as D, synth
from numpower_amd._lib import BINARY_OPS, UNARY_OPS, FusedOp, Timer, check, load
D.init(0); lib = load()
R, Cc = 25000, 4000
N = R * Cc
a = D.DeviceArray.from_host(synth.uniform((N,), 5))
o = D.DeviceArray((N,))
row = D.DeviceArray.from_host(synth.uniform((Cc,), 9))
col = D.DeviceArray.from_host(synth.uniform((R,), 10))
EXP = FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0)
ADD = FusedOp(1, BINARY_OPS["add"], 1, 0, 0, 0, 0, 0)
cases = [("exp(X)", [EXP], None, 0), ("exp(X) + row", [EXP, ADD], row, 2), ("exp(X) + col", [EXP, ADD], col, 3),
         ("X + row", [ADD], row, 2), ("X + col", [ADD], col, 3), ("X + X", [ADD], a, 0)]
t = Timer()
for rnd in range(3):
    for name, ops, vec, kind in cases:
        prog = (FusedOp * len(ops))(*ops)
        n_in = 1 if vec is None else 2
        ptrs = (C.c_void_p * n_in)(*([a.ptr] if vec is None else [a.ptr, vec.ptr]))
        kinds = (C.c_int * n_in)(*([0] if vec is None else [0, kind]))
        run = lambda: check(lib.np_fused_chain(ptrs, kinds, n_in, prog, len(ops), o.ptr, R, Cc))
        for _ in range(3): run()
        D.sync(); t.start()
        for _ in range(20): run()
        t.stop()
        print("round %d  %-14s %7.1f us" % (rnd, name, t.elapsed_ms() / 20 * 1e3), flush=True)
    for name, kind, vec in (("np_binary X + row", "row", row), ("np_binary X + col", "col", col)):
        run = lambda: D.binary("add", a, "full", vec, kind, R, Cc, out=o)
        for _ in range(3): run()
        D.sync(); t.start()
        for _ in range(20): run()
        t.stop()
        print("round %d  %-14s %7.1f us" % (rnd, name, t.elapsed_ms() / 20 * 1e3), flush=True)
