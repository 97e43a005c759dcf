"""Chain ends that reduce over an axis: sum(exp(X), axis) fused (np_fused_chain_reduce_axis) vs
exp -> temporary -> np_reduce_axis.  Usage: python tools/fused_axis_ab.py"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from numpower_amd import device as D, _lib
from numpower_amd._lib import UNARY_OPS, FusedOp, Timer, check
D.init(0); lib = _lib.load(); t = Timer()
prog = (FusedOp * 1)(FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0))
for rows, cols in ((25000, 4000), (65536, 4096), (1_000_000, 100), (4_000_000, 24), (200_000, 512), (4096, 25000), (400, 250_000), (3, 30_000_000), (100_000, 1000)):
    n = rows * cols
    x = D.DeviceArray((rows, cols)); D.fill(x, 0.5); tmp = D.DeviceArray((rows, cols))
    ptrs = (C.c_void_p * 1)(x.ptr); kinds = (C.c_int * 1)(0)
    for axis in (1, 0):
        out = D.DeviceArray((rows if axis == 1 else cols,))
        def fused(): check(lib.np_fused_chain_reduce_axis(ptrs, kinds, 1, prog, 1, 0, rows, cols, axis, out.ptr))
        def unfused():
            D.unary("exp", x, out=tmp)
            check(lib.np_reduce_axis(0, tmp.ptr, rows if axis == 1 else 1, cols if axis == 1 else rows, 1 if axis == 1 else cols, out.ptr, 0))
        res = []
        for fn in (fused, unfused):
            for _ in range(2): fn()
            D.sync(); t.start()
            for _ in range(5): fn()
            t.stop(); res.append(t.elapsed_ms() / 5)
        print("  %8d x %7d axis %d: fused %7.3f ms (%5.0f GB/s over 4 B/elem)   unfused %7.3f ms   x%.2f"
              % (rows, cols, axis, res[0], 4.0 * n / res[0] / 1e6, res[1], res[1] / res[0]), flush=True)
        out.free()
    x.free(); tmp.free()
