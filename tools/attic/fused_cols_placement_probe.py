"""Dev tool: is sum(exp(X), 0)'s run-to-run spread (74 vs 90 us on 25000 x 4000) a matter of WHICH pool block serves the
1 MB of chunk partials?  Time the call repeatedly while holding 0, 1, 2 ... dummy 1 MB blocks (each makes the call's
scratch come from a different cached block), and after the row-kernel form of the same chain (the sequence in which
tools/fused_cols_ab.py saw the slow mode).  Usage: python tools/fused_cols_placement_probe.py"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from numpower_amd import device as D, synth
from numpower_amd._lib import UNARY_OPS, FusedOp, Timer, check, load

D.init(0)
lib = load()
R, Cc = 25000, 4000
da = D.DeviceArray.from_host(synth.uniform((R * Cc,), 5, 0.0, 1.0))
prog = (FusedOp * 1)(FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0))
ptrs = (C.c_void_p * 1)(da.ptr)
kinds = (C.c_int * 1)(0)
out0, out1 = D.DeviceArray((Cc,)), D.DeviceArray((R,))
t = Timer()


def run(axis, out, iters=30):
    for _ in range(3):
        check(lib.np_fused_chain_reduce_axis(ptrs, kinds, 1, prog, 1, 0, R, Cc, axis, out.ptr))
    D.sync()
    t.start()
    for _ in range(iters):
        check(lib.np_fused_chain_reduce_axis(ptrs, kinds, 1, prog, 1, 0, R, Cc, axis, out.ptr))
    t.stop()
    return t.elapsed_ms() / iters * 1e3


held = []
for k in range(6):
    line = "held %d dummy 1 MB blocks:" % k
    for rep in range(3):
        line += "  cols %.1f us" % run(0, out0)
    line += "   | after rows (%.1f us): cols %.1f us" % (run(1, out1), run(0, out0))
    print(line, flush=True)
    p = C.c_void_p()
    check(lib.np_malloc(C.byref(p), 1 << 20))
    held.append(p)
    print("   dummy block at %#x" % p.value)
for p in held:
    check(lib.np_free(p))
