"""Dev tool: pow on 1e8 floats under different grid caps (np_elementwise_set_variant(100 + 10 * b): 2 b workgroups per
CU, non-temporal accesses, UNROLL 2; 0 = the default = 8 per CU for pow), add as the yardstick.
Usage: python tools/pow_grid_ab.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from numpower_amd import device as D, synth
from numpower_amd._lib import Timer, check, load

D.init(0)
lib = load()
N = 100_000_000
a = D.DeviceArray.from_host(synth.uniform((N,), 5))
b = D.DeviceArray.from_host(synth.uniform((N,), 6))
o = D.DeviceArray((N,))
t = Timer()


def run(op, iters=20):
    for _ in range(3):
        D.binary(op, a, "full", b, "full", 1, N, out=o)
    D.sync()
    t.start()
    for _ in range(iters):
        D.binary(op, a, "full", b, "full", 1, N, out=o)
    t.stop()
    return t.elapsed_ms() / iters * 1e3


for rnd in range(3):
    print("-- round", rnd, flush=True)
    print("   add                      %6.1f us" % run("add"))
    for v in (0, 110, 120, 130, 140, 160, 180, 190):
        check(lib.np_elementwise_set_variant(v))
        us = run("pow")
        print("   pow variant %3d          %6.1f us  %5.0f GB/s" % (v, us, 12.0 * N / us / 1e3), flush=True)
    check(lib.np_elementwise_set_variant(0))
