"""Dev tool: pow on 1e8 floats under different grid caps (np_elementwise_set_variant(100 + 10 * b): 2 b workgroups per
CU, non-temporal accesses, UNROLL 2; 0 = the default: uncapped, the log2 table in two VGPRs read by ds_bpermute;
9000 = the per-wave LDS copy of the table instead), add as the yardstick.  The register-table form is first checked bit
for bit against the LDS form on ragged sizes (partial last waves: bpermute needs the table's lanes active).
Usage: python tools/pow_grid_ab.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
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


import numpy as np
for n in (1, 3, 4, 255, 256, 257, 1000, 4099, 65536 + 5, 1_000_003):
    x = synth.uniform((n,), 11) * 4.0 + 0.01
    y = synth.uniform((n,), 12) * 6.0 - 3.0
    if n > 256:
        x[5], x[77], x[200] = -2.0, 0.0, np.inf
        y[5], y[78], y[201] = 3.0, np.nan, -np.inf
    da, db = D.DeviceArray.from_host(x), D.DeviceArray.from_host(y)
    check(lib.np_elementwise_set_variant(9000))
    ref = D.binary("pow", da, "full", db, "full", 1, n).to_host()
    check(lib.np_elementwise_set_variant(0))
    got = D.binary("pow", da, "full", db, "full", 1, n).to_host()
    check(lib.np_elementwise_set_variant(0))
    assert np.array_equal(ref.view(np.uint32), got.view(np.uint32)), n
print("register-table pow bit-identical to the LDS-table form on 10 sizes", flush=True)

for rnd in range(3):
    print("-- round", rnd, flush=True)
    print("   add                      %6.1f us" % run("add"))
    for v in (0, 9000, 0, 9000):
        check(lib.np_elementwise_set_variant(v))
        us = run("pow")
        print("   pow variant %3d          %6.1f us  %5.0f GB/s" % (v, us, 12.0 * N / us / 1e3), flush=True)
    check(lib.np_elementwise_set_variant(0))
