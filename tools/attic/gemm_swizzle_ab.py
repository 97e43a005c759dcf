import sys
sys.path.insert(0, '.')
import numpy as np
from numpower_amd import device as D, synth
from numpower_amd._lib import Timer, check, load
D.init(0); lib = load()
n = 4096
A = D.DeviceArray.from_host(synth.uniform((n, n), 3, -1, 1)); B = D.DeviceArray.from_host(synth.uniform((n, n), 4, -1, 1)); Cm = D.DeviceArray((n, n))
for _ in range(60): D.sgemm(A, B, out=Cm)
res = {}
for rnd in range(3):
    for v in (0, 7, 27, 47, 87, 167):
        check(lib.np_sgemm_set_variant(v))
        for _ in range(5): D.sgemm(A, B, out=Cm)
        D.sync(); t = Timer(); t.start()
        for _ in range(40): D.sgemm(A, B, out=Cm)
        t.stop(); res.setdefault(v, []).append(2.0 * n ** 3 / (t.elapsed_ms() / 40) / 1e9)
check(lib.np_sgemm_set_variant(0))
for v, r in res.items(): print("variant %3d (swizzle group %2d): %s" % (v, v // 10, " ".join("%.1f" % x for x in r)))
