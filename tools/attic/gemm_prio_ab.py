"""A/B of the priority alternation between the two workgroups of a CU in sgemm_dma_kernel (np_sgemm_set_variant
(-(100 + p)): p K-tiles per phase, p = 0 off) at 4096^3: interleaved rounds, 40 back-to-back launches per
measurement, plus a bit-for-bit comparison of the products (the alternation changes timing, not arithmetic).
Usage: python tools/gemm_prio_ab.py [n]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from numpower_amd import device as D, synth
from numpower_amd._lib import Timer, check, load

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
D.init(0)
lib = load()
A = D.DeviceArray.from_host(synth.uniform((n, n), 3, -1, 1))
B = D.DeviceArray.from_host(synth.uniform((n, n), 4, -1, 1))
Cm = D.DeviceArray((n, n))
periods = [0, 2, 4, 8, 16, 32, 64]


def run(iters=40):
    for _ in range(5):
        D.sgemm(A, B, out=Cm)
    D.sync()
    t = Timer()
    t.start()
    for _ in range(iters):
        D.sgemm(A, B, out=Cm)
    t.stop()
    return 2.0 * n ** 3 / (t.elapsed_ms() / iters) / 1e9


ref = None
res = {p: [] for p in periods}
for rnd in range(4):
    for p in periods:
        check(lib.np_sgemm_set_variant(-(100 + p)))
        res[p].append(run())
        if rnd == 0:
            got = Cm.to_host()
            if ref is None:
                ref = got
            assert (got.view(np.uint32) == ref.view(np.uint32)).all(), "period %d changes the product" % p
check(lib.np_sgemm_set_variant(-100))
for p in periods:
    print("prio period %3d  TFLOP/s: %s   median %.1f" % (p, " ".join("%.1f" % x for x in res[p]), float(np.median(res[p]))))
