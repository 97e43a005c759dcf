"""Per-launch time of the 4096^3 GEMM from an idle chip: 3 s of idling, then 60 launches back to back, each bracketed by
its own event pair; again after the H2D copies bench.py does right before its warm-up.  Shows how much of a
20-step timed region (the driver's K) is clock ramp.  Usage: python tools/gemm_ramp.py"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from numpower_amd import device as D, synth
from numpower_amd._lib import Timer

D.init(0)
n = 4096
hA, hB = synth.uniform((n, n), 3, -1, 1), synth.uniform((n, n), 4, -1, 1)
A, B, Cm = D.DeviceArray.from_host(hA), D.DeviceArray.from_host(hB), D.DeviceArray((n, n))


def burst(k, label):
    ts = [Timer() for _ in range(k)]
    for t in ts:
        t.start()
        D.sgemm(A, B, out=Cm)
        t.stop()
    ms = [t.elapsed_ms() for t in ts]
    print("%-34s %s" % (label, " ".join("%.0f" % (x * 1e3) for x in ms)), flush=True)
    return ms


time.sleep(3.0)
burst(60, "after 3 s idle (us per launch)")
time.sleep(3.0)
A2, B2 = D.DeviceArray.from_host(hA), D.DeviceArray.from_host(hB)     # what precedes bench.py's warm-up
burst(60, "after idle + 2 x 64 MB H2D")
time.sleep(0.5)
burst(30, "after 0.5 s idle")

# does other GPU work bring the clock up?  50 ms of the HBM-bound add kernel, resp. of exp (more VALU), then the GEMMs
N = 100_000_000
x, y, z = D.DeviceArray((N,)), D.DeviceArray((N,)), D.DeviceArray((N,))
D.fill(x, 0.5)
D.fill(y, 0.25)
for label, fn, reps in (("add", lambda: D.binary("add", x, "full", y, "full", 1, N, out=z), 260),
                        ("exp", lambda: D.unary("exp", x, out=z), 400),
                        ("gemm 2048^3 x 12", None, 12)):
    time.sleep(3.0)
    if fn is None:
        m = 2048
        a2, b2, c2 = D.DeviceArray((m, m)), D.DeviceArray((m, m)), D.DeviceArray((m, m))
        D.fill(a2, 0.5)
        D.fill(b2, 0.25)
        for _ in range(reps * 25):
            D.sgemm(a2, b2, out=c2)
    else:
        for _ in range(reps):
            fn()
    burst(30, "3 s idle, ~50 ms of %s, then:" % label)
