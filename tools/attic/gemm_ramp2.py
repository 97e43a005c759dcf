"""What is the 30-launch descent of the 4096^3 GEMM (tools/gemm_ramp.py)?  Power management or cache warm-up:
the same burst (a) on 2048^3 (48 MB working set, fits any cache), (b) on 4096^3 with all-zero inputs (same
addresses and traffic, far less switching power), (c) on 4096^3 random inputs again.  Usage: python tools/gemm_ramp2.py"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from numpower_amd import device as D, synth
from numpower_amd._lib import Timer

D.init(0)


def burst(A, B, Cm, k, label):
    ts = [Timer() for _ in range(k)]
    for t in ts:
        t.start()
        D.sgemm(A, B, out=Cm)
        t.stop()
    print("%-38s %s" % (label, " ".join("%.0f" % (t.elapsed_ms() * 1e3) for t in ts)), flush=True)


for n, kind in ((2048, "random"), (4096, "zeros"), (4096, "random"), (4096, "zeros"), (8192, "random")):
    if kind == "random":
        A, B = D.DeviceArray.from_host(synth.uniform((n, n), 3, -1, 1)), D.DeviceArray.from_host(synth.uniform((n, n), 4, -1, 1))
    else:
        A, B = D.DeviceArray((n, n)), D.DeviceArray((n, n))
        D.fill(A, 0.0)
        D.fill(B, 0.0)
    Cm = D.DeviceArray((n, n))
    D.sync()
    time.sleep(3.0)
    burst(A, B, Cm, 45 if n < 8192 else 12, "%d^3 %s, 3 s idle:" % (n, kind))
    for d in (A, B, Cm):
        d.free()
