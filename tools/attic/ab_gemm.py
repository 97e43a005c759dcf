"""One build's 4096^3 GEMM rate for same-box A/B of two builds (NP_HIP_LIB): launches 1-25 from idle (the window the
driver's K = 20 / W = 5 sees) and the plateau (launches 41-100).  Usage: NP_HIP_LIB=... python tools/ab_gemm.py"""
import sys
import time
sys.path.insert(0, '.')
from numpower_amd import device as D, synth
from numpower_amd._lib import Timer
D.init(0)
n = 4096
A = D.DeviceArray.from_host(synth.uniform((n, n), 3, -1, 1)); B = D.DeviceArray.from_host(synth.uniform((n, n), 4, -1, 1)); Cm = D.DeviceArray((n, n))
D.sync(); time.sleep(1.0)
ts = [Timer() for _ in range(100)]
for t in ts:
    t.start(); D.sgemm(A, B, out=Cm); t.stop()
ms = [t.elapsed_ms() for t in ts]
flop = 2.0 * n ** 3
win = sum(ms[5:25]) / 20
plateau = sorted(ms[40:])[30]
print("launches 6-25: %.1f TFLOP/s   plateau (median of 41-100): %.1f TFLOP/s" % (flop / win / 1e9, flop / plateau / 1e9))
