"""Is the add-1e8 rate a function of what the GPU did just before?  Batches of 25 launches (bench.py's protocol)
back to back for ~1.5 s from a cold start, then again after 0.5 s / 2 s / 8 s of idling, then right after a
burst of 100 GEMMs; GB/s per batch.  Usage: python tools/add_ramp.py"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from numpower_amd import device as D, synth
from numpower_amd._lib import Timer

D.init(0)
N = 100_000_000
a = D.DeviceArray.from_host(synth.uniform((N,), 5))
b = D.DeviceArray.from_host(synth.uniform((N,), 6))
o = D.DeviceArray((N,))
n = 4096
A = D.DeviceArray.from_host(synth.uniform((n, n), 3, -1, 1))
B = D.DeviceArray.from_host(synth.uniform((n, n), 4, -1, 1))
Cm = D.DeviceArray((n, n))


def batches(k, label):
    out = []
    for _ in range(k):
        t = Timer()
        t.start()
        for _ in range(25):
            D.binary("add", a, "full", b, "full", 1, N, out=o)
        t.stop()
        out.append(12.0 * N / (t.elapsed_ms() / 25) / 1e6)
    print("%-34s %s" % (label, " ".join("%.0f" % x for x in out)), flush=True)


time.sleep(3.0)
batches(40, "cold start (3 s idle)")
for gap in (0.5, 2.0, 8.0):
    time.sleep(gap)
    batches(12, "after %.1f s idle" % gap)
for _ in range(100):
    D.sgemm(A, B, out=Cm)
batches(12, "right after 100 GEMMs")
for _ in range(400):
    D.sgemm(A, B, out=Cm)
batches(12, "right after 400 GEMMs")
