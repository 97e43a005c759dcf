"""Where in a run the headline's timed region sits: W = 5 untimed + K = 20 timed 4096^3 products (the driver's flags)
(a) first thing in a fresh process, (b) again after 2 s of idling, (c) right behind ~1 s of HBM-bound launches (what the
second half of the metric and the other configs are), (d / e) the same with 100 / 500 ms of idling in between.
The kernel is the same every time; what differs is the power state the device is in when the region starts.
Usage: python tools/headline_order_ab.py [rounds]"""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from numpower_amd import device as D          # noqa: E402
from numpower_amd import synth                 # noqa: E402
from numpower_amd._lib import Timer, check, load   # noqa: E402

n = 4096
A = synth.uniform((n, n), 3, -1.0, 1.0)
B = synth.uniform((n, n), 4, -1.0, 1.0)
dA, dB, dC = D.DeviceArray.from_host(A), D.DeviceArray.from_host(B), D.DeviceArray((n, n))
lib = load()


def region(w=5, k=20):
    for _ in range(w):
        D.sgemm(dA, dB, out=dC)
    check(lib.np_sync())
    t = Timer()
    t0 = time.perf_counter()
    t.start()
    for _ in range(k):
        D.sgemm(dA, dB, out=dC)
    t.stop()
    check(lib.np_sync())
    wall = time.perf_counter() - t0
    return 2.0 * n ** 3 * k / wall / 1e12, 2.0 * n ** 3 * k / (t.elapsed_ms() * 1e-3) / 1e12


first = region()
print("(a) first thing in the process        %.1f TFLOP/s wall  %.1f events" % first, flush=True)
N = 100_000_000
a, b, o = D.DeviceArray((N,)), D.DeviceArray((N,)), D.DeviceArray((N,))
D.fill(a, 0.25)
D.fill(b, 0.5)


def hbm_load(seconds):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            D.binary("add", a, "full", b, "full", 1, N, out=o)
        check(lib.np_sync())


for r in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    time.sleep(2.0)
    print("(b) after 2 s of idling                %.1f TFLOP/s wall  %.1f events" % region(), flush=True)
    for idle in (0.0, 0.1, 0.5):
        time.sleep(1.0)
        hbm_load(1.0)
        time.sleep(idle)
        print("(c) behind 1 s of adds + %3d ms idle   %.1f TFLOP/s wall  %.1f events" % ((idle * 1e3,) + region()), flush=True)
    hbm_load(1.0)
    print("(f) behind 1 s of adds, W = 5 K = 50   %.1f TFLOP/s wall  %.1f events" % region(5, 50), flush=True)
    time.sleep(1.0)
    for _ in range(100):
        D.sgemm(dA, dB, out=dC)
    check(lib.np_sync())
    print("(g) behind 100 products                %.1f TFLOP/s wall  %.1f events" % region(), flush=True)
