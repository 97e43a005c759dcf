"""A/B harness (dev tool): elementwise launch-config variants on add / exp at 1e8 floats, random data."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from numpower_amd import device as D, synth
from numpower_amd._lib import load, Timer
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 and sys.argv[1][0] != "-" else [104, 101, 102, 108]
D.init(0); lib = load()
N = 100_000_000
a = D.DeviceArray.from_host(synth.uniform((N,), 5)); b = D.DeviceArray.from_host(synth.uniform((N,), 6)); o = D.DeviceArray((N,))
PER_LAUNCH = "--per-launch" in sys.argv   # median of individually timed launches (how tools/explore/add_bw times)
def t(fn, n=20):
    for _ in range(3): fn()
    D.sync()
    if PER_LAUNCH:
        ts = []
        for _ in range(n):
            tm = Timer(); tm.start(); fn(); tm.stop(); ts.append(tm.elapsed_ms())
        return float(np.median(ts))
    tm = Timer(); tm.start()
    for _ in range(n): fn()
    tm.stop(); return tm.elapsed_ms() / n
res = {}
for r in range(4):
    for v in variants:
        lib.np_elementwise_set_variant(v)
        res.setdefault(("add", v), []).append(1.2e9 / t(lambda: D.binary("add", a, "full", b, "full", 1, N, out=o)) / 1e6)
        res.setdefault(("exp", v), []).append(0.8e9 / t(lambda: D.unary("exp", a, out=o)) / 1e6)
        res.setdefault(("addrow", v), []).append(0.8e9 / t(lambda: D.binary("add", a, "full", b, "row", 25000, 4000, out=o)) / 1e6)
for k, v in sorted(res.items()):
    print("%-8s variant %4d  GB/s: %s  median %.0f" % (k[0], k[1], " ".join("%.0f" % x for x in v), float(np.median(v))))
