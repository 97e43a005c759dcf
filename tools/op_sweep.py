"""Throughput of every elementwise op of the C ABI at 10^8 fp32 elements (HBM roofline view):
37 unary ops (8 B/elem) and 13 binary ops (12 B/elem), random inputs in each op's domain.
Usage: python tools/op_sweep.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from numpower_amd import device as D, synth
from numpower_amd._lib import BINARY_OPS, UNARY_OPS, Timer
D.init(0)
N = 100_000_000
DOMAIN = {"sqrt": (0.0, 100.0), "log": (1e-3, 1e3), "log2": (1e-3, 1e3), "log10": (1e-3, 1e3), "log1p": (-0.5, 100.0),
          "logb": (1e-3, 1e3), "arcsin": (-1.0, 1.0), "arccos": (-1.0, 1.0), "arccosh": (1.0, 100.0),
          "arctanh": (-0.99, 0.99), "rsqrt": (1e-3, 1e3), "reciprocal": (0.1, 10.0), "exp": (-10.0, 10.0),
          "exp2": (-10.0, 10.0), "expm1": (-10.0, 10.0), "sinh": (-10.0, 10.0), "cosh": (-10.0, 10.0)}
x = D.DeviceArray((N,)); y = D.DeviceArray((N,)); o = D.DeviceArray((N,))
t = Timer()


def run(fn, nbytes):
    for _ in range(2): fn()
    D.sync(); t.start()
    reps = 10
    for _ in range(reps): fn()
    t.stop(); ms = t.elapsed_ms() / reps
    return ms, nbytes / ms / 1e6


print("unary (8 B/elem)")
last = None
for name in UNARY_OPS:
    lo, hi = DOMAIN.get(name, (-100.0, 100.0))
    if (lo, hi) != last:
        x.free()
        x = D.DeviceArray.from_host(synth.uniform((N,), 7, lo, hi))
        last = (lo, hi)
    p0, p1 = (-1.0, 1.0) if name == "clip" else (2.0, 0.0) if name == "round" else (0.0, 0.0)
    ms, gbps = run(lambda: D.unary(name, x, p0, p1, out=o), 8.0 * N)
    print("  %-11s %7.3f ms %7.0f GB/s %5.1f %%" % (name, ms, gbps, gbps / 80.0), flush=True)
print("binary (12 B/elem), both operands full arrays")
x.free(); x = D.DeviceArray.from_host(synth.uniform((N,), 8, 0.5, 4.0))
y.free(); y = D.DeviceArray.from_host(synth.uniform((N,), 9, 0.5, 4.0))
for name in BINARY_OPS:
    quirk = N if name in ("multiply", "mod", "equal", "not_equal") else None
    ms, gbps = run(lambda: D.binary(name, x, "full", y, "full", 1, N, quirk_numel_a=quirk, out=o), 12.0 * N)
    print("  %-13s %7.3f ms %7.0f GB/s %5.1f %%" % (name, ms, gbps, gbps / 80.0), flush=True)
