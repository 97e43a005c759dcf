"""A/B harness (dev tool): interleaved rounds over sgemm variants at 4096^3 (+ correctness)."""
import os
os.environ.setdefault("NP_HIP_USE_TUNING_BUILD", "1")   # needs `python -m numpower_amd.build --tuning`
import sys, json
os.environ.setdefault("NP_ALLOW_ABLATION", "1")   # variants >= 1000 time parts of the kernel by switching them off
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from numpower_amd import device as D, synth
from numpower_amd._lib import load, Timer
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [3, 5, 6]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
D.init(0); lib = load()
A = synth.uniform((n, n), 3, -1, 1); B = synth.uniform((n, n), 4, -1, 1)
dA, dB, dC = D.DeviceArray.from_host(A), D.DeviceArray.from_host(B), D.DeviceArray((n, n))
rows = [0, 17, n // 2 + 3, n - 1]
ref = A[rows].astype(np.float64) @ B.astype(np.float64)
scale = np.abs(A[rows]).astype(np.float64) @ np.abs(B).astype(np.float64)
for v in variants:
    lib.np_sgemm_set_variant(v); D.fill(dC, 0.0); D.sgemm(dA, dB, out=dC)
    got = dC.to_host()[rows].astype(np.float64)
    print("variant", v, "max norm err", float((np.abs(got - ref) / scale).max()), flush=True)
# odd shapes through the pipelined kernels
for (m, nn, k) in [(257, 129, 65), (100, 90, 70), (1000, 1000, 1000), (130, 260, 16), (128, 128, 8), (5, 3, 2)]:
    a = synth.uniform((m, k), 1, -1, 1); b = synth.uniform((k, nn), 2, -1, 1)
    r64 = a.astype(np.float64) @ b.astype(np.float64); sc = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    for v in variants:
        lib.np_sgemm_set_variant(v)
        got = D.sgemm(D.DeviceArray.from_host(a), D.DeviceArray.from_host(b)).to_host()
        e = float((np.abs(got - r64) / sc).max())
        print("  shape", (m, nn, k), "variant", v, "err", e, "OK" if e < 1e-6 else "FAIL", flush=True)
res = {v: [] for v in variants}
for r in range(rounds):
    for v in variants:
        lib.np_sgemm_set_variant(v)
        for _ in range(3): D.sgemm(dA, dB, out=dC)
        D.sync(); t = Timer(); t.start()
        for _ in range(20): D.sgemm(dA, dB, out=dC)
        t.stop(); ms = t.elapsed_ms() / 20
        res[v].append(2 * n**3 / ms / 1e9)
for v in variants:
    print("variant %3d  TF: %s  median %.1f" % (v, " ".join("%.1f" % x for x in res[v]), float(np.median(res[v]))), flush=True)
