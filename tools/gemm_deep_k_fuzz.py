"""Random deep-K products through the DEFAULT planner (few tiles, K 2048 .. 300000: the K-chunked k-quartered plans, the thin K-chunk
kernels, the older chunked forms — whatever the planner picks), each against fp64 (1e-6 of sum |a||b|) and run twice (bit-identical).
Usage: python tools/gemm_deep_k_fuzz.py [cases] [seed]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from numpower_amd import device as D
from numpower_amd._lib import check, load

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
D.init(0)
lib = load()
out = (C.c_double * 11)()
bad, chunked = 0, 0
for i in range(cases):
    m, n = (int(10 ** rng.uniform(0.3, 2.9)) for _ in range(2))
    k = int(10 ** rng.uniform(3.3, 5.5))
    if rng.random() < 0.5:
        k = k // 4 * 4
        n = max(4, n // 4 * 4)
    A = rng.uniform(-1, 1, (m, k)).astype(np.float32)
    B = rng.uniform(-1, 1, (k, n)).astype(np.float32)
    a, b, c = D.DeviceArray.from_host(A), D.DeviceArray.from_host(B), D.DeviceArray((m, n))
    check(lib.np_sgemm_debug_plan(m, n, k, 1, 0, out))
    chunked += out[0] >= 6 and out[1] > 0
    runs = []
    for _ in range(2):
        D.fill(c, float("nan"))
        D.sgemm(a, b, out=c)
        runs.append(c.to_host().copy())
    want = A.astype(np.float64) @ B.astype(np.float64)
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    ok = bool((np.abs(runs[0] - want) <= 1e-6 * scale).all()) and bool((runs[0].view(np.uint32) == runs[1].view(np.uint32)).all())
    if not ok:
        bad += 1
        print("MISMATCH %d x %d x %d plan %s" % (m, n, k, list(out)[:4]), flush=True)
    for d in (a, b, c):
        d.free()
print("%d cases (%d on the K-chunked k-quartered plan), %d mismatches, np_sync rc %d" % (cases, chunked, bad, lib.np_sync()))
