"""Dev tool: per-workgroup shader-clock / wall-clock records of the sgemm kernels, to tell
"cycles lost to stalls" from "clock lowered by DVFS".  Usage: python tools/gemm_probe.py 3,6"""
import sys, ctypes as C
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from numpower_amd import device as D, synth
from numpower_amd._lib import load, check, Timer
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [3, 6]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
D.init(0); lib = load()
lib.np_debug_sgemm_probe.argtypes = [C.c_void_p]
A = synth.uniform((n, n), 3, -1, 1); B = synth.uniform((n, n), 4, -1, 1)
dA, dB, dC = D.DeviceArray.from_host(A), D.DeviceArray.from_host(B), D.DeviceArray((n, n))
nblk = (n // 128) ** 2   # upper bound; the 256x128 DMA kernel uses half of the records
probe = D.DeviceArray((nblk * 16,))   # 8 x u64 per block
ideal_cycles_per_wave = (n // 16) * 32 * 64   # K-tiles x MFMAs x 64 cycles
for v in variants:
    lib.np_sgemm_set_variant(v)
    for _ in range(10): D.sgemm(dA, dB, out=dC)   # warm, steady clocks
    D.sync()
    check(lib.np_memset0(probe.ptr, nblk * 64))
    lib.np_debug_sgemm_probe(probe.ptr)
    t = Timer(); t.start(); D.sgemm(dA, dB, out=dC); t.stop(); ms = t.elapsed_ms()
    lib.np_debug_sgemm_probe(None)
    raw = probe.to_host().view(np.uint64).reshape(nblk, 8)
    raw = raw[raw[:, 1] != 0]   # records actually written
    c = (raw[:, 1] - raw[:, 0]).astype(np.float64); w = (raw[:, 3] - raw[:, 2]).astype(np.float64) / 100e6
    span = (raw[:, 3].max() - raw[:, 2].min()) / 100e6
    clk = c / w
    print("variant %d: event %.4f ms (%.1f TF) | span of all WGs %.4f ms | per-WG wall %.4f..%.4f ms (mean %.4f)"
          % (v, ms, 2 * n**3 / ms / 1e9, span * 1e3, w.min() * 1e3, w.max() * 1e3, w.mean() * 1e3))
    print("   per-WG cycles mean %.0f (min %.0f max %.0f); effective shader clock %.3f GHz (min %.3f max %.3f)"
          % (c.mean(), c.min(), c.max(), clk.mean() / 1e9, clk.min() / 1e9, clk.max() / 1e9))
    start = (raw[:, 2] - raw[:, 2].min()).astype(np.float64) / 100e6 * 1e6
    print("   WG start skew: max %.1f us ; distinct start clusters (>50us apart): %d" % (start.max(), int((np.diff(np.sort(start)) > 50).sum()) + 1))
    res_per_cu = 4 if v in (1, 3) else 2
    if v in (0, 7): ideal_cycles_per_wave_v = (n // 16) * 64 * 64
    else: ideal_cycles_per_wave_v = ideal_cycles_per_wave
    print("   ideal MFMA cycles per wave %d x %d co-resident waves/SIMD = %d ; WG cycles / that = %.3f"
          % (ideal_cycles_per_wave_v, res_per_cu, ideal_cycles_per_wave_v * res_per_cu, c.mean() / (ideal_cycles_per_wave_v * res_per_cu)))
    for x in range(8):
        m = raw[:, 4] == x
        if m.any(): print("   xcc %d: %4d WGs, mean wall %.4f ms, mean clk %.3f GHz" % (x, int(m.sum()), w[m].mean() * 1e3, clk[m].mean() / 1e9))
