"""Profiling driver: a few clean launches of the small / mid GEMM kernels at the sizes they are for (768^3 on sgemm_kq_kernel's
48 x 48 tiles, 512^3 on its 32 x 32 tiles, 1024^3 on sgemm_dmas_kernel's 64 x 64 tiles, 4096^3 on sgemm_dma_kernel) for
rocprofv3 --pmc passes (L2 hits / misses, requests per kernel).
Usage: python tools/prof_small_gemm.py [iters]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from numpower_amd import device as D, synth
from numpower_amd._lib import check, load

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
D.init(0)
lib = load()
for n in (768, 512, 1024, 4096):
    A = D.DeviceArray.from_host(synth.uniform((n, n), 3, -1, 1))
    B = D.DeviceArray.from_host(synth.uniform((n, n), 4, -1, 1))
    Cm = D.DeviceArray((n, n))
    for _ in range(iters + 3):
        D.sgemm(A, B, out=Cm)
    D.sync()
print("done")
