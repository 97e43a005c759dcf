"""Profiling driver: a few clean launches of the mid-size GEMM tiles (sgemm_dmas_kernel) for rocprofv3 --pmc passes.
Usage: python tools/prof_mid_gemm.py [iters]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from numpower_amd import device as D, synth
from numpower_amd._lib import check, load

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
D.init(0)
lib = load()
for n, shape in ((1024, 2), (1024, 1), (2048, 0), (768, 2)):
    A = D.DeviceArray.from_host(synth.uniform((n, n), 3, -1, 1))
    B = D.DeviceArray.from_host(synth.uniform((n, n), 4, -1, 1))
    Cm = D.DeviceArray((n, n))
    check(lib.np_sgemm_set_variant(-(1000 + 100 * shape + 1)))
    for _ in range(iters):
        D.sgemm(A, B, out=Cm)
    D.sync()
check(lib.np_sgemm_set_variant(-999))
print("done")
