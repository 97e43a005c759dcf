"""np_sgemm 4096^3 with the LDS-DMA kernel under different XCD-aware tile orders (variant = 7 +
10 * GROUP, GROUP = tile rows per band; 7 = plain row-major order): time per launch here, L2->fabric
read traffic from a rocprofv3 --pmc FETCH_SIZE pass over the same script.
Usage: python tools/gemm_swizzle_pmc.py [size]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from numpower_amd import device as D, synth
from numpower_amd._lib import load, Timer, check
D.init(0); lib = load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
A = D.DeviceArray.from_host(synth.uniform((n, n), 3, -1, 1)); B = D.DeviceArray.from_host(synth.uniform((n, n), 4, -1, 1)); Cm = D.DeviceArray((n, n))
for group in (0, 1, 2, 4, 8, 16):
    check(lib.np_sgemm_set_variant(7 + 10 * group))
    for _ in range(2): D.sgemm(A, B, out=Cm)
    D.sync(); t = Timer(); t.start()
    reps = 20
    for _ in range(reps): D.sgemm(A, B, out=Cm)
    t.stop(); ms = t.elapsed_ms() / reps
    print("group %2d  %.3f ms  %.1f TFLOP/s" % (group, ms, 2.0 * n ** 3 / ms / 1e9), flush=True)
check(lib.np_sgemm_set_variant(0))
