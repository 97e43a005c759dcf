"""For rocprofv3 --kernel-trace --stats: three deep-K products through the default planner, 300 launches each behind a warm-up, so that
the per-kernel averages are those of the K-chunked k-quartered launch and of its fold (profiles/r05/gemm_deep_k_kernel_stats.csv).
Usage: rocprofv3 --kernel-trace --stats -d out -o k --output-format csv -- python tools/prof_deep_k.py"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from numpower_amd import device as D, synth

D.init(0)
for (m, n, k) in ((100, 100, 100000), (128, 128, 65536), (64, 64, 100000)):
    a = D.DeviceArray.from_host(synth.uniform((m, k), 31, -1.0, 1.0))
    b = D.DeviceArray.from_host(synth.uniform((k, n), 32, -1.0, 1.0))
    c = D.DeviceArray((m, n))
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:
        for _ in range(50):
            D.sgemm(a, b, out=c)
        D.sync()
    for _ in range(300):
        D.sgemm(a, b, out=c)
    D.sync()
    print("%d x %d x %d done" % (m, n, k), flush=True)
    for d in (a, b, c):
        d.free()
