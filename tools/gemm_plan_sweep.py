"""Measure forced np_sgemm plans (NP_SGEMM_PLAN="cfg,tail_rows,S", one subprocess per plan) to
calibrate plan_sgemm's model.  Usage: python tools/gemm_plan_sweep.py"""
import os
os.environ.setdefault("NP_HIP_USE_TUNING_BUILD", "1")   # needs `python -m numpower_amd.build --tuning`
import subprocess
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def one(m, n, k):
    from numpower_amd import device as D
    from numpower_amd._lib import Timer
    D.init(0)
    a = D.DeviceArray((m, k)); b = D.DeviceArray((k, n)); c = D.DeviceArray((m, n))
    D.fill(a, 0.5); D.fill(b, 0.25)
    D.unary("sin", a, out=a); D.unary("cos", b, out=b)
    reps = max(3, min(50, int(1e11 / (2.0 * m * n * k))))
    for _ in range(3): D.sgemm(a, b, out=c)
    D.sync(); t = Timer(); t.start()
    for _ in range(reps): D.sgemm(a, b, out=c)
    t.stop(); ms = t.elapsed_ms() / reps
    print("%5d x %5d x %6d plan %-10s %8.3f ms %6.1f TFLOP/s" % (m, n, k, os.environ.get("NP_SGEMM_PLAN", "model"), ms,
                                                               2.0 * m * n * k / ms / 1e9), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 4:
        one(*[int(v) for v in sys.argv[1:]])
        sys.exit(0)
    cases = {
        (1280, 1280, 8192): ["2,0,1", "1,0,1", "0,0,1", "0,99,2", "0,99,3", "0,99,5", "1,99,2", "1,99,3", "2,99,2"],
        (2304, 2304, 4096): ["2,0,1", "1,0,1", "0,0,1", "0,99,2", "0,99,3", "0,2,4", "0,1,8", "1,99,2"],
        (3072, 3072, 3072): ["2,0,1", "1,0,1", "0,0,1", "0,2,5", "0,2,4", "0,2,6", "0,1,8", "0,3,4", "1,99,2", "0,99,2"],
        (2048, 2048, 2048): ["2,0,1", "1,0,1", "0,0,1", "0,99,2", "0,99,3", "0,99,4", "1,99,2"],
        (1536, 1536, 1536): ["2,0,1", "1,0,1", "0,0,1", "0,99,2", "0,99,3", "0,99,4", "1,99,2", "1,99,3"],
        (1024, 1024, 1024): ["2,0,1", "1,0,1", "0,0,1", "0,99,2", "0,99,4", "0,99,8", "1,99,2", "1,99,4"],
        (100, 100, 100000): ["2,99,128", "2,99,64", "2,99,32", "1,99,128", "1,99,64", "1,99,256"],
        (4096, 4096, 4096): ["0,0,1", "0,8,2"],
    }
    for shape, plans in cases.items():
        for plan in [None] + plans:
            env = dict(os.environ)
            if plan:
                env["NP_SGEMM_PLAN"] = plan
            else:
                env["NP_SGEMM_PLAN_DEBUG"] = "1"
            subprocess.run([sys.executable, __file__] + [str(v) for v in shape], env=env, check=False)
