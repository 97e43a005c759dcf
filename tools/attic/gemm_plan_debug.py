"""Dev tool (tuning build only: NP_HIP_USE_TUNING_BUILD=1 NP_SGEMM_PLAN_DEBUG=1): print the planner's choice for a list
of shapes — one product each, the trace goes to stderr.  Usage: python tools/gemm_plan_debug.py 1001x1001x1001 ..."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from numpower_amd import device as D
D.init(0)
for arg in sys.argv[1:]:
    m, n, k = (int(v) for v in arg.split("x"))
    a = D.DeviceArray((m, k)); b = D.DeviceArray((k, n))
    D.fill(a, 0.5); D.fill(b, 0.25)
    c = D.sgemm(a, b)
    D.sync()
    for x in (a, b, c):
        x.free()
