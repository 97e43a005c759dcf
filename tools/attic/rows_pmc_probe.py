"""Dev tool (run under rocprofv3 --pmc): sum(exp(X), axis 1) over well-packed mid-length rows — the lane-group kernel's
regime — a few clean launches per shape.  Usage: python tools/rows_pmc_probe.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from numpower_amd import _lib, synth
from numpower_amd.lazy import Lazy   # noqa: F401
from numpower_amd.ndarray import NDArray

lib = _lib.load()
_lib.check(lib.np_init(0))
for rows, cols in ((1_600_000, 64), (400_000, 256), (98_000, 1024), (25_000, 4000)):
    gx = NDArray.array(synth.uniform((rows, cols), 3, -1.0, 1.0)).gpu()
    for _ in range(5):
        y = gx.lazy().exp().sum(axis=1)
    _lib.check(lib.np_sync())
    del gx, y
