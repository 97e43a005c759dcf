"""Order statistics: the bracket path (sample -> bracket -> one filtering pass -> three passes over the
copied keys) against the plain three passes, same process, same buffer.  Usage: python tools/select_ab.py"""
import ctypes as C
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np

from numpower_amd import _lib, synth

lib = _lib.load()
_lib.check(lib.np_init(0))
import os
n = int(os.environ.get("SELECT_AB_N", "100000000"))
cases = {
    "uniform01": lambda: synth.uniform((n,), 5, 0.0, 1.0),
    "signed_wide": lambda: (synth.uniform((n,), 6, -1.0, 1.0) * np.exp(synth.uniform((n,), 7, -8.0, 8.0))).astype(np.float32),
    "relu": lambda: np.maximum(synth.uniform((n,), 8, -1.0, 1.0), np.float32(0)),
    "sorted": lambda: np.sort(synth.uniform((n,), 9, 0.0, 1.0)),
}
import os
if os.environ.get("SELECT_AB_CASES"):
    cases = {k: v for k, v in cases.items() if k in os.environ["SELECT_AB_CASES"].split(",")}
variants = [(int(v), "variant%s_ms" % v) for v in os.environ.get("SELECT_AB_VARIANTS", "").split(",") if v] or [(0, "plain_ms"), (1, "bracket_ms")]
buf = _lib.DeviceBuffer(4 * n)
out2 = _lib.DeviceBuffer(8)
for name, make in cases.items():
    h = make()
    _lib.check(lib.np_memcpy_h2d(buf.ptr, h.ctypes.data, 4 * n))
    for k_name, k in (("median", n // 2), ("p99", n * 99 // 100), ("min", 0)):
        row = {"n": n, "data": name, "rank": k_name}
        for variant, label in variants:
            _lib.check(lib.np_select_set_variant(1))
            _lib.check(lib.np_select_set_variant(variant if variant != 1 else 2048))
            for _ in range(3):
                _lib.check(lib.np_order_stat_dev(buf.ptr, n, k, out2.ptr))
            t = _lib.Timer()
            t.start()
            for _ in range(20):
                _lib.check(lib.np_order_stat_dev(buf.ptr, n, k, out2.ptr))
            t.stop()
            _lib.check(lib.np_sync())
            row[label] = round(t.elapsed_ms() / 20, 4)
        print(json.dumps(row), flush=True)
