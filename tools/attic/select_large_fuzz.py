"""np_order_stat on arrays large enough to take the bracket path by default (n >= 2^26): random length, distribution and
ranks, against a host sort of the keys.  Usage: python tools/select_large_fuzz.py [cases] [seed]"""
import ctypes as C
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np

from numpower_amd import _lib, synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
lib = _lib.load()
_lib.check(lib.np_init(0))
rng = np.random.default_rng(1000 + seed)


def keys(x):
    u = x.view(np.uint32)
    return np.where(u & 0x80000000, ~u, u | 0x80000000).astype(np.uint32)


def from_keys(k):
    k = np.asarray(k, np.uint32)
    return np.where(k & 0x80000000, k & 0x7fffffff, ~k).astype(np.uint32).view(np.float32)


taken = checked = 0
for case in range(cases):
    n = int(rng.integers(1 << 26, 130_000_000))
    u = synth.uniform((n,), 5000 + case + 100 * seed, 0.0, 1.0)
    style = case % 6
    if style == 0:
        x = u
    elif style == 1:
        x = ((u - np.float32(0.5)) * np.exp(synth.uniform((n,), 6000 + case, -30.0, 30.0))).astype(np.float32)
    elif style == 2:
        x = np.where(u < rng.uniform(0.05, 0.6), np.float32(rng.uniform(-1, 1)), u - np.float32(0.5)).astype(np.float32)
    elif style == 3:
        x = np.sort(u)[::(1 if case % 12 < 6 else -1)].copy()
    elif style == 4:
        x = np.rint(u * np.float32(rng.integers(2, 200))).astype(np.float32)
    else:
        x = (u * np.float32(1e-3) + np.float32(rng.uniform(0.5, 4.0))).astype(np.float32)
    x = np.ascontiguousarray(x, np.float32)
    t0 = time.perf_counter()
    want = np.sort(keys(x))
    buf = _lib.DeviceBuffer(4 * n)
    _lib.check(lib.np_memcpy_h2d(buf.ptr, x.ctypes.data, 4 * n))
    out = (C.c_float * 2)()
    path = C.c_int(-1)
    paths = []
    for k in sorted({0, n - 1, n // 2, int(rng.integers(0, n)), int(rng.integers(0, n)), int(rng.integers(0, 5000)), n - 1 - int(rng.integers(0, 5000))}):
        _lib.check(lib.np_order_stat(buf.ptr, n, k, out))
        exp = from_keys([want[k], want[min(k + 1, n - 1)]]).view(np.uint32).tolist()
        got = np.float32([out[0], out[1]]).view(np.uint32).tolist()
        assert got == exp, (case, style, n, k, got, exp)
        _lib.check(lib.np_select_last_path(C.byref(path)))
        paths.append(path.value)
        checked += 1
    taken += sum(paths)
    buf.free()
    print("case %2d style %d n %9d paths %s  (%.1f s)" % (case, style, n, paths, time.perf_counter() - t0), flush=True)
print("ok: %d selections checked, %d on the bracket path" % (checked, taken))
