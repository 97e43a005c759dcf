"""Placement copies through the C ABI (what NDArray_ToGPU / NDArray_ToCPU cost): pageable host
buffers of 16 MB / 400 MB / 2 GB, H2D and D2H (into an already-touched and into a fresh
destination).  Usage: python tools/pcie_ab.py"""
import ctypes as C
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))


def one():
    import numpy as np
    from numpower_amd import _lib
    lib = _lib.load()
    _lib.check(lib.np_init(0))
    for nbytes in (16 << 20, 400_000_000, 2_000_000_000):
        src = np.ones(nbytes // 4, np.float32)
        dev = _lib.DeviceBuffer(nbytes)
        for _ in range(2):
            _lib.check(lib.np_memcpy_h2d(dev.ptr, src.ctypes.data, nbytes))
        t0 = time.perf_counter(); reps = 5
        for _ in range(reps):
            _lib.check(lib.np_memcpy_h2d(dev.ptr, src.ctypes.data, nbytes))
        h2d = nbytes * reps / (time.perf_counter() - t0) / 1e9
        dst = np.zeros(nbytes // 4, np.float32)
        _lib.check(lib.np_memcpy_d2h(dst.ctypes.data, dev.ptr, nbytes))
        t0 = time.perf_counter()
        for _ in range(reps):
            _lib.check(lib.np_memcpy_d2h(dst.ctypes.data, dev.ptr, nbytes))
        d2h = nbytes * reps / (time.perf_counter() - t0) / 1e9
        ok = bool((dst[:: max(1, dst.size // 1000)] == 1.0).all() and dst[-1] == 1.0)
        fresh = []
        for _ in range(3):
            d2 = np.empty(nbytes // 4, np.float32)          # untouched pages
            t0 = time.perf_counter()
            _lib.check(lib.np_memcpy_d2h(d2.ctypes.data, dev.ptr, nbytes))
            fresh.append(nbytes / (time.perf_counter() - t0) / 1e9)
            ok = ok and bool(d2[-1] == 1.0 and d2[0] == 1.0)
            del d2
        print("%5d MB  H2D %5.1f GB/s   D2H %5.1f GB/s   D2H into fresh pages %5.1f GB/s  ok=%s"
              % (nbytes >> 20, h2d, d2h, max(fresh), ok), flush=True)
        dev.free()


if __name__ == "__main__":
    one()
