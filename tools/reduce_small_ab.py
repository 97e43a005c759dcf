"""nd::sum() of a few hundred KB to a few MB (BASELINE config 1's 1000 x 1000 is 4 MB): the forms np_reduce_all can take, side by
side on one box.  For each n: the call as a host sees it (wall clock per call, 300 calls back to back) and as bench.py's C1 entry
measures it (an event pair around ONE call, median of 50).
  two launches          np_reduce_set_variant(2000000 + 256): first pass + a one-workgroup fold kernel (rounds 2-4)
  grouped tickets       the first pass's last workgroup folds, tickets taken in 32 groups (round 5 default up to 1100 workgroups)
  N fat workgroups      np_reduce_set_variant(3000000 + N): at most N workgroups for n <= 4 M, one ticket
Usage: python tools/reduce_small_ab.py"""
import ctypes as C
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from numpower_amd import _lib, synth
from numpower_amd._lib import Timer, check

lib = _lib.load()
check(lib.np_init(0))
out = C.c_float(0.0)
FORMS = [("two_launches", (2000256, 3000000)), ("grouped_tickets", (2001100, 3000000)), ("grouped_to_4096", (2004096, 3000000)),
         ("fat_256", (2001100, 3000256)), ("fat_512", (2001100, 3000512)), ("fat_128", (2001100, 3000128))]
# warm the clock
big = _lib.DeviceBuffer(4 * 50_000_000)
for _ in range(400):
    check(lib.np_reduce_all(0, big.ptr, 50_000_000, C.byref(out)))
for n in (100_000, 300_000, 1_000_000, 2_000_000, 4_000_000):
    h = synth.uniform((n,), 3, 0.0, 1.0)
    a = _lib.DeviceBuffer(4 * n)
    check(lib.np_memcpy_h2d(a.ptr, h.ctypes.data, 4 * n))
    row = {"n": n}
    vals = {}
    for rnd in range(2):
        for name, variants in FORMS:
            for v in variants:
                check(lib.np_reduce_set_variant(v))
            for _ in range(20):
                check(lib.np_reduce_all(0, a.ptr, n, C.byref(out)))
            t0 = time.perf_counter()
            for _ in range(300):
                check(lib.np_reduce_all(0, a.ptr, n, C.byref(out)))
            wall = (time.perf_counter() - t0) / 300 * 1e6
            timers = [Timer() for _ in range(50)]
            for t in timers:
                t.start()
                check(lib.np_reduce_all(0, a.ptr, n, C.byref(out)))
                t.stop()
            ev = sorted(t.elapsed_ms() for t in timers)[25] * 1e3
            row[name] = [round(min(wall, row.get(name, [1e9])[0]), 2), round(min(ev, row.get(name, [0, 1e9])[1]), 2)]
            vals[name] = out.value
    row["values_equal_two_vs_grouped"] = vals["two_launches"] == vals["grouped_tickets"] == vals["grouped_to_4096"]
    print(json.dumps(row), flush=True)
    a.free()
check(lib.np_reduce_set_variant(2001100))
check(lib.np_reduce_set_variant(3000000))
