"""Why does pow(1e8) time 10-13 % slower when every launch is bracketed by its own event pair than back to back (VERDICT r03
weak #8), when add does not?  The shader clock right behind each form (np_debug_clock_mhz: a 20 us probe kernel on the same
stream) next to the time per launch, for add (HBM-bound) and pow (fp64 VALU next to the HBM bound).
Usage: python tools/pow_clock_probe.py"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from numpower_amd import device as D, synth
from numpower_amd._lib import Timer, check, load

D.init(0)
lib = load()
N = 100_000_000
x = D.DeviceArray.from_host(synth.uniform((N,), 5, 0.0, 1.0))
y = D.DeviceArray.from_host(synth.uniform((N,), 6, 0.0, 1.0))
o = D.DeviceArray((N,))
mhz = C.c_float(0.0)


def clock():
    check(lib.np_debug_clock_mhz(C.byref(mhz)))
    return mhz.value


for rnd in range(3):
    for op in ("add", "pow"):
        fn = lambda: D.binary(op, x, "full", y, "full", 1, N, out=o)
        for _ in range(5):
            fn()
        D.sync()
        t = Timer()
        t.start()
        for _ in range(20):
            fn()
        t.stop()
        back = t.elapsed_ms() / 20 * 1e3
        mhz_back = clock()
        timers = [Timer() for _ in range(20)]
        for tm in timers:
            tm.start()
            fn()
            tm.stop()
        mhz_each = clock()
        each = sorted(tm.elapsed_ms() * 1e3 for tm in timers)
        # the same 20 launches with a clock probe kernel in front of each (what the clock is WHILE the events sit between them)
        print("%-4s back to back %6.1f us/launch (clock behind it %4.0f MHz)   individually timed median %6.1f us, min %6.1f, max %6.1f (clock behind them %4.0f MHz)"
              % (op, back, mhz_back, each[10], each[0], each[-1], mhz_each), flush=True)
