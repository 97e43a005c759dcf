"""Write-aligned transposes (output rows off the 128-byte grid): the tile order — diagonal walk in launch order (0), the same walk
dealt to the XCDs in eight contiguous runs (1), column strips dealt to the XCDs in runs (2); np_layout_set_variant(17000 + k).
Same box, alternating.  Usage: python tools/walign_order_ab.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from numpower_amd import device as D, synth
from numpower_amd._lib import Timer, check, load

D.init(0)
lib = load()
t = Timer()
N = 100_000_000
src, dst = D.DeviceArray.from_host(synth.uniform((N,), 3, -1, 1)), D.DeviceArray((N,))
host = src.to_host()
for _ in range(200):
    check(lib.np_transpose2d(src.ptr, dst.ptr, 1, 8192, 8192))
D.sync()
for rows, cols in ((8191, 8193), (8196, 8192), (8190, 8194), (8193, 8191), (8000, 8001), (4096, 4099), (8192, 10001), (16000, 5001), (3200, 30001), (10001, 9999), (12345, 6789), (6001, 6003), (4099, 4099), (1500, 65536), (65536, 1500), (3000, 30001)):
    best = {}
    for rnd in range(6):
        for order in (0, 1, 2):
            check(lib.np_layout_set_variant(17000 + order))
            check(lib.np_layout_set_variant(17010 + min(order, 1)))     # (the plain tile kernel knows orders 0 and 1)
            for _ in range(3):
                check(lib.np_transpose2d(src.ptr, dst.ptr, 1, rows, cols))
            D.sync()
            t.start()
            for _ in range(20):
                check(lib.np_transpose2d(src.ptr, dst.ptr, 1, rows, cols))
            t.stop()
            best[order] = min(best.get(order, 1e9), t.elapsed_ms() / 20)
            if rnd == 0:
                got = dst.to_host()[:rows * cols].reshape(cols, rows)
                assert (got == host[:rows * cols].reshape(rows, cols).T).all(), (rows, cols, order)
    check(lib.np_layout_set_variant(17003))          # back to the default rule
    check(lib.np_layout_set_variant(17013))
    print("  %6d x %-6d " % (rows, cols) + "  ".join("order %d %5.0f GB/s" % (o, 8.0 * rows * cols / best[o] / 1e6) for o in (0, 1, 2)), flush=True)
