"""Write-aligned transposes (output rows off the 128-byte grid) on 128 x 128 against 64 x 64 tiles, same box, alternating:
np_layout_set_variant(7) forces the small tiles.  Usage: python tools/walign_tile_ab.py"""
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
big, out = D.DeviceArray((N,)), D.DeviceArray((N,))
D.fill(big, 1.5)
for _ in range(200):
    check(lib.np_transpose2d(big.ptr, out.ptr, 1, 8191, 8193))
D.sync()
for rows, cols in ((4099, 4099), (2049, 2051), (3001, 2999), (5000, 4099), (8191, 8193), (12345, 6789), (1500, 65535), (65535, 1500), (6001, 6003)):
    best = {}
    for rnd in range(3):
        for v in (0, 7):
            check(lib.np_layout_set_variant(v))
            for _ in range(5):
                check(lib.np_transpose2d(big.ptr, out.ptr, 1, rows, cols))
            D.sync()
            t.start()
            for _ in range(20):
                check(lib.np_transpose2d(big.ptr, out.ptr, 1, rows, cols))
            t.stop()
            ms = t.elapsed_ms() / 20
            best[v] = min(ms, best.get(v, 1e9))
    check(lib.np_layout_set_variant(0))
    tiles = ((cols + 127) // 128) * ((rows + 31 + 127) // 128)
    print("  %6d x %6d  (%5d tiles of 128)   128-tiles %7.3f ms %6.0f GB/s    64-tiles %7.3f ms %6.0f GB/s" % (
        rows, cols, tiles, best[0], 8.0 * rows * cols / best[0] / 1e6, best[7], 8.0 * rows * cols / best[7] / 1e6), flush=True)
# bit-exactness of the small-tile form
for rows, cols in ((4099, 4099), (4100, 5003), (8191, 2051)):
    x = synth.uniform((rows, cols), 5, -1, 1)
    dx, dy = D.DeviceArray.from_host(x), D.DeviceArray((cols, rows))
    check(lib.np_layout_set_variant(7))
    check(lib.np_transpose2d(dx.ptr, dy.ptr, 1, rows, cols))
    check(lib.np_layout_set_variant(0))
    print("  64-tile form bit-exact on %d x %d: %s" % (rows, cols, bool((dy.to_host().view(np.uint32) == np.ascontiguousarray(x.T).view(np.uint32)).all())), flush=True)
