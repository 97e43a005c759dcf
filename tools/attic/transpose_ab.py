"""Dev tool: transpose tile-size A/B (64 vs 128) at a few shapes."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from numpower_amd import device as D, synth
from numpower_amd._lib import load, Timer
D.init(0); lib = load()
for (r, c) in [(65536, 4096), (16384, 4096), (8192, 8192), (4096, 65536), (2048, 2048), (1000, 3000)]:
    x = D.DeviceArray.from_host(synth.uniform((r, c), 1)) ; o = D.DeviceArray((c, r))
    for tile in [int(t) for t in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['64', '128', '64', '128'])]:
        lib.np_layout_set_variant(tile)
        for _ in range(3): D.transpose2d(x, out=o)
        D.sync(); t = Timer(); t.start()
        for _ in range(20): D.transpose2d(x, out=o)
        t.stop(); ms = t.elapsed_ms() / 20
        print("%6d x %6d tile %6d: %.4f ms %6.0f GB/s" % (r, c, tile, ms, 8.0 * r * c / ms / 1e6), flush=True)
        if r * c <= 2048 * 2048 * 4:
            ok = (o.to_host() == x.to_host().T).all(); print("   correct:", bool(ok))
    x.free(); o.free()
