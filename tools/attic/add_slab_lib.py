"""The slab experiment (tools/explore/add_bw.hip slab) through the LIBRARY's add kernel: operands as views of one
1.3 GB np_malloc block at chosen distances, both address orders, vs three separate np_malloc blocks (what the pool
does today).  Run it several times (fresh process = fresh physical placement).  Usage: python tools/add_slab_lib.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from numpower_amd import device as D, synth
from numpower_amd._lib import Timer

D.init(0)
N = 100_000_000
MiB = 1 << 20


def rate(a, b, o):
    for _ in range(3):
        D.binary("add", a, "full", b, "full", 1, N, out=o)
    D.sync()
    out = []
    for _ in range(3):
        t = Timer()
        t.start()
        for _ in range(25):
            D.binary("add", a, "full", b, "full", 1, N, out=o)
        t.stop()
        out.append(12.0 * N / (t.elapsed_ms() / 25) / 1e6)
    return " ".join("%.0f" % x for x in out)


ha, hb = synth.uniform((N,), 5), synth.uniform((N,), 6)
a, b, o = D.DeviceArray.from_host(ha), D.DeviceArray.from_host(hb), D.DeviceArray((N,))
print("separate np_malloc blocks a=%#x b=%#x o=%#x:  %s" % (a.ptr, b.ptr, o.ptr, rate(a, b, o)), flush=True)
print("   same blocks, o and a swapped:              %s" % rate(o, b, a), flush=True)
for d in (a, b, o):
    d.free()
slab = D.DeviceArray((1500 * MiB // 4,))
for sp in (384 * MiB, 384 * MiB + 1024, 392 * MiB + 1024, 400 * MiB + 1024, 512 * MiB):
    x = slab.view(0, (N,))
    y = slab.view(sp // 4, (N,))
    z = slab.view(2 * sp // 4, (N,))
    from numpower_amd._lib import check, load
    check(load().np_memcpy_h2d(x.ptr, ha.ctypes.data, 4 * N))
    check(load().np_memcpy_h2d(y.ptr, hb.ctypes.data, 4 * N))
    print("slab, spacing %3d MiB + %4d B:  a<b<o %s   o<b<a %s" % (sp // MiB, sp % MiB, rate(x, y, z), rate(z, y, x)), flush=True)
