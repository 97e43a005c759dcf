import sys
sys.path.insert(0, '.')
from numpower_amd import device as D, synth
from numpower_amd._lib import Timer
D.init(0)
N = 100_000_000
a = D.DeviceArray.from_host(synth.uniform((N,), 5)); b = D.DeviceArray.from_host(synth.uniform((N,), 6)); o = D.DeviceArray((N,))
out = []
for _ in range(4):
    for _ in range(3): D.binary("add", a, "full", b, "full", 1, N, out=o)
    D.sync(); t = Timer(); t.start()
    for _ in range(25): D.binary("add", a, "full", b, "full", 1, N, out=o)
    t.stop(); out.append(12.0 * N / (t.elapsed_ms() / 25) / 1e6)
print("add GB/s", " ".join("%.0f" % x for x in out), "a=%#x b=%#x o=%#x" % (a.ptr, b.ptr, o.ptr))
