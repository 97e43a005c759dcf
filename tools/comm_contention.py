"""Does an RCCL transfer issued next to a running GEMM get through?  (the overlapped pipeline of config 5
assumes it does; world > 1 cannot run on a one-GPU lease, the loopback transfer can.)
  alone        `count` self transfers of `bytes` on the communication stream of an idle device
  under 4096^3 the same while a loop of 4096^3 products keeps the library stream busy (512 workgroups = ONE resident round:
               a foreign kernel gets a CU only when a GEMM workgroup retires)
  under slab   ... while a loop of 64 x 1024^3 batched products (config 5's slab per rank: 2048 workgroups, four rounds —
               workgroups retire all the time) keeps it busy
Per transfer: duration from its own event pair on the communication stream (np_comm_debug_loopback_timed).
Usage: python tools/comm_contention.py [MiB per transfer = 32] [count = 8]"""
import ctypes as C
import socket
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from numpower_amd import device as D
from numpower_amd._lib import check, load

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 32
count = int(sys.argv[2]) if len(sys.argv) > 2 else 8
D.init(0)
lib = load()
with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
check(lib.np_comm_init(0, 1, ("tcp://127.0.0.1:%d" % port).encode()))
nbytes = mib << 20
src, dst = D.DeviceArray((nbytes // 4,)), D.DeviceArray((nbytes // 4,))
D.fill(src, 1.25)
ms = (C.c_float * count)()


def transfers():
    check(lib.np_comm_debug_loopback_timed(src.ptr, dst.ptr, nbytes, count, ms))
    return np.array(list(ms))


def report(label, t):
    print("%-14s per transfer ms: %s   median %.3f  max %.3f  (%.0f GB/s at the median)" % (
        label, " ".join("%.3f" % x for x in t), np.median(t), t.max(), nbytes / np.median(t) / 1e6), flush=True)


transfers()
alone = transfers()
report("alone", alone)

n = 4096
A, B, Cm = D.DeviceArray((n, n)), D.DeviceArray((n, n)), D.DeviceArray((n, n))
D.fill(A, 0.5)
D.fill(B, 0.25)
for _ in range(5):
    D.sgemm(A, B, out=Cm)
D.sync()
for rnd in range(3):
    t0 = time.perf_counter()
    for _ in range(40):
        D.sgemm(A, B, out=Cm)          # ~40 ms of queued work
    t_enq = time.perf_counter() - t0
    busy = transfers()
    t_tr = time.perf_counter() - t0
    D.sync()
    t_all = time.perf_counter() - t0
    report("under 4096^3", busy)
    print("               enqueue %.1f ms, transfers back at %.1f ms, GEMM loop done at %.1f ms (%s)" % (
        t_enq * 1e3, t_tr * 1e3, t_all * 1e3, "GEMMs still running when the transfers ended" if t_tr < 0.9 * t_all else "NO OVERLAP WINDOW"), flush=True)

per, m = 64, 1024
bA, bB, bC = D.DeviceArray((per, m, m)), D.DeviceArray((per, m, m)), D.DeviceArray((per, m, m))
D.fill(bA, 0.5)
D.fill(bB, 0.25)


def slab():
    check(lib.np_sgemm_strided_batched(per, m, m, m, bA.ptr, m * m, bB.ptr, m * m, bC.ptr, m * m))


for _ in range(5):
    slab()
D.sync()
for rnd in range(3):
    t0 = time.perf_counter()
    for _ in range(40):
        slab()
    busy = transfers()
    t_tr = time.perf_counter() - t0
    D.sync()
    t_all = time.perf_counter() - t0
    report("under slab", busy)
    print("               transfers back at %.1f ms, GEMM loop done at %.1f ms" % (t_tr * 1e3, t_all * 1e3), flush=True)
check(lib.np_comm_destroy())
