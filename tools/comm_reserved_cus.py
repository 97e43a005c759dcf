"""Experiment behind DESIGN.md section 7's "next step": does keeping a few CUs away from the GEMM (a CU-masked stream,
hipExtStreamCreateWithCUMask, installed as the library stream through np_set_stream) let an RCCL transfer start at once instead
of waiting for GEMM workgroups to retire?  For several masks: per-transfer time of 32 MiB self-transfers issued next to
(a) a queue of config 5's slab GEMMs and (b) a queue of 4096^3 products, and what the GEMM itself loses to the missing CUs.
Masks: 'none' (the plain stream), 'top n' (CUs 256 - n .. 255 off), 'spread n' (every (256 / n)-th CU off).
Usage: python tools/comm_reserved_cus.py"""
import ctypes as C
import socket
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from numpower_amd import device as D
from numpower_amd._lib import Timer, check, load

D.init(0)
lib = load()
hip = C.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(C.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = C.c_int
with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
check(lib.np_comm_init(0, 1, ("tcp://127.0.0.1:%d" % port).encode()))
nbytes, count = 32 << 20, 8
src, dst = D.DeviceArray((nbytes // 4,)), D.DeviceArray((nbytes // 4,))
D.fill(src, 1.25)
ms = (C.c_float * count)()
t = Timer()


def transfers():
    check(lib.np_comm_debug_loopback_timed(src.ptr, dst.ptr, nbytes, count, ms))
    return np.array(list(ms))


def masked_stream(off):
    bits = [1] * 256
    for i in off:
        bits[i] = 0
    words = (C.c_uint32 * 8)(*[sum(bits[32 * w + b] << b for b in range(32)) for w in range(8)])
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
    return st


per, m, n = 64, 1024, 4096
bA, bB, bC = D.DeviceArray((per, m, m)), D.DeviceArray((per, m, m)), D.DeviceArray((per, m, m))
A, B, Cm = D.DeviceArray((n, n)), D.DeviceArray((n, n)), D.DeviceArray((n, n))
for d in (bA, A):
    D.fill(d, 0.5)
for d in (bB, B):
    D.fill(d, 0.25)


def slab():
    check(lib.np_sgemm_strided_batched(per, m, m, m, bA.ptr, m * m, bB.ptr, m * m, bC.ptr, m * m))


def big():
    D.sgemm(A, B, out=Cm)


transfers()
print("alone: median %.3f ms" % np.median(transfers()), flush=True)
MASKS = [("none", None), ("top 8", range(248, 256)), ("top 16", range(240, 256)), ("spread 8", range(31, 256, 32)),
         ("spread 16", range(15, 256, 16)), ("spread 32", range(7, 256, 8))]
for name, off in MASKS:
    if off is not None:
        st = masked_stream(list(off))
        check(lib.np_set_stream(st))
    line = "%-10s" % name
    for label, fn in (("slab", slab), ("4096^3", big)):
        for _ in range(5):
            fn()
        D.sync()
        t.start()
        for _ in range(20):
            fn()
        t.stop()
        gemm_ms = t.elapsed_ms() / 20
        worst, med = 0.0, []
        for rnd in range(3):
            for _ in range(40):
                fn()
            tr = transfers()
            D.sync()
            worst = max(worst, float(tr.max()))
            med.append(float(np.median(tr)))
        line += "   %s: GEMM %.3f ms, transfer median %.3f max %.3f ms" % (label, gemm_ms, float(np.median(med)), worst)
    print(line, flush=True)
    if off is not None:
        check(lib.np_set_stream(None))
check(lib.np_comm_destroy())
