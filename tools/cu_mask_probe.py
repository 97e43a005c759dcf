"""The CU-mask bit <-> (XCD, SE, CU) mapping of hipExtStreamCreateWithCUMask on this box, read off the hardware, and what a BALANCED
mask (the same number of CUs taken from every XCD) buys the communication kernels next to a running GEMM (VERDICT r04 next #8b;
DESIGN.md section 7).
  1. streams with single bits, then with the eight bits 8 j .. 8 j + 7 set, 256 one-wave workgroups on each reporting HW_REG_HW_ID /
     HW_REG_XCC_ID (np_debug_hw_ids) -> bit b belongs to XCC b % 8, slot b // 8; an XCC with no bit set is NOT restricted
  2. masks built from that map: k CUs removed from EVERY XCD (k = 1, 2, 4), against round 4's naive masks; for each, the slab GEMM of
     config 5 and a 4096^3 product on the masked stream, 32 MiB RCCL self-transfers issued next to them on the communication stream
Keep-criterion (VERDICT): opt-in only if the slab GEMM loses <= 3 % and the transfer delay drops >= 3x.
Usage: python tools/cu_mask_probe.py"""
import collections
import ctypes as C
import socket
import sys
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
hip.hipStreamDestroy.argtypes = [C.c_void_p]


def masked_stream(on_bits):
    words = (C.c_uint32 * 8)(*[sum(1 << b for b in range(32) if (32 * w + b) in on_bits) for w in range(8)])
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
    return st


def decode(hw_id, xcc):
    # gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]; XCC_ID: xcc_id[3:0]
    return (xcc & 0xF, (hw_id >> 13) & 0x7, (hw_id >> 12) & 0x1, (hw_id >> 8) & 0xF)


WG = 64
buf = (C.c_uint * (2 * WG))()


def where(on_bits, wgs=WG):
    st = masked_stream(on_bits)
    check(lib.np_set_stream(st))
    out = (C.c_uint * (2 * wgs))()
    check(lib.np_debug_hw_ids(out, wgs))
    check(lib.np_set_stream(None))
    hip.hipStreamDestroy(st)
    return sorted({decode(out[2 * w], out[2 * w + 1]) for w in range(wgs)})


all_buf = (C.c_uint * (2 * 4096))()
check(lib.np_debug_hw_ids(all_buf, 4096))
everything = sorted({decode(all_buf[2 * w], all_buf[2 * w + 1]) for w in range(4096)})
print("unmasked stream: %d distinct (xcc, se, sh, cu); per xcc: %s" % (
    len(everything), dict(collections.Counter(c[0] for c in everything))), flush=True)
# step 1: single bits.  (First finding, lease 6 of round 5: a mask with ONE bit set restricts ONE XCC to one CU and leaves the seven
# XCCs whose share of the mask is empty unrestricted: bit b belongs to XCC b % 8.)
for bit in (0, 1, 7, 8, 9, 255):
    per = collections.Counter(c[0] for c in where({bit}, 256))
    print("only bit %3d set: CUs seen per xcc %s" % (bit, dict(sorted(per.items()))), flush=True)
# step 2: slot j = the eight bits 8 j .. 8 j + 7 (one per XCC): which (se, sh, cu) is slot j in each XCC?
slot_cus = {}
for j in range(32):
    slot_cus[j] = where(set(range(8 * j, 8 * j + 8)), 256)
ok = all(len(v) == 8 and sorted(c[0] for c in v) == list(range(8)) for v in slot_cus.values())
print("every slot (bits 8 j .. 8 j + 7) enables exactly one CU in each of the 8 XCCs: %s" % ok, flush=True)
print("slot -> (se, sh, cu) in xcc 0: %s" % {j: [c[1:] for c in v if c[0] == 0] for j, v in slot_cus.items()}, flush=True)
same = all(len({c[1:] for c in v}) == 1 for v in slot_cus.values())
print("a slot is the same (se, sh, cu) in every xcc: %s" % same, flush=True)
covered = {c for v in slot_cus.values() for c in v}
print("CUs covered by the 32 slots: %d of %d" % (len(covered), len(everything)), flush=True)

# ---- part 2: balanced masks next to a GEMM ----
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
print("transfer alone: median %.3f ms" % np.median(transfers()), flush=True)
MASKS = [("none", None)]
for k in (1, 2, 4):
    off = set(range(256 - 8 * k, 256))                # the k highest slots of every XCC (k = 1 is round 4's "top 8")
    MASKS.append(("balanced %d/xcc (%d off)" % (k, len(off)), off))
MASKS += [("xcc 7's whole share (round 4's \"every 8th bit\")", set(range(7, 256, 8))), ("8 CUs of xcc 0 only", set(range(0, 64, 8)))]
# each balanced mask twice: with the planners left at 256 CUs, and told the CU count the masked stream really has (np_debug_set_cus)
MASKS = [m + (False,) for m in MASKS] + [(name + ", planned for %d CUs" % (256 - len(off)), off, True) for name, off in MASKS[1:4]]
for name, off, replan in MASKS:
    st = None
    if off is not None:
        st = masked_stream(set(range(256)) - off)
        check(lib.np_set_stream(st))
        if replan:
            check(lib.np_debug_set_cus(256 - len(off)))
    line = "%-44s" % name
    for label, fn in (("slab", slab), ("4096^3", big)):
        for _ in range(30):
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
    if st is not None:
        check(lib.np_debug_set_cus(0))
        check(lib.np_set_stream(None))
        hip.hipStreamDestroy(st)
check(lib.np_comm_destroy())
