"""Dev tool: what does cutting one rank's slab of config 5 (64 x 1024^2) into pieces cost, step by step?

    A  GEMM launches only (chunks x np_sgemm_strided_batched back to back)
    B  + hipEventRecord on the library stream after each piece
    C  + hipStreamWaitEvent of the communication stream on that event
    D  the real pipeline, ncclAllGather transport (np_allgather_async mode 1, in place: one-rank no-op)
    E  the real pipeline, p2p transport (mode 2, in place: nothing to send)
    F  np_sgemm_strided_batched_allgather(chunks), HIP events + one launch per piece   (np_comm_set_variant(1))
    G  ... device-side flags + one launch per piece                                       (np_comm_set_variant(2))
    H  ... device-side flags + ONE progress-reporting GEMM launch (the default)           (np_comm_set_variant(0))
    (A-E run under variant 1, i.e. what the event form costs piece by piece)
Each after 40 warm-up launches (past the power ramp), 20 repetitions, event-timed on the library stream with a
np_comm_wait at the end of each repetition.

    python tools/chunk_overhead.py"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from numpower_amd import device as D                      # noqa: E402
from numpower_amd._lib import Timer, check, load          # noqa: E402

D.init(0)
lib = load()
hip = C.CDLL("libamdhip64.so")
check(lib.np_comm_init(0, 1, b"/tmp/np_chunk_overhead.id"))
print("sync mode after init (1 = device-side flags passed the self-test):", lib.np_comm_sync_mode())
n, slab = 1024, 64
mat = n * n
A, B, full = D.DeviceArray((slab, n, n)), D.DeviceArray((slab, n, n)), D.DeviceArray((slab, n, n))
D.fill(A, 0.5)
D.fill(B, 0.25)
D.unary("sin", A, out=A)
D.unary("cos", B, out=B)
stream, comm = C.c_void_p(lib.np_get_stream()), C.c_void_p(lib.np_comm_stream())
events = []
for _ in range(16):
    e = C.c_void_p()
    assert hip.hipEventCreateWithFlags(C.byref(e), 2) == 0      # hipEventDisableTiming
    events.append(e)


def gemm(lo, cnt):
    check(lib.np_sgemm_strided_batched(cnt, n, n, n, A.ptr + lo * mat * 4, mat, B.ptr + lo * mat * 4, mat, full.ptr + lo * mat * 4, mat))


def variant(kind, chunks):
    piece = slab // chunks

    def run():
        for c in range(chunks):
            lo = c * piece
            gemm(lo, piece)
            if kind in "BC":
                assert hip.hipEventRecord(events[c], stream) == 0
            if kind == "C":
                assert hip.hipStreamWaitEvent(comm, events[c], 0) == 0
            if kind in "DE":
                check(lib.np_allgather_async(full.ptr + lo * mat * 4, full.ptr + lo * mat * 4, piece * mat * 4,
                                             piece * mat * 4 if kind == "D" else slab * mat * 4, 1 if kind == "D" else 2))
        if kind in "DE":
            check(lib.np_comm_wait())
    if kind in "FGH":
        def whole():
            check(lib.np_comm_set_variant({"F": 1, "G": 2, "H": 0}[kind]))
            check(lib.np_sgemm_strided_batched_allgather(slab, n, n, n, A.ptr, mat, B.ptr, mat, full.ptr, chunks, 0))
        return whole

    def with_events():
        check(lib.np_comm_set_variant(1))
        run()
    return with_events


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    D.sync()
    t = Timer()
    t.start()
    for _ in range(reps):
        fn()
    t.stop()
    return t.elapsed_ms() / reps


for _ in range(40):
    gemm(0, slab)
D.sync()
print("one launch of 64: %.4f ms" % timed(lambda: gemm(0, slab)))
for chunks in (1, 2, 4, 8, 16):
    line = "chunks %2d:" % chunks
    for kind in "ABCDEFGH":
        line += "  %s %.4f" % (kind, timed(variant(kind, chunks)))
    print(line, flush=True)
check(lib.np_comm_set_variant(0))
print("one launch of 64 again: %.4f ms" % timed(lambda: gemm(0, slab)))
big = D.DeviceArray((4096, 4096)); bigb = D.DeviceArray((4096, 4096)); bigc = D.DeviceArray((4096, 4096))
D.fill(big, 0.5); D.fill(bigb, 0.25); D.unary("sin", big, out=big); D.unary("cos", bigb, out=bigb)
for _ in range(30):
    D.sgemm(big, bigb, out=bigc)
print("4096^3 (headline shape, progress off): %.4f ms" % timed(lambda: D.sgemm(big, bigb, out=bigc)))
check(lib.np_comm_destroy())
