"""Dev tool (run under `rocprofv3 --kernel-trace`): the two-stream pipeline of np_comm.hip on one GPU.

A one-rank communicator has no peer, so an in-place gather moves nothing; the OUT-OF-PLACE form does: the rank's own
piece is copied into the result on the communication stream (RCCL's one-rank all-gather / the p2p path's own-piece
copy).  The loop below is config 5's pipeline with that copy standing in for the xGMI transfer:
    GEMM(piece c) on the library stream  ->  event  ->  copy(piece c) on the communication stream  ||  GEMM(piece c+1)
tools/overlap_trace_report.py reads the kernel trace and reports how much of the copies' time ran INSIDE a GEMM's
[start, end] interval on another stream.

    python tools/overlap_trace.py [reps] [chunks]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from numpower_amd import device as D                      # noqa: E402
from numpower_amd._lib import Timer, check, load          # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
D.init(0)
lib = load()
check(lib.np_comm_init(0, 1, b"/tmp/np_overlap_trace.id"))
n, slab = 1024, 64
mat = n * n
A, B = D.DeviceArray((slab, n, n)), D.DeviceArray((slab, n, n))
D.fill(A, 0.5)
D.fill(B, 0.25)
D.unary("sin", A, out=A)
D.unary("cos", B, out=B)
work, full = D.DeviceArray((slab, n, n)), D.DeviceArray((slab, n, n))
piece = slab // chunks


def serial():
    check(lib.np_sgemm_strided_batched(slab, n, n, n, A.ptr, mat, B.ptr, mat, work.ptr, mat))
    check(lib.np_allgather(work.ptr, full.ptr, slab * mat * 4))          # library stream: behind the GEMM


def pipelined(mode):
    for c in range(chunks):
        lo = c * piece
        check(lib.np_sgemm_strided_batched(piece, n, n, n, A.ptr + lo * mat * 4, mat, B.ptr + lo * mat * 4, mat,
                                           work.ptr + lo * mat * 4, mat))
        check(lib.np_allgather_async(work.ptr + lo * mat * 4, full.ptr + lo * mat * 4, piece * mat * 4, piece * mat * 4, mode))
    check(lib.np_comm_wait())


def timed(fn):
    for _ in range(2):
        fn()
    D.sync()
    t = Timer()
    t.start()
    for _ in range(reps):
        fn()
    t.stop()
    return t.elapsed_ms() / reps


scratch = D.DeviceArray((slab, n, n))


def single_launch(loopback):
    """np_sgemm_strided_batched_allgather: ONE progress-reporting GEMM launch; piece c's transfer (here: an RCCL
    send/recv from the rank to itself, the only peer there is) is released by the GEMM's tile counter."""
    def run():
        check(lib.np_sgemm_strided_batched_allgather(slab, n, n, n, A.ptr, mat, B.ptr, mat, full.ptr, chunks, 2))
    check(lib.np_comm_debug_loopback(scratch.ptr if loopback else None, slab * mat * 4 if loopback else 0))
    t = timed(run)
    check(lib.np_comm_debug_loopback(None, 0))
    return t


print("serial (one stream)        %.3f ms per slab" % timed(serial))
print("ONE launch + tile counters, nothing to move      %.3f ms per slab (%d pieces)" % (single_launch(False), chunks))
print("ONE launch + tile counters, RCCL self send/recv  %.3f ms per slab (%d pieces)" % (single_launch(True), chunks))
assert (scratch.to_host()[piece - 1] == full.to_host()[slab - 1]).all()      # every piece lands at the start of the scratch: the last one stays
print("pipelined, ncclAllGather   %.3f ms per slab (%d pieces)" % (timed(lambda: pipelined(1)), chunks))
print("pipelined, p2p own-piece   %.3f ms per slab (%d pieces)" % (timed(lambda: pipelined(2)), chunks))
check(lib.np_sgemm_strided_batched(slab, n, n, n, A.ptr, mat, B.ptr, mat, work.ptr, mat))
D.sync()
print("GEMM alone                 %.3f ms per slab" % timed(lambda: check(lib.np_sgemm_strided_batched(slab, n, n, n, A.ptr, mat, B.ptr, mat, work.ptr, mat))))
assert (full.to_host()[slab - 1] == work.to_host()[slab - 1]).all()
check(lib.np_comm_destroy())
