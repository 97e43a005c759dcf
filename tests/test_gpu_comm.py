"""np_comm_* — the C-ABI collective (BASELINE config 5's all-gather without torch): world 1 on the one GPU a
test box has (RCCL refuses two ranks on one device, so the multi-rank form runs only where the driver has a
multi-GPU node: bench.py --gpus N exercises it there), both rendezvous forms, error paths, and the sharded
batched matmul written the way a C host would write it: np_sgemm_strided_batched into the rank's slab of the
full result + np_allgather in place."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

from numpower_amd import synth
from numpower_amd._lib import check, load, NumPowerError

pytestmark = pytest.mark.gpu


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("form", ["tcp", "file"])
def test_world_1_allgather_max_barrier(form, hip, tmp_path):
    lib = load()
    endpoint = ("tcp://127.0.0.1:%d" % free_port()) if form == "tcp" else str(tmp_path / "np_comm_id")
    assert lib.np_comm_world() == 0 and lib.np_comm_rank() == -1
    with pytest.raises(NumPowerError, match="no communicator"):
        check(lib.np_allgather(1, 2, 4))
    check(lib.np_comm_init(0, 1, endpoint.encode()))
    try:
        assert lib.np_comm_world() == 1 and lib.np_comm_rank() == 0
        with pytest.raises(NumPowerError, match="already exists"):
            check(lib.np_comm_init(0, 1, endpoint.encode()))
        x = synth.uniform((3, 1000), 5, -1.0, 1.0)
        dx, dy = hip.DeviceArray.from_host(x), hip.DeviceArray((3, 1000))
        check(lib.np_allgather(dx.ptr, dy.ptr, x.nbytes))                 # out of place
        assert (dy.to_host() == x).all()
        check(lib.np_allgather(dx.ptr, dx.ptr, x.nbytes))                 # in place (rank 0's slab is the buffer)
        assert (dx.to_host() == x).all()
        m = C.c_float(0.0)
        check(lib.np_comm_max(3.25, C.byref(m)))
        assert m.value == 3.25
        check(lib.np_comm_barrier())
    finally:
        check(lib.np_comm_destroy())
    assert lib.np_comm_world() == 0
    if form == "file":
        assert not os.path.exists(endpoint)


def test_sharded_batched_matmul_through_the_c_abi(hip):
    """config 5 as a C host writes it (world 1: the slab is the whole batch)."""
    lib = load()
    batch, n = 4, 256
    A = synth.uniform((batch, n, n), 12, -1.0, 1.0)
    B = synth.uniform((batch, n, n), 13, -1.0, 1.0)
    check(lib.np_comm_init(0, 1, ("tcp://127.0.0.1:%d" % free_port()).encode()))
    try:
        rank, world = lib.np_comm_rank(), lib.np_comm_world()
        per = batch // world
        dA, dB = hip.DeviceArray.from_host(A[rank * per:(rank + 1) * per]), hip.DeviceArray.from_host(B[rank * per:(rank + 1) * per])
        full = hip.DeviceArray((batch, n, n))
        mine = full.ptr + rank * per * n * n * 4
        check(lib.np_sgemm_strided_batched(per, n, n, n, dA.ptr, n * n, dB.ptr, n * n, mine, n * n))
        check(lib.np_allgather(mine, full.ptr, per * n * n * 4))
        got = full.to_host().astype(np.float64)
    finally:
        check(lib.np_comm_destroy())
    want = A.astype(np.float64) @ B.astype(np.float64)
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    assert (np.abs(got - want) <= 1e-6 * scale).all()


@pytest.mark.parametrize("variant", [0, 1, 2, 3], ids=["default", "events", "flags+launch-per-piece", "flags+one-launch"])
@pytest.mark.parametrize("chunks,mode", [(1, 0), (1, 1), (1, 2), (2, 0), (4, 2), (5, 0), (8, 0), (64, 0)])
def test_overlapped_sharded_matmul_is_bit_identical(chunks, mode, variant, hip, oracle):
    """np_sgemm_strided_batched_allgather (GEMM pieces on the library stream, each piece's gather on the communication
    stream) against the plain form — np_sgemm_strided_batched + np_allgather on one stream — bit for bit, for every
    chunk count / transport, and against the oracle's loop of 2-D matmuls (the reference has no batched entry point,
    linalg.c:239-242)."""
    lib = load()
    batch, m, k, n = 8, 96, 160, 64
    A = synth.uniform((batch, m, k), 12, -1.0, 1.0)
    B = synth.uniform((batch, k, n), 13, -1.0, 1.0)
    check(lib.np_comm_init(0, 1, ("tcp://127.0.0.1:%d" % free_port()).encode()))
    try:
        assert lib.np_comm_stream() and lib.np_comm_stream() != lib.np_get_stream()
        dA, dB = hip.DeviceArray.from_host(A), hip.DeviceArray.from_host(B)
        plain, over = hip.DeviceArray((batch, m, n)), hip.DeviceArray((batch, m, n))
        check(lib.np_sgemm_strided_batched(batch, m, n, k, dA.ptr, m * k, dB.ptr, k * n, plain.ptr, m * n))
        check(lib.np_allgather(plain.ptr, plain.ptr, batch * m * n * 4))
        check(lib.np_comm_set_variant(variant))
        if variant != 1:
            assert lib.np_comm_sync_mode() == 1, "device-side flags did not pass the self-test on this box"
        for rep in range(3):                              # repeated: the tile counters must be clean again after each call
            hip.fill(over, float("nan"))
            check(lib.np_sgemm_strided_batched_allgather(batch, m, n, k, dA.ptr, m * k, dB.ptr, k * n, over.ptr, chunks, mode))
            got = over.to_host()                          # to_host() is on the library stream: ordered behind np_comm_wait
            assert not np.isnan(got).any(), rep
        want = plain.to_host()
    finally:
        check(lib.np_comm_set_variant(0))
        check(lib.np_comm_destroy())
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    for i in range(batch):
        ref = oracle.matmul(A[i], B[i])
        scale = np.abs(A[i]).astype(np.float64) @ np.abs(B[i]).astype(np.float64)
        assert (np.abs(got[i] - ref) / scale).max() <= 1e-5


@pytest.mark.parametrize("shape", [(12, 1001, 1003, 1001), (4, 1024, 1024, 1024), (5, 1000, 1002, 1000)],
                         ids=["odd rows", "aligned", "N % 4 = 2"])
def test_overlapped_sharded_matmul_progress_launch_any_alignment(shape, hip):
    """Matrices large enough for the ONE progress-reporting launch of the LDS-DMA kernel (its workgroups count finished
    tiles per piece), including operands whose rows are not float4-loadable (the kernel reads them as they are and
    runs its ragged instantiation; twelve odd matrices fill the machine well enough for the planner to pick it), and
    shapes whose plan is not that launch (five 1000 x 1002 matrices: 64 x 64 tiles) and fall back to one launch per piece —
    where a piece of ONE matrix must still run the whole batch's configuration (np::sgemm_batched_piece), not the
    single-product planner's.  Bit-identical to the plain form, within 1e-6 |A|.|B| of fp64, three pieces."""
    lib = load()
    batch, m, k, n = shape
    A = synth.uniform((batch, m, k), 14, -1.0, 1.0)
    B = synth.uniform((batch, k, n), 15, -1.0, 1.0)
    check(lib.np_comm_init(0, 1, ("tcp://127.0.0.1:%d" % free_port()).encode()))
    try:
        dA, dB = hip.DeviceArray.from_host(A), hip.DeviceArray.from_host(B)
        plain, over = hip.DeviceArray((batch, m, n)), hip.DeviceArray((batch, m, n))
        check(lib.np_sgemm_strided_batched(batch, m, n, k, dA.ptr, m * k, dB.ptr, k * n, plain.ptr, m * n))
        for rep in range(2):
            hip.fill(over, float("nan"))
            check(lib.np_sgemm_strided_batched_allgather(batch, m, n, k, dA.ptr, m * k, dB.ptr, k * n, over.ptr, 3, 0))
            got = over.to_host()
            assert not np.isnan(got).any(), rep
        want = plain.to_host()
    finally:
        check(lib.np_comm_destroy())
    assert (got.view(np.uint32) == want.view(np.uint32)).all()
    ref = A.astype(np.float64) @ B.astype(np.float64)
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    assert (np.abs(got.reshape(ref.shape) - ref) / scale).max() <= 1e-6


def test_pieces_of_a_batch_run_the_batch_s_kernels(hip):
    """np_sgemm_strided_batched_piece: five 1000 x 1002 x 1000 products as pieces of 2, 2 and 1 matrices are bit for bit
    the one call over all five (a one-matrix piece through np_sgemm_strided_batched itself is planned as a single
    product — split-K here — and differs in its last bits: the reason the entry point exists)."""
    lib = load()
    batch, m, n, k = 5, 1000, 1002, 1000
    A = synth.uniform((batch, m, k), 16, -1.0, 1.0)
    B = synth.uniform((batch, k, n), 17, -1.0, 1.0)
    dA, dB = hip.DeviceArray.from_host(A), hip.DeviceArray.from_host(B)
    one, pieces = hip.DeviceArray((batch, m, n)), hip.DeviceArray((batch, m, n))
    check(lib.np_sgemm_strided_batched(batch, m, n, k, dA.ptr, m * k, dB.ptr, k * n, one.ptr, m * n))
    for lo, count in ((0, 2), (2, 2), (4, 1)):
        check(lib.np_sgemm_strided_batched_piece(count, batch, m, n, k, dA.ptr + 4 * lo * m * k, m * k, dB.ptr + 4 * lo * k * n, k * n,
                                                 pieces.ptr + 4 * lo * m * n, m * n))
    assert (one.to_host().view(np.uint32) == pieces.to_host().view(np.uint32)).all()
    with pytest.raises(NumPowerError, match="a piece of 3 matrices of a batch of 2"):
        check(lib.np_sgemm_strided_batched_piece(3, 2, m, n, k, dA.ptr, m * k, dB.ptr, k * n, pieces.ptr, m * n))


def test_async_gather_out_of_place_and_strided(hip):
    """np_allgather_async: the copy lands on the communication stream behind what the library stream produced, and
    np_comm_wait orders the library stream behind it — out of place (world 1: the own piece is copied into place),
    with a destination stride, both transports."""
    lib = load()
    check(lib.np_comm_init(0, 1, ("tcp://127.0.0.1:%d" % free_port()).encode()))
    try:
        n = 1 << 20
        src, dst = hip.DeviceArray((n,)), hip.DeviceArray((2 * n,))
        for mode in (0, 1, 2):
            hip.fill(src, 1.5 + mode)         # produced on the library stream ...
            hip.fill(dst, -1.0)
            check(lib.np_allgather_async(src.ptr, dst.ptr, n * 4, n * 4, mode))      # ... picked up by the other stream
            check(lib.np_comm_wait())
            out = dst.to_host()
            assert (out[:n] == 1.5 + mode).all() and (out[n:] == -1.0).all()
        hip.fill(dst, -1.0)
        check(lib.np_allgather_async(src.ptr, dst.ptr + 4 * 100, n * 4, 2 * n * 4, 0))   # strided: p2p path, own piece copied
        check(lib.np_comm_wait())
        out = dst.to_host()
        assert (out[100:100 + n] == 3.5).all() and (out[:100] == -1.0).all()
        with pytest.raises(NumPowerError, match="contiguous destinations"):
            check(lib.np_allgather_async(src.ptr, dst.ptr, n * 4, 2 * n * 4, 1))
        with pytest.raises(NumPowerError, match="overlap"):
            check(lib.np_allgather_async(src.ptr, dst.ptr, n * 4, n * 2, 0))
        with pytest.raises(NumPowerError, match="NP_GATHER_COLLECTIVE"):
            check(lib.np_sgemm_strided_batched_allgather(4, 8, 8, 8, src.ptr, 64, src.ptr, 64, dst.ptr, 2, 1))
    finally:
        check(lib.np_comm_destroy())


def test_p2p_transport_to_self(hip):
    """The grouped ncclSend / ncclRecv pair the chunked gather is made of, run on the only peer a one-GPU box has:
    the rank itself."""
    lib = load()
    check(lib.np_comm_init(0, 1, ("tcp://127.0.0.1:%d" % free_port()).encode()))
    try:
        x = synth.uniform((1 << 18,), 21, -1.0, 1.0)
        src, dst = hip.DeviceArray.from_host(x), hip.DeviceArray((1 << 18,))
        hip.fill(dst, 0.0)
        check(lib.np_comm_debug_sendrecv_self(src.ptr, dst.ptr, x.nbytes))
        assert (dst.to_host() == x).all()
    finally:
        check(lib.np_comm_destroy())


def test_async_gather_does_not_hold_up_the_library_stream(hip):
    """The point of the second stream, shown without a race on timing noise: a large out-of-place gather (1 GiB:
    ~0.4 ms of copy on one GPU) is handed to the communication stream, then a tiny kernel goes to the library stream.
    Event-timed on the LIBRARY stream alone — no np_comm_wait inside the bracket — the pair must take a small
    fraction of the copy's own time: the library stream did not wait for the transfer.  With np_allgather (same
    stream) the same bracket contains the whole copy."""
    lib = load()
    from numpower_amd._lib import Timer
    check(lib.np_comm_init(0, 1, ("tcp://127.0.0.1:%d" % free_port()).encode()))
    try:
        n = 1 << 28                                   # 2^28 floats = 1 GiB
        src, dst, tiny = hip.DeviceArray((n,)), hip.DeviceArray((n,)), hip.DeviceArray((1024,))
        hip.fill(src, 1.0)
        hip.fill(dst, 0.0)
        check(lib.np_sync())

        def bracket(fn, reps=5):
            best = 1e9
            for _ in range(reps):
                check(lib.np_comm_wait())
                check(lib.np_sync())
                t = Timer()
                t.start()
                fn()
                hip.fill(tiny, 2.0)                   # the library stream's next piece of work
                t.stop()
                best = min(best, t.elapsed_ms())
                check(lib.np_comm_wait())
                check(lib.np_sync())
            return best

        t_same = bracket(lambda: check(lib.np_allgather(src.ptr, dst.ptr, n * 4)))
        t_async = bracket(lambda: check(lib.np_allgather_async(src.ptr, dst.ptr, n * 4, n * 4, 0)))
        print("1 GiB gather + tiny kernel, timed on the library stream: same stream %.3f ms, two streams %.3f ms" % (t_same, t_async))
        assert (dst.to_host()[::4097] == 1.0).all()
        assert t_same >= 0.2, t_same                 # the copy really is in the one-stream bracket
        assert t_async <= 0.25 * t_same, (t_same, t_async)
    finally:
        check(lib.np_comm_destroy())


def test_bad_arguments(hip):
    lib = load()
    for args in ((1, 1, b"tcp://127.0.0.1:1"), (-1, 2, b"x"), (0, 0, b"x"), (0, 1, b"")):
        with pytest.raises(NumPowerError):
            check(lib.np_comm_init(*args))


def test_a_transfer_issued_next_to_a_running_gemm_gets_through(hip):
    """The overlapped pipeline of config 5 rests on an RCCL transfer making progress WHILE a GEMM owns the CUs (VERDICT r03
    weak #4: a scheduling assumption).  World > 1 cannot run on a one-GPU lease; the transport can — self send / recv pairs on
    the communication stream (np_comm_debug_loopback_timed: not ordered behind the library stream, each with its own event
    pair) next to a queue of GEMMs on the library stream.  What was measured (profiles/r04/comm_contention.log) and is held
    here with slack: a 32 MiB transfer takes 0.034 ms alone; next to the slab GEMM of config 5 (64 x 1024^3: four rounds of
    workgroups, some retire all the time) 0.06-0.33 ms — it waits for CUs, a fraction of a round; next to a 4096^3 product
    (512 workgroups = ONE resident round) up to one whole product, ~1 ms.  Not the 1.3 x stand-alone one would wish for —
    but bounded by one round of the GEMM's workgroups, and far from "after the queue": the transfers are back while >= 80 % of
    the GEMM work is still queued."""
    import time
    lib = load()
    check(lib.np_comm_init(0, 1, ("tcp://127.0.0.1:%d" % free_port()).encode()))
    try:
        nbytes, count = 32 << 20, 8
        src, dst = hip.DeviceArray((nbytes // 4,)), hip.DeviceArray((nbytes // 4,))
        hip.fill(src, 1.25)
        hip.fill(dst, 0.0)
        ms = (C.c_float * count)()

        def transfers():
            check(lib.np_comm_debug_loopback_timed(src.ptr, dst.ptr, nbytes, count, ms))
            return np.array(list(ms))

        transfers()
        alone = float(np.median(transfers()))
        assert (dst.to_host() == np.float32(1.25)).all()
        assert alone < 1.0, alone                       # measured 0.2 ms (32 MB inside one GPU)

        def under(launch, loops):
            for _ in range(3):
                launch()
            hip.sync()
            t0 = time.perf_counter()
            for _ in range(loops):
                launch()
            t = transfers()
            back = time.perf_counter() - t0
            hip.sync()
            return t, back, time.perf_counter() - t0

        per, m = 64, 1024
        A, B, Cm = hip.DeviceArray((per, m, m)), hip.DeviceArray((per, m, m)), hip.DeviceArray((per, m, m))
        hip.fill(A, 0.5)
        hip.fill(B, 0.25)
        t, back, done = under(lambda: check(lib.np_sgemm_strided_batched(per, m, m, m, A.ptr, m * m, B.ptr, m * m, Cm.ptr, m * m)), 40)
        assert back < 0.5 * done, ("the transfers waited for the GEMM queue", back, done)    # behind the queue: back ~ done
        assert t.max() <= alone + 1.6, ("next to the slab GEMM", t, alone)       # measured <= 0.33 ms; behind the queue: 40 ms
        n = 4096
        A2, B2, C2 = hip.DeviceArray((n, n)), hip.DeviceArray((n, n)), hip.DeviceArray((n, n))
        hip.fill(A2, 0.5)
        hip.fill(B2, 0.25)
        t, back, done = under(lambda: hip.sgemm(A2, B2, out=C2), 40)
        assert back < 0.5 * done, ("the transfers waited for the GEMM queue", back, done)
        assert t.max() <= alone + 4.0, ("next to 4096^3 (one resident round): at most ~one product", t, alone)   # measured <= 0.98 ms; behind the queue: 40 ms
    finally:
        check(lib.np_comm_destroy())


def test_transfer_wait_limit_is_configurable(hip):
    """np_comm_set_wait_limit: the bound on the library stream's wait for transfers (it used to be unbounded, a
    dead peer was an unkillable hang).  Refuses nonsense, and a tight limit does not disturb transfers that do complete."""
    lib = load()
    with pytest.raises(NumPowerError, match="not a limit"):
        check(lib.np_comm_set_wait_limit(-1.0))
    with pytest.raises(NumPowerError, match="not a limit"):
        check(lib.np_comm_set_wait_limit(float("nan")))
    batch, n = 8, 128
    A = synth.uniform((batch, n, n), 12, -1.0, 1.0)
    B = synth.uniform((batch, n, n), 13, -1.0, 1.0)
    check(lib.np_comm_init(0, 1, ("tcp://127.0.0.1:%d" % free_port()).encode()))
    try:
        check(lib.np_comm_set_wait_limit(0.5))
        dA, dB = hip.DeviceArray.from_host(A), hip.DeviceArray.from_host(B)
        out = hip.DeviceArray((batch, n, n))
        for chunks in (1, 4):
            check(lib.np_sgemm_strided_batched_allgather(batch, n, n, n, dA.ptr, n * n, dB.ptr, n * n, out.ptr, chunks, 0))
            got = out.to_host().astype(np.float64)
            scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
            assert (np.abs(got - A.astype(np.float64) @ B.astype(np.float64)) <= 1e-6 * scale).all()
        check(lib.np_comm_set_wait_limit(0.0))      # never give up (the behaviour before this round) is still selectable
        check(lib.np_sgemm_strided_batched_allgather(batch, n, n, n, dA.ptr, n * n, dB.ptr, n * n, out.ptr, 2, 0))
        check(lib.np_sync())
    finally:
        check(lib.np_comm_set_wait_limit(600.0))
        check(lib.np_comm_destroy())
