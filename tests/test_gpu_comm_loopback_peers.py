"""The sharded batched matmul of BASELINE config 5 with PEER PROCESSES on the one GPU of a usual lease.

RCCL refuses two ranks on one device, so tests/test_gpu_comm_multi.py (real peers over RCCL) skips wherever this repository has
ever run.  Here the same worker script runs as 2, 3 and 4 ranks that SHARE device 0, over tests/loopback_rccl — a stand-in for
librccl.so.1 (test infrastructure, header of loopback_rccl.hip) that np_comm.hip picks up through LD_LIBRARY_PATH, no switch in
the product: rendezvous over TCP, every form of np_sgemm_strided_batched_allgather (one all-gather, grouped send / recv, 2 / 3 / 4
overlapped pieces, the library's own piece count; device-side flags, HIP events, one launch per piece, ONE progress-reporting
launch feeding the transfers), np_comm_max / barrier, destroy — every rank's replicated result bit-identical to every other
rank's and every other form's, within 1e-5 of the oracle's loop of 2-D matmuls.

What this proves and what it does not: np_comm's own logic with a peer — piece order and addresses, the two streams and their
device-side ordering, ranks that arrive skewed — not xGMI, peer-to-peer addressing or RCCL's kernels (the transport is a shared
host segment).  The first node still measures everything; this is what one GPU allows."""
import os
import textwrap
from pathlib import Path

import numpy as np
import pytest

from tests.test_gpu_comm_multi import ROOT, WORKER, _run_workers, check_abi_worker_results

pytestmark = pytest.mark.gpu
LOOPBACK = ROOT / "tests" / "loopback_rccl" / "lib"
LOOPBACK_VERSION = 99901


def _env():
    if not (LOOPBACK / "librccl.so.1").exists():          # test infrastructure: built by build_all(); a box that got the sources only builds it here
        from numpower_amd import build
        build.build_loopback_rccl()
    assert (LOOPBACK / "librccl.so.1").exists(), "tests/loopback_rccl/lib/librccl.so.1 is not built (python -m numpower_amd.build)"
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = str(LOOPBACK) + (":" + env["LD_LIBRARY_PATH"] if env.get("LD_LIBRARY_PATH") else "")
    return env


@pytest.mark.parametrize("world", [2, 3, 4])
def test_sharded_matmul_across_ranks_that_share_one_gpu(world, tmp_path, oracle):
    _run_workers(WORKER, world, tmp_path, "rank", extra=("0",), env=_env(), timeout=300)
    check_abi_worker_results(world, tmp_path, oracle, rccl_version=LOOPBACK_VERSION)


# One rank arrives late at every call, a gathered vector is checked element by element, and the ranks disagree about nothing:
# np_allgather (library stream), the *_async form + np_comm_wait, np_comm_max, np_comm_barrier.
SKEW_WORKER = textwrap.dedent("""
    import ctypes as C, sys, time
    import numpy as np
    sys.path.insert(0, %r)
    from numpower_amd import device as D
    from numpower_amd._lib import check, load
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    D.init(0)
    lib = load()
    if rank == world - 1:
        time.sleep(1.5)                                   # the last rank is late for the rendezvous ...
    check(lib.np_comm_init(rank, world, ("tcp://127.0.0.1:%%d" %% port).encode()))
    results = {}
    for rep, n in enumerate((1, 1000, 300_001, 3_000_000)):
        mine = (np.arange(n, dtype=np.float32) + 1.0) * (rank + 1)
        d_mine, d_all = D.DeviceArray.from_host(mine), D.DeviceArray((world * n,))
        D.fill(d_all, float("nan"))
        if rank == rep %% world:
            time.sleep(0.3)                               # ... and a different rank is late for every collective
        check(lib.np_allgather(d_mine.ptr, d_all.ptr, 4 * n))
        got = d_all.to_host()
        want = np.concatenate([(np.arange(n, dtype=np.float32) + 1.0) * (r + 1) for r in range(world)])
        assert (got == want).all(), (rep, n)
        D.fill(d_all, float("nan"))
        check(lib.np_allgather_async(d_mine.ptr, d_all.ptr, 4 * n, 4 * n, 1))          # one collective on the communication stream
        check(lib.np_comm_wait())
        assert (d_all.to_host() == want).all(), ("async", rep, n)
        d_wide = D.DeviceArray((world * (n + 16),))                                     # destinations 64 bytes apart: grouped send / recv
        D.fill(d_wide, -1.0)
        check(lib.np_allgather_async(d_mine.ptr, d_wide.ptr, 4 * n, 4 * (n + 16), 2))
        check(lib.np_comm_wait())
        wide = d_wide.to_host().reshape(world, n + 16)
        assert (wide[:, :n].reshape(-1) == want).all() and (wide[:, n:] == -1.0).all(), ("p2p", rep, n)
        d_wide.free()
        mx = C.c_float(0)
        check(lib.np_comm_max(float(10 * rank + rep), C.byref(mx)))
        assert mx.value == float(10 * (world - 1) + rep), (mx.value, rep)
        check(lib.np_comm_barrier())
        d_mine.free(); d_all.free()
    check(lib.np_comm_destroy())
    # a second communicator in the same processes, rendezvous through a FILE this time (rank 0 publishes the id, peers poll)
    import os
    check(lib.np_comm_init(rank, world, os.path.join(os.path.dirname(out), "rendezvous.id").encode()))
    assert lib.np_comm_world() == world and lib.np_comm_rank() == rank
    mine = np.full(1000, float(rank + 1), dtype=np.float32)
    d_mine, d_all = D.DeviceArray.from_host(mine), D.DeviceArray((world * 1000,))
    check(lib.np_allgather(d_mine.ptr, d_all.ptr, 4000))
    assert (d_all.to_host().reshape(world, 1000) == np.arange(1, world + 1, dtype=np.float32)[:, None]).all()
    check(lib.np_comm_barrier())
    check(lib.np_comm_destroy())
    assert lib.np_comm_world() == 0
    np.savez(out, ok=np.ones(1, dtype=np.float32))
    print("OK")
""") % str(ROOT)


@pytest.mark.parametrize("world", [2, 3])
def test_gathers_with_skewed_ranks(world, tmp_path):
    _run_workers(SKEW_WORKER, world, tmp_path, "skew", env=_env(), timeout=200)
    for r in range(world):
        assert np.load(tmp_path / ("skew%d.npz" % r))["ok"][0] == 1


# A peer that dies between two gathers.  The survivor's second sharded product is enqueued (the call itself does not block), its
# library stream waits for transfers that never arrive — for np_comm_set_wait_limit seconds, then the device error word is
# raised: np_sync says NP_ERR_DEVICE ("gave up"), nothing is handed over as if it had been gathered.  np_comm_destroy then finds
# the communication stream still inside the collective library's kernel and aborts the communicator; only then can the error be
# acknowledged (np_clear_device_error waits for the whole device: refused while that kernel is there); the process goes on
# computing.  (With RCCL the stuck kernel is RCCL's and ncclCommAbort is what makes it leave; here it is the loopback library's.)
DEAD_PEER_WORKER = textwrap.dedent("""
    import ctypes as C, os, sys, time
    import numpy as np
    sys.path.insert(0, %r)
    from numpower_amd import device as D, synth
    from numpower_amd._lib import check, load
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    D.init(0)
    lib = load()
    check(lib.np_comm_init(rank, world, ("tcp://127.0.0.1:%%d" %% port).encode()))
    slab, m, k, n = 3, 128, 96, 64
    A = np.stack([synth.uniform((m, k), 70 + i, -1.0, 1.0) for i in range(rank * slab, (rank + 1) * slab)])
    B = np.stack([synth.uniform((k, n), 80 + i, -1.0, 1.0) for i in range(rank * slab, (rank + 1) * slab)])
    dA, dB, full = D.DeviceArray.from_host(A), D.DeviceArray.from_host(B), D.DeviceArray((world * slab, m, n))
    check(lib.np_sgemm_strided_batched_allgather(slab, m, n, k, dA.ptr, m * k, dB.ptr, k * n, full.ptr, 2, 0))
    check(lib.np_sync())
    first = full.to_host()
    check(lib.np_comm_barrier())
    if rank == 1:
        np.savez(out, ok=np.ones(1, dtype=np.float32))
        print("OK", flush=True)
        os._exit(0)                                        # gone: no destroy, no second gather
    time.sleep(0.5)
    check(lib.np_comm_set_wait_limit(2.0))
    D.fill(full, float("nan"))
    rc_call = lib.np_sgemm_strided_batched_allgather(slab, m, n, k, dA.ptr, m * k, dB.ptr, k * n, full.ptr, 2, 0)
    t0 = time.time()
    rc_sync = lib.np_sync() if rc_call == 0 else rc_call
    waited = time.time() - t0
    message = lib.np_last_error().decode()
    assert rc_sync != 0 and "gave up" in message, (rc_call, rc_sync, message)
    assert waited < 20.0, waited
    assert lib.np_sync() != 0                              # sticky until acknowledged
    # acknowledging means waiting for the whole device — which a transfer that waits for the dead peer would never let return:
    # refused (after its grace period) until the communicator is gone
    bits = C.c_uint(0)
    t0 = time.time()
    rc_clear = lib.np_clear_device_error(C.byref(bits))
    clear_s = time.time() - t0
    assert rc_clear != 0 and "np_comm_destroy" in lib.np_last_error().decode() and clear_s < 8.0, (rc_clear, clear_s, lib.np_last_error())
    t0 = time.time()
    rc_destroy = lib.np_comm_destroy()
    destroy_s = time.time() - t0
    destroy_message = lib.np_last_error().decode()
    assert rc_destroy != 0 and "aborted" in destroy_message, (rc_destroy, destroy_message)
    assert destroy_s < 20.0, destroy_s          # its grace period follows the wait limit (2 s), then 5 s for the communication stream
    check(lib.np_clear_device_error(C.byref(bits)))
    assert bits.value != 0
    check(lib.np_sync())
    # the process goes on: a local product of its own slab, bit-identical to its window of the first (gathered) result
    local = D.DeviceArray((slab, m, n))
    check(lib.np_sgemm_strided_batched(slab, m, n, k, dA.ptr, m * k, dB.ptr, k * n, local.ptr, m * n))
    check(lib.np_sync())
    assert (local.to_host().view(np.uint32) == first[:slab].view(np.uint32)).all()
    np.savez(out, ok=np.ones(1, dtype=np.float32), waited=np.array([waited, destroy_s], dtype=np.float32))
    print("OK", flush=True)
""") % str(ROOT)


def test_a_peer_that_dies_ends_as_an_error_not_as_a_hang(tmp_path):
    env = _env()
    env["NP_LOOPBACK_GIVE_UP_S"] = "120"          # the stand-in's own waits must not end first: ncclCommAbort is under test
    _run_workers(DEAD_PEER_WORKER, 2, tmp_path, "dead", env=env, timeout=200)
    got = np.load(tmp_path / "dead0.npz")
    assert got["ok"][0] == 1 and got["waited"][0] < 20.0
