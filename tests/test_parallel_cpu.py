"""N > 1 path on CPU: world_size-2 (and 3, ragged) gloo runs of the batch-sharding logic in
numpower_amd/parallel.py.  The compute callable is a plain torch.bmm here (test stand-in: the HIP
kernel itself is covered by the -m gpu tests); what is under test is the slab partition, the
in-place all-gather and the ragged padding path."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from numpower_amd import parallel, synth


def test_slab_partition_is_contiguous_and_balanced():
    for batch in (512, 7, 1, 10):
        for world in (1, 2, 3, 4, 8):
            slabs = parallel.all_slabs(batch, world)
            assert slabs[0].start == 0 and slabs[-1].stop == batch
            for a, b in zip(slabs, slabs[1:]):
                assert a.stop == b.start
            sizes = [s.size for s in slabs]
            assert max(sizes) - min(sizes) <= 1
    assert parallel.slab_for(512, 8, 3) == parallel.Slab(3, 192, 256)   # batch b -> rank b // 64


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, batch, gather, q, kind="matmul"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if kind == "elementwise":      # per-slice elementwise: slices of the leading axis are sharded
            slab = parallel.slab_for(batch, world, rank)
            mk = lambda seed: torch.stack([torch.from_numpy(synth.uniform((3, 5), seed + i, -1, 1))   # noqa: E731
                                           for i in range(slab.start, slab.stop)]) if slab.size else torch.empty((0, 3, 5))
            x, y = mk(300), mk(400)
            res = parallel.sharded_elementwise([x, y], batch, lambda a, b, out: torch.add(a, b, out=out),
                                               dist=dist, gather=gather)
            q.put((rank, res.numpy()))
            return
        m, k, n = 6, 5, 4
        slab = parallel.slab_for(batch, world, rank)
        a = torch.stack([torch.from_numpy(synth.uniform((m, k), 100 + i, -1, 1)) for i in range(slab.start, slab.stop)]) \
            if slab.size else torch.empty((0, m, k))
        b = torch.stack([torch.from_numpy(synth.uniform((k, n), 200 + i, -1, 1)) for i in range(slab.start, slab.stop)]) \
            if slab.size else torch.empty((0, k, n))

        def compute(x, y, out):
            torch.bmm(x, y, out=out)

        res = parallel.sharded_batched_matmul(a, b, batch, compute, dist=dist, gather=gather,
                                              overlap_chunks=int(kind[len("overlap"):] or 2) if kind.startswith("overlap") else 1)
        q.put((rank, res.numpy()))
    finally:
        dist.destroy_process_group()


def _run(world, batch, gather, kind="matmul"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, batch, gather, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


def _expected(batch):
    return np.stack([synth.uniform((6, 5), 100 + i, -1, 1) @ synth.uniform((5, 4), 200 + i, -1, 1)
                     for i in range(batch)])


@pytest.mark.parametrize("world,batch", [(2, 8), (2, 7), (3, 8)])
def test_sharded_batched_matmul_gathered(world, batch):
    out = _run(world, batch, gather=True)
    want = _expected(batch)
    for rank in range(world):
        np.testing.assert_allclose(out[rank], want, rtol=1e-6, atol=1e-6)


def test_sharded_batched_matmul_left_sharded():
    out = _run(2, 8, gather=False)
    want = _expected(8)
    np.testing.assert_allclose(out[0], want[:4], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(out[1], want[4:], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("world,batch,gather", [(2, 8, False), (2, 7, True), (3, 8, True)])
def test_sharded_elementwise(world, batch, gather):
    """north_star's "per-slice elementwise": left sharded there is no collective at all; gathered
    it is the same single all-gather (ragged slabs included)."""
    out = _run(world, batch, gather, kind="elementwise")
    want = np.stack([synth.uniform((3, 5), 300 + i, -1, 1) + synth.uniform((3, 5), 400 + i, -1, 1)
                     for i in range(batch)])
    for rank in range(world):
        if gather:
            np.testing.assert_array_equal(out[rank], want)
        else:
            slab = parallel.slab_for(batch, world, rank)
            np.testing.assert_array_equal(out[rank], want[slab.start:slab.stop])


@pytest.mark.parametrize("world,batch,chunks", [(2, 8, 2), (2, 10, 3), (3, 12, 2), (3, 15, 4), (2, 4, 8)])
def test_sharded_batched_matmul_overlapped_gather(world, batch, chunks):
    """overlap_chunks: per-piece compute + asynchronous point-to-point exchange straight into the right windows of
    the replicated result — the chunk index arithmetic of np_sgemm_strided_batched_allgather (np_comm_piece), with
    real peers: world 2 and 3, pieces that divide the slab and pieces that do not, more pieces than matrices."""
    out = _run(world, batch, gather=True, kind="overlap%d" % chunks)
    want = _expected(batch)
    for rank in range(world):
        np.testing.assert_allclose(out[rank], want, rtol=1e-6, atol=1e-6)


def test_piece_arithmetic_matches_the_c_abi():
    """np_comm_piece: contiguous, covering, as equal as they come, first pieces larger; clipped to the slab."""
    for slab in (1, 5, 8, 64, 65):
        for chunks in (1, 2, 3, 4, 8, 100):
            pcs = parallel.pieces_of(slab, chunks)
            assert len(pcs) == min(chunks, slab)
            assert pcs[0][0] == 0 and sum(c for _, c in pcs) == slab
            for (lo, c), (lo2, _) in zip(pcs, pcs[1:]):
                assert lo + c == lo2
            sizes = [c for _, c in pcs]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    assert parallel.pieces_of(64, 4) == [(0, 16), (16, 16), (32, 16), (48, 16)]


@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("chunks", [1, 2, 3, 8])
def test_c_abi_exchange_plan_replicates_the_result(world, chunks):
    """np_comm_debug_plan: the sends / receives np_sgemm_strided_batched_allgather issues on every rank, from the same
    functions as the real path (np_comm_piece, piece_addresses, p2p_peers) — played out on numpy buffers for worlds of
    2, 3 and 8 ranks: every send has exactly one matching receive of the same piece and size, every byte of every peer's
    slab arrives exactly once in the right place, nothing lands in a rank's own slab."""
    import ctypes as C
    from numpower_amd._lib import check, load
    lib = load()
    slab, item = 5, 12           # 5 items of 12 bytes per rank: pieces that do not divide evenly
    total = world * slab * item
    truth = np.arange(total, dtype=np.int64) % 251 + 1                      # the replicated result every rank must end with
    bufs = []
    for r in range(world):
        b = np.zeros(total, dtype=np.int64)
        b[r * slab * item:(r + 1) * slab * item] = truth[r * slab * item:(r + 1) * slab * item]   # its own slab, computed in place
        bufs.append(b)
    plans = []
    for r in range(world):
        n = C.c_size_t(0)
        check(lib.np_comm_debug_plan(r, world, slab, item, chunks, None, 0, C.byref(n)))
        assert n.value == min(chunks, slab) * (world - 1)
        rec = (C.c_ulonglong * (6 * n.value))()
        check(lib.np_comm_debug_plan(r, world, slab, item, chunks, rec, n.value, C.byref(n)))
        plans.append([tuple(rec[6 * i + k] for k in range(6)) for i in range(n.value)])
    written = [np.zeros(total, dtype=np.int64) for _ in range(world)]
    for r in range(world):
        for piece, to, send_off, nbytes, frm, recv_off in plans[r]:
            # the matching receive on the destination rank: same piece, same size, from this rank, into this rank's window
            match = [q for q in plans[to] if q[0] == piece and q[4] == r]
            assert len(match) == 1 and match[0][3] == nbytes, (r, to, piece)
            dst_off = match[0][5]
            assert dst_off == send_off                                      # a replicated result: same place on every rank
            assert r * slab * item <= send_off and send_off + nbytes <= (r + 1) * slab * item   # sent from its own slab only
            bufs[to][dst_off:dst_off + nbytes] = bufs[r][send_off:send_off + nbytes]
            written[to][dst_off:dst_off + nbytes] += 1
    for r in range(world):
        assert (bufs[r] == truth).all(), r
        own = slice(r * slab * item, (r + 1) * slab * item)
        assert (written[r][own] == 0).all()
        others = np.ones(total, dtype=bool)
        others[own] = False
        assert (written[r][others] == 1).all()


def test_piece_count_model_is_pinned():
    """np_sgemm_strided_batched_allgather(chunks = 0) asks the library's step model for the number of pieces
    (np_comm.hip: model_pieces) — host arithmetic, reachable without a device through
    np_comm_debug_model.  Pinned here: the choices for BASELINE config 5 at 2 / 4 / 8 ranks and the model's properties
    (nothing to overlap on one rank; a transfer-bound step is never modelled slower in pieces than whole; tiny slabs stay
    whole; a tie goes to fewer pieces)."""
    import ctypes as C
    from numpower_amd._lib import check, load
    lib = load()
    pick, ms = C.c_int(0), (C.c_double * 5)()

    def model(world, slab, m=1024, n=1024, k=1024, cus=256):
        check(lib.np_comm_debug_model(world, slab, m, n, k, cus, C.byref(pick), ms))
        return pick.value, list(ms)

    assert model(1, 512)[0] == 1                                   # one rank: nothing travels
    for world in (2, 4, 8):
        k, t = model(world, 512 // world)
        assert k in (2, 4, 8, 16)
        whole = t[0]
        best = min(t)
        assert t[[1, 2, 4, 8, 16].index(k)] <= 1.02 * best         # within 2 % of the best modelled step ...
        assert all(x > 1.02 * best for x in t[:[1, 2, 4, 8, 16].index(k)])   # ... and the smallest such count
        assert best < whole                                        # overlapping is modelled to pay at config 5's sizes
        # GEMM 2 * slab * 1024^3 / 130e12, transfer slab * 4 MiB / 153 GB/s: whole = their sum (+ the issue cost)
        slab = 512 // world
        assert abs(whole - (2.0 * slab * 1024 ** 3 / 130e12 * 1e3 + slab * 4 * 1024 ** 2 / 153e9 * 1e3 + 0.01)) < 1e-9
    assert model(8, 64)[0] == 8                                    # config 5 on a node (the default bench.py's "overlapped_auto" leg runs)
    assert model(8, 1)[0] == 1 and model(8, 1)[1][1] > 1e299      # a slab of one matrix cannot be cut
    k_small, _ = model(8, 64, 16, 16, 16)                          # microscopic matrices: the per-piece issue cost decides
    assert k_small == 1
    with pytest.raises(Exception, match="bad arguments"):
        check(lib.np_comm_debug_model(0, 64, 1024, 1024, 1024, 256, C.byref(pick), ms))
