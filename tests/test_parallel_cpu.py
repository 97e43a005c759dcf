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
