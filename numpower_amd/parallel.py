"""Multi-GPU extension of the hot path: batch-parallel sharding.

The reference has nothing multi-GPU (NDArray::setDevice only, numpower.c:615-635).  Every hot-path
op is independent per array, so ranks normally run as replicas with no communication at all.  The
one sharded case (BASELINE config 5) is a batched matmul: batch b of B goes to rank b // (B / N)
— contiguous slabs, so each rank's share is ONE strided-batched GEMM launch — and, when a
replicated result is wanted, the result slabs are combined with ONE all-gather (RCCL over xGMI:
every peer's slab arrives over its own link).  No other collective exists on the path.

One process per GPU, torch.distributed as plumbing (backend "nccl" = RCCL on ROCm, "gloo" in the
CPU tests).  The compute is injected as a callable so that this module has no kernel code and the
CPU tests can drive the partition + gather logic without a GPU.
"""
from __future__ import annotations

from dataclasses import dataclass


@dataclass(frozen=True)
class Slab:
    """Contiguous share [start, stop) of a batch owned by one rank."""
    rank: int
    start: int
    stop: int

    @property
    def size(self) -> int:
        return self.stop - self.start


def slab_for(batch: int, world_size: int, rank: int) -> Slab:
    """Contiguous, balanced partition: the first batch % world ranks get one extra matrix."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, extra = divmod(batch, world_size)
    start = rank * base + min(rank, extra)
    return Slab(rank, start, start + base + (1 if rank < extra else 0))


def all_slabs(batch: int, world_size: int):
    return [slab_for(batch, world_size, r) for r in range(world_size)]


def pieces_of(slab_size: int, chunks: int):
    """[(lo, count)] of a slab cut into `chunks` pieces exactly as np_sgemm_strided_batched_allgather cuts it — the
    arithmetic is asked of the library itself (np_comm_piece: pure, needs no device), so the torch path and the C-ABI
    path cannot drift apart."""
    import ctypes as C

    from ._lib import check, load
    lib = load()
    chunks = max(1, min(int(chunks), int(slab_size))) if slab_size else 1
    out = []
    for c in range(chunks):
        lo, count = C.c_size_t(0), C.c_size_t(0)
        check(lib.np_comm_piece(slab_size, chunks, c, C.byref(lo), C.byref(count)))
        out.append((int(lo.value), int(count.value)))
    return out


def exchange_piece(dist, full, slab_size: int, lo: int, count: int):
    """Piece [lo, lo + count) of every rank's slab straight into place: full[r * slab + lo ...] <- rank r's piece, as
    one batch of point-to-point transfers (RCCL: one grouped ncclSend / ncclRecv exchange, each peer's piece over that
    peer's own xGMI link; no flattened staging tensor, no copy-out — what dist.all_gather() on a list of non-contiguous
    windows would do).  Same matching as np_comm.hip: at step s rank r sends to r + s and receives from r - s.
    Returns the work handles; the own piece is already in place."""
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = full[rank * slab_size + lo:rank * slab_size + lo + count]
    ops = []
    for step in range(1, world):
        to, frm = (rank + step) % world, (rank - step) % world
        ops.append(dist.P2POp(dist.isend, mine, to))
        ops.append(dist.P2POp(dist.irecv, full[frm * slab_size + lo:frm * slab_size + lo + count], frm))
    return dist.batch_isend_irecv(ops) if ops else []


def _sharded(slabs_in, batch: int, item_shape, compute, dist, gather: bool, overlap_chunks: int = 1):
    """Shared body: every rank computes its contiguous slab of `batch` independent items straight
    into its window of the result; gather=True adds the ONE all-gather of the path.

    overlap_chunks > 1 (equal slabs only): the slab is computed in that many pieces (pieces_of) and
    each piece's exchange is issued asynchronously right behind its compute — the transfers run on
    the process group's own stream while the next piece computes, so the xGMI transfer (the longer
    leg at 8 GPUs: 1.75 ms against 0.9 ms of GEMM for BASELINE config 5) hides behind compute
    instead of following it.  The C-ABI form of the same pipeline is
    np_sgemm_strided_batched_allgather (np_comm.hip)."""
    import torch

    if dist is None:
        import torch.distributed as dist   # noqa: F811
    world, rank = dist.get_world_size(), dist.get_rank()
    slab = slab_for(batch, world, rank)
    first = slabs_in[0]
    for x in slabs_in:
        if x.shape[0] != slab.size:
            raise ValueError("rank %d expects a slab of %d items, got %d" % (rank, slab.size, x.shape[0]))
    item_shape = tuple(int(v) for v in item_shape)
    if not gather:
        out = torch.empty((slab.size,) + item_shape, dtype=first.dtype, device=first.device)
        compute(*slabs_in, out)
        return out
    full = torch.empty((batch,) + item_shape, dtype=first.dtype, device=first.device)
    mine = full[slab.start:slab.stop]
    if world > 1 and batch % world == 0 and overlap_chunks > 1:
        handles = []
        # a compute that knows how (hip_compute) is told the piece belongs to a slab of slab.size items: it then runs the
        # kernels the whole slab would have run, and the pipelined result is bit-identical to the one-call form
        extra = {"whole": slab.size} if getattr(compute, "accepts_whole", False) else {}
        for lo, count in pieces_of(slab.size, overlap_chunks):
            compute(*[x[lo:lo + count] for x in slabs_in], mine[lo:lo + count], **extra)
            # piece c of every rank's slab lands at rank * slab + lo of the result
            handles.extend(exchange_piece(dist, full, slab.size, lo, count))
        for h in handles:
            h.wait()
        return full
    compute(*slabs_in, mine)   # written in place: the gather needs no staging copy
    if world == 1:
        return full
    if batch % world == 0:
        # equal slabs: a single in-place all-gather into the result tensor
        dist.all_gather_into_tensor(full.view(-1), mine.reshape(-1))
    else:
        # ragged slabs: pad every contribution to the largest slab so that it is still ONE
        # equal-count all-gather, then drop the padding
        slabs = all_slabs(batch, world)
        pad = max(s.size for s in slabs)
        send = torch.zeros((pad,) + item_shape, dtype=full.dtype, device=full.device)
        send[:slab.size].copy_(mine)
        recv = torch.empty((world, pad) + item_shape, dtype=full.dtype, device=full.device)
        dist.all_gather_into_tensor(recv.view(-1), send.view(-1))
        for s in slabs:
            if s.rank != rank:
                full[s.start:s.stop].copy_(recv[s.rank, :s.size])
    return full


def sharded_batched_matmul(a_slab, b_slab, batch: int, compute, dist=None, gather: bool = True,
                           overlap_chunks: int = 1):
    """Batched matmul with the batch sharded over the ranks of `dist`'s default group.

    a_slab / b_slab : this rank's slab of the operands, torch tensors [slab, M, K] / [slab, K, N]
                      on the rank's device
    compute(a, b, out): writes the slab product into `out` [slab, M, N] (the HIP strided-batched
                      GEMM on GPU ranks)
    gather=True     : returns the replicated [batch, M, N] result (one all-gather);
    gather=False    : returns this rank's [slab, M, N] result only (no collective at all).
    """
    return _sharded([a_slab, b_slab], batch, (a_slab.shape[1], b_slab.shape[2]), compute, dist, gather,
                    overlap_chunks)


def sharded_elementwise(slabs, batch: int, compute, dist=None, gather: bool = False):
    """Per-slice elementwise work with the leading axis sharded (BASELINE north_star: "per-slice
    elementwise"): `slabs` are this rank's [slab, ...] shares of one or more equally shaped
    operands, compute(*slabs, out) writes the slab result.  Elementwise results are normally left
    sharded (gather=False: no collective on the path at all); gather=True replicates them with the
    same single all-gather as the batched matmul."""
    return _sharded(list(slabs), batch, tuple(slabs[0].shape[1:]), compute, dist, gather)


def _bind_device(lib, device):
    """The library is one device + one stream per process: bind it to the GPU the operands live on
    before anything is launched (np_init is idempotent for the device it already holds).  Without
    this a rank r > 0 whose caller never called np_init(r) would run — and take its split-K scratch
    from — GPU 0."""
    from ._lib import check
    index = device.index if device.index is not None else 0
    check(lib.np_init(index))


def hip_elementwise(op: str):
    """compute() for GPU ranks: one np_binary (two operands) or np_unary (one operand) launch over
    the whole slab, ordered with torch's work like hip_compute."""
    def compute(*args):
        import torch

        from ._lib import BINARY_OPS, NP_FULL, UNARY_OPS, check, load
        lib = load()
        *ins, out = args
        n = out.numel()
        _bind_device(lib, out.device)
        handle = torch.cuda.current_stream(out.device).cuda_stream
        if handle:
            check(lib.np_set_stream(handle))
        else:
            torch.cuda.synchronize(out.device)
        if len(ins) == 2:
            quirk = op in ("multiply", "mod", "equal", "not_equal")   # as NDArray_*_Float sets them
            check(lib.np_binary(BINARY_OPS[op], ins[0].data_ptr(), NP_FULL, ins[1].data_ptr(), NP_FULL,
                                out.data_ptr(), 1, n, 1 if quirk else 0, lib.np_avx_body_end(n) if quirk else 0))
        else:
            check(lib.np_unary(UNARY_OPS[op], ins[0].data_ptr(), out.data_ptr(), n, 0.0, 0.0))
        if not handle:
            check(lib.np_sync())
    return compute


def hip_compute(a, b, out, whole=None):
    """compute() for GPU ranks: one np_sgemm_strided_batched launch, ordered with torch's work (whole: the operands are
    a piece of a slab of that many matrices — np_sgemm_strided_batched_piece).

    If torch's current stream is a real stream the library is switched onto it (RCCL then sees the
    GEMM in stream order).  The legacy default stream has handle 0, which np_set_stream reads as
    "library-owned stream": in that case the GEMM runs on the library stream between two explicit
    synchronisations."""
    import torch

    from ._lib import check, load
    lib = load()
    s, m, k = a.shape
    n = b.shape[2]
    _bind_device(lib, a.device)
    handle = torch.cuda.current_stream(a.device).cuda_stream
    if handle:
        check(lib.np_set_stream(handle))
    else:
        torch.cuda.synchronize(a.device)      # inputs produced on the null stream are complete
    if whole is not None and whole > s:
        check(lib.np_sgemm_strided_batched_piece(s, whole, m, n, k, a.data_ptr(), m * k, b.data_ptr(), k * n,
                                                 out.data_ptr(), m * n))
    else:
        check(lib.np_sgemm_strided_batched(s, m, n, k, a.data_ptr(), m * k, b.data_ptr(), k * n,
                                           out.data_ptr(), m * n))
    if not handle:
        check(lib.np_sync())                  # result complete before the collective is enqueued


hip_compute.accepts_whole = True
