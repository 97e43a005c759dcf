"""The multi-GPU path on the one GPU that is available: world size 1 over RCCL ("nccl" backend),
HIP strided-batched GEMM on torch's stream, in-place all-gather — the same code path that
bench.py --gpus N and numpower_amd.parallel take on every rank (the partition and the ragged
gather are covered with world sizes 2 and 3 on CPU/gloo in tests/test_parallel_cpu.py)."""
import os

import numpy as np
import pytest

from numpower_amd import synth

pytestmark = pytest.mark.gpu


def test_sharded_batched_matmul_world1_rccl(hip):
    import torch
    import torch.distributed as dist

    from numpower_amd import parallel

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        batch, m, k, n = 6, 256, 128, 384
        A = np.stack([synth.uniform((m, k), 100 + i, -1, 1) for i in range(batch)])
        B = np.stack([synth.uniform((k, n), 200 + i, -1, 1) for i in range(batch)])
        ref = np.einsum("bmk,bkn->bmn", A.astype(np.float64), B.astype(np.float64))
        scale = np.einsum("bmk,bkn->bmn", np.abs(A).astype(np.float64), np.abs(B).astype(np.float64))
        for use_side_stream in (False, True):
            ctx = torch.cuda.stream(torch.cuda.Stream()) if use_side_stream else torch.cuda.stream(None)
            with ctx:
                a = torch.from_numpy(A).cuda()
                b = torch.from_numpy(B).cuda()
                full = parallel.sharded_batched_matmul(a, b, batch, parallel.hip_compute, dist=dist, gather=True)
                mine = parallel.sharded_batched_matmul(a, b, batch, parallel.hip_compute, dist=dist, gather=False)
                torch.cuda.synchronize()
            for got in (full.cpu().numpy(), mine.cpu().numpy()):
                assert (np.abs(got - ref) <= 1e-6 * scale).all()
            # per-slice elementwise on the same plumbing: HIP np_binary / np_unary over the slab
            with ctx:
                prod = parallel.sharded_elementwise([a, a], batch, parallel.hip_elementwise("multiply"), dist=dist)
                ex = parallel.sharded_elementwise([a], batch, parallel.hip_elementwise("exp"), dist=dist, gather=True)
                torch.cuda.synchronize()
            assert np.array_equal(prod.cpu().numpy(), A * A)
            assert np.allclose(ex.cpu().numpy(), np.exp(A.astype(np.float64)), rtol=1e-6)
    finally:
        dist.destroy_process_group()
        from numpower_amd._lib import check, load
        check(load().np_set_stream(None))   # back to the library-owned stream for the other tests
