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


def test_bad_arguments(hip):
    lib = load()
    for args in ((1, 1, b"tcp://127.0.0.1:1"), (-1, 2, b"x"), (0, 0, b"x"), (0, 1, b"")):
        with pytest.raises(NumPowerError):
            check(lib.np_comm_init(*args))
