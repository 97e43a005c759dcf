"""np_graph_*: a sequence of library calls captured into a HIP graph replays to the same bits as the
eager calls, on new input contents, and the pool behaves under capture."""
import ctypes as C

import numpy as np
import pytest

from numpower_amd import synth

pytestmark = pytest.mark.gpu


def test_captured_chain_replays_bit_identically(hip):
    from numpower_amd import _lib
    from numpower_amd import device as D
    from numpower_amd._lib import BINARY_OPS, UNARY_OPS
    lib = _lib.load()
    n = 1000 * 1000
    a, b = D.DeviceArray((n,)), D.DeviceArray((n,))
    t1, t2, out = D.DeviceArray((n,)), D.DeviceArray((n,)), D.DeviceArray((1000, 1))
    x = D.DeviceArray((1000, 64)); y = D.DeviceArray((64, 32)); z = D.DeviceArray((1000, 32))

    def sequence():
        _lib.check(lib.np_binary(BINARY_OPS["multiply"], a.ptr, 0, b.ptr, 0, t1.ptr, 1, n, 1, n // 8 * 8))
        _lib.check(lib.np_unary(UNARY_OPS["exp"], t1.ptr, t2.ptr, n, 0.0, 0.0))
        _lib.check(lib.np_binary(BINARY_OPS["add"], t2.ptr, 0, a.ptr, 0, t1.ptr, 1, n, 0, 0))
        _lib.check(lib.np_reduce_axis(0, t1.ptr, 1000, 1000, 1, out.ptr, 0))     # row sums, result stays on the device
        _lib.check(lib.np_sgemm(1000, 32, 64, x.ptr, y.ptr, z.ptr))

    def load(seed):
        for buf, s, shape in ((a, seed, (n,)), (b, seed + 1, (n,)), (x, seed + 2, (1000, 64)), (y, seed + 3, (64, 32))):
            h = synth.uniform(shape, s, -1.0, 1.0)
            _lib.check(lib.np_memcpy_h2d(buf.ptr, h.ctypes.data, h.nbytes))

    load(1)
    sequence()                      # warm-up: the pool now owns every scratch block the sequence needs
    _lib.check(lib.np_sync())
    _lib.check(lib.np_graph_begin())
    sequence()
    g = C.c_void_p()
    _lib.check(lib.np_graph_end(C.byref(g)))
    for seed in (10, 20):
        load(seed)
        sequence()
        eager = (t1.to_host().copy(), out.to_host().copy(), z.to_host().copy())
        for buf in (t1, t2, out, z):
            D.fill(buf, 0.0)
        _lib.check(lib.np_graph_launch(g))
        replay = (t1.to_host(), out.to_host(), z.to_host())
        for e, r in zip(eager, replay):
            assert (e.view(np.uint32) == r.view(np.uint32)).all()
    _lib.check(lib.np_graph_destroy(g))


@pytest.mark.parametrize("shape,variant", [((512, 512, 4096), -999), ((1024, 1024, 1024), -999), ((512, 512, 2048), -1204), ((1536, 1536, 1536), -4)],
                         ids=["planner's split plan", "64x64 LDS-DMA tiles", "64x64 tiles, K split 4 ways in-launch", "stream-K"])
def test_captured_matmul_with_an_in_launch_fold_replays(shape, variant, hip):
    """The GEMM forms that fold partial tiles inside one launch (sgemm_dmas_kernel's distributed split-K fold, stream-K) borrow
    tickets / flags that the launch itself puts back to zero, and a pooled workspace: captured once, they must replay — twice
    in a row, on new operand contents — to the bits of the eager call."""
    from numpower_amd import _lib
    from numpower_amd import device as D
    lib = _lib.load()
    m, n, k = shape
    a, b, c = D.DeviceArray((m, k)), D.DeviceArray((k, n)), D.DeviceArray((m, n))

    def load(seed):
        for buf, s, sh in ((a, seed, (m, k)), (b, seed + 1, (k, n))):
            h = synth.uniform(sh, s, -1.0, 1.0)
            _lib.check(lib.np_memcpy_h2d(buf.ptr, h.ctypes.data, h.nbytes))

    _lib.check(lib.np_sgemm_set_variant(variant))
    try:
        load(1)
        _lib.check(lib.np_sgemm(m, n, k, a.ptr, b.ptr, c.ptr))      # warm-up: the pool owns the workspace now
        _lib.check(lib.np_sync())
        _lib.check(lib.np_graph_begin())
        _lib.check(lib.np_sgemm(m, n, k, a.ptr, b.ptr, c.ptr))
        g = C.c_void_p()
        _lib.check(lib.np_graph_end(C.byref(g)))
        for seed in (10, 20):
            load(seed)
            _lib.check(lib.np_sgemm(m, n, k, a.ptr, b.ptr, c.ptr))
            eager = c.to_host().copy()
            for _ in range(2):
                D.fill(c, float("nan"))
                _lib.check(lib.np_graph_launch(g))
                assert (c.to_host().view(np.uint32) == eager.view(np.uint32)).all(), (shape, variant, seed)
        _lib.check(lib.np_graph_destroy(g))
    finally:
        _lib.check(lib.np_sgemm_set_variant(-999))
        _lib.check(lib.np_sgemm_set_variant(-2))
    assert lib.np_sync() == 0, lib.np_last_error()
