"""Run-to-run bit identity at BASELINE's sizes.  The reference's GPU reductions use float atomics (cuda_math.cu:803,777: the order of the
additions changes from run to run); every reduction here folds its partials in an order fixed by the launch geometry — whichever
workgroup arrives last, whatever the scheduler did.  Each call is repeated and the bits of the results compared; sizes are the ones
where a last-workgroup fold, a split-K fold or a stream-K fix-up is in play."""
import ctypes as C

import numpy as np
import pytest

from numpower_amd import synth

pytestmark = pytest.mark.gpu
REPEATS = 6


def _same(runs):
    first = np.asarray(runs[0], dtype=np.float32).view(np.uint32)
    return all((np.asarray(r, dtype=np.float32).view(np.uint32) == first).all() for r in runs[1:])


def test_reductions_and_statistics_repeat_bit_for_bit(hip):
    from numpower_amd import _lib
    from numpower_amd._lib import BINARY_OPS, UNARY_OPS, FusedOp
    lib = _lib.load()
    D = hip
    for n in (100_000_000, 1_000_003, 999_999, 65_537, 4099):
        x = synth.uniform((n,), 41, -1.0, 1.0)
        w = synth.uniform((n,), 42, 0.5, 1.5)
        dx, dw = D.DeviceArray.from_host(x), D.DeviceArray.from_host(w)
        host, h2 = C.c_float(), C.c_float()
        for op in range(5):
            runs = []
            for _ in range(REPEATS):
                _lib.check(lib.np_reduce_all(op, dx.ptr, n, C.byref(host)))
                runs.append([host.value])
            assert _same(runs), ("reduce_all", op, n)
        runs = []
        for _ in range(REPEATS):
            _lib.check(lib.np_moments(dx.ptr, n, C.byref(host), C.byref(h2)))
            runs.append([host.value, h2.value])
        assert _same(runs), ("moments", n)
        runs = []
        for _ in range(REPEATS):
            _lib.check(lib.np_weighted_sums(dx.ptr, dw.ptr, n, C.byref(host), C.byref(h2)))
            runs.append([host.value, h2.value])
        assert _same(runs), ("weighted_sums", n)
        ops = (FusedOp * 2)(FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0), FusedOp(1, BINARY_OPS["multiply"], 1, 0, 0, 0, 0, 0))
        ptrs = (C.c_void_p * 2)(dx.ptr, dw.ptr)
        kinds = (C.c_int * 2)(0, 0)
        for rop in (0, 3, 4):
            runs = []
            for _ in range(REPEATS):
                _lib.check(lib.np_fused_chain_reduce(ptrs, kinds, 2, ops, 2, rop, 1, n, C.byref(host)))
                runs.append([host.value])
            assert _same(runs), ("chain reduce", rop, n)
        idx = D.DeviceArray((1,))
        runs = []
        for _ in range(REPEATS):
            _lib.check(lib.np_argreduce(1, dx.ptr, 1, n, 1, idx.ptr))
            runs.append(idx.to_host())
        assert _same(runs), ("argmax", n)
        for d in (dx, dw, idx):
            d.free()


@pytest.mark.parametrize("rows,cols", [(65536, 4096), (25000, 4000), (9973, 9973), (500000, 200), (30_000_000, 3), (3, 30_000_000)])
def test_axis_reductions_repeat_bit_for_bit(hip, rows, cols):
    from numpower_amd import _lib
    from numpower_amd._lib import UNARY_OPS, FusedOp
    lib = _lib.load()
    D = hip
    x = synth.uniform((rows, cols), 43, -1.0, 1.0)
    dx = D.DeviceArray.from_host(x)
    o0, o1 = D.DeviceArray((cols,)), D.DeviceArray((rows,))
    ops = (FusedOp * 1)(FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0))
    ptrs = (C.c_void_p * 1)(dx.ptr)
    kinds = (C.c_int * 1)(0)
    for what, call, out in (
            ("sum axis 0", lambda: lib.np_reduce_axis(0, dx.ptr, 1, rows, cols, o0.ptr, 0), o0),
            ("sum axis 1", lambda: lib.np_reduce_axis(0, dx.ptr, rows, cols, 1, o1.ptr, 0), o1),
            ("mean axis 0", lambda: lib.np_reduce_axis(4, dx.ptr, 1, rows, cols, o0.ptr, 0), o0),
            ("prod axis 1", lambda: lib.np_reduce_axis(1, dx.ptr, rows, cols, 1, o1.ptr, 0), o1),
            ("sum(exp) axis 0", lambda: lib.np_fused_chain_reduce_axis(ptrs, kinds, 1, ops, 1, 0, rows, cols, 0, o0.ptr), o0),
            ("sum(exp) axis 1", lambda: lib.np_fused_chain_reduce_axis(ptrs, kinds, 1, ops, 1, 0, rows, cols, 1, o1.ptr), o1)):
        runs = []
        for _ in range(REPEATS):
            assert call() == 0, what
            runs.append(out.to_host())
        assert _same(runs), (what, rows, cols)
    for d in (dx, o0, o1):
        d.free()


@pytest.mark.parametrize("m,n,k", [(4096, 4096, 4096), (2560, 2560, 2560), (3000, 3000, 3000), (1024, 1024, 1024), (768, 768, 768), (100, 100, 100000),
                                   (256, 256, 32768), (4096, 4096, 256), (4097, 4097, 4097), (64, 64, 4_000_000), (1, 4096, 4096), (5, 7, 100003)])
def test_products_repeat_bit_for_bit(hip, m, n, k):
    from numpower_amd import _lib
    lib = _lib.load()
    D = hip
    a = synth.uniform((m, k), 44, -1.0, 1.0)
    b = synth.uniform((k, n), 45, -1.0, 1.0)
    da, db, dc = D.DeviceArray.from_host(a), D.DeviceArray.from_host(b), D.DeviceArray((m, n))
    runs = []
    for _ in range(REPEATS):
        _lib.check(lib.np_memset0(dc.ptr, 4 * m * n))
        _lib.check(lib.np_sgemm(m, n, k, da.ptr, db.ptr, dc.ptr))
        runs.append(dc.to_host())
    assert _same(runs), (m, n, k)
    y = D.DeviceArray((m,))
    x = D.DeviceArray.from_host(b[:, 0].copy())
    runs = []
    for _ in range(REPEATS):
        _lib.check(lib.np_sgemv(m, k, da.ptr, x.ptr, y.ptr))
        runs.append(y.to_host())
    assert _same(runs), ("sgemv", m, k)
    for d in (da, db, dc, x, y):
        d.free()
