"""Shapes at the edges of the launch geometry: more slices than a grid dimension holds (65535), one very long extent next to tiny
ones, tiny extents everywhere.  Every case against numpy (fp64 accumulation where it matters), through the C ABI — the shapes a
random sweep over 'reasonable' sizes (tests/test_gpu_fuzz.py) does not reach."""
import ctypes as C

import numpy as np
import pytest

from numpower_amd import synth

pytestmark = pytest.mark.gpu


def _close(got, want, scale=None, tol=1e-5):
    want = np.asarray(want, dtype=np.float64)
    scale = np.abs(want) if scale is None else np.asarray(scale, dtype=np.float64)
    return bool((np.abs(np.asarray(got, dtype=np.float64) - want) <= tol * scale + 1e-30).all())


@pytest.mark.parametrize("shape,axis", [
    ((100000, 50, 8), 1), ((70000, 3, 130), 1), ((66000, 7, 4), 1), ((200000, 40, 1), 1), ((1, 70000, 1), 1),
    ((70001, 1, 5), 1), ((3, 70000, 3), 1), ((131075, 2, 2), 1), ((2, 3, 300000), 1), ((300000, 3, 2), 0), ((70000, 2, 33), 2),
    ((1, 1, 1), 1), ((65536, 65, 1), 1), ((65537, 4, 64), 1),
])
def test_axis_reductions(hip, shape, axis):
    D = hip
    lib = __import__("numpower_amd._lib", fromlist=["load"]).load()
    x = synth.uniform(shape, 901 + sum(shape) % 97, -1.0, 1.0)
    dx = D.DeviceArray.from_host(x)
    outer = int(np.prod(shape[:axis], dtype=np.int64))
    inner = int(np.prod(shape[axis + 1:], dtype=np.int64))
    out = D.DeviceArray((outer * inner,))
    x64 = x.astype(np.float64)
    for code, fn in ((0, np.sum), (3, np.max), (2, np.min), (4, np.mean)):
        assert lib.np_reduce_axis(code, dx.ptr, outer, shape[axis], inner, out.ptr, 0) == 0
        got = out.to_host()
        want = fn(x64, axis=axis).reshape(-1)
        if code in (2, 3):
            assert (got == want.astype(np.float32)).all(), (shape, axis, code)
        else:
            assert _close(got, want, np.abs(x64).sum(axis=axis).reshape(-1) / (shape[axis] if code == 4 else 1)), (shape, axis, code)
    idx = D.DeviceArray((outer * inner,))
    for is_max, fn in ((1, np.argmax), (0, np.argmin)):
        assert lib.np_argreduce(is_max, dx.ptr, outer, shape[axis], inner, idx.ptr) == 0
        assert (idx.to_host() == fn(x, axis=axis).reshape(-1).astype(np.float32)).all(), (shape, axis, is_max)
    for d in (dx, out, idx):
        d.free()


@pytest.mark.parametrize("batch,rows,cols", [
    (1, 20_000_000, 3), (1, 3, 20_000_000), (65535, 5, 7), (40000, 33, 1), (1, 1, 1), (1, 70000, 70), (3, 1, 100000), (2, 100000, 1),
    (1, 9_000_000, 17), (1, 17, 9_000_000), (300, 129, 127),
])
def test_transposes(hip, batch, rows, cols):
    D = hip
    lib = __import__("numpower_amd._lib", fromlist=["load"]).load()
    x = synth.uniform((batch, rows, cols), 77 + rows % 13, -1.0, 1.0)
    dx, dt = D.DeviceArray.from_host(x), D.DeviceArray((batch, cols, rows))
    assert lib.np_transpose2d(dx.ptr, dt.ptr, batch, rows, cols) == 0
    assert (dt.to_host().view(np.uint32) == np.ascontiguousarray(x.transpose(0, 2, 1)).view(np.uint32)).all()
    assert lib.np_permute(dx.ptr, dt.ptr, 3, (C.c_int * 3)(batch, rows, cols), (C.c_int * 3)(0, 2, 1)) == 0
    assert (dt.to_host().view(np.uint32) == np.ascontiguousarray(x.transpose(0, 2, 1)).view(np.uint32)).all()
    dx.free()
    dt.free()


@pytest.mark.parametrize("shape,perm", [
    ((70000, 2, 3, 5), (3, 1, 2, 0)), ((2, 70000, 3, 2), (0, 2, 1, 3)), ((3, 2, 70001), (2, 0, 1)), ((100000, 3, 1, 2, 1, 2), (5, 4, 3, 2, 1, 0)),
    ((66000, 4, 4), (1, 0, 2)), ((4, 66000, 4), (1, 0, 2)), ((2, 2, 2, 2, 2, 2, 2, 70000), (7, 0, 1, 2, 3, 4, 5, 6)),
])
def test_permutes(hip, shape, perm):
    D = hip
    lib = __import__("numpower_amd._lib", fromlist=["load"]).load()
    x = synth.uniform(shape, 31 + len(shape), -1.0, 1.0)
    dx, dt = D.DeviceArray.from_host(x), D.DeviceArray((x.size,))
    nd = len(shape)
    assert lib.np_permute(dx.ptr, dt.ptr, nd, (C.c_int * nd)(*shape), (C.c_int * nd)(*perm)) == 0
    assert (dt.to_host().view(np.uint32) == np.ascontiguousarray(x.transpose(perm)).reshape(-1).view(np.uint32)).all()
    dx.free()
    dt.free()


@pytest.mark.parametrize("m,n,k", [
    (20_000_000, 16, 16), (16, 20_000_000, 16), (9_000_001, 3, 5), (3, 9_000_001, 5), (1, 1, 1), (1, 5_000_000, 1), (5_000_000, 1, 1),
    (70000, 70, 9), (70, 70000, 9), (1, 1, 3_000_000), (2, 70000, 2), (1_100_000, 130, 2),
])
def test_products(hip, m, n, k):
    D = hip
    lib = __import__("numpower_amd._lib", fromlist=["load"]).load()
    a = synth.uniform((m, k), 5 + m % 11, -1.0, 1.0)
    b = synth.uniform((k, n), 6 + n % 11, -1.0, 1.0)
    da, db, dc = D.DeviceArray.from_host(a), D.DeviceArray.from_host(b), D.DeviceArray((m, n))
    assert lib.np_sgemm(m, n, k, da.ptr, db.ptr, dc.ptr) == 0, lib.np_last_error()
    want = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    assert _close(dc.to_host(), want, scale)
    if n == 1:
        y = D.DeviceArray((m,))
        assert lib.np_sgemv(m, k, da.ptr, db.ptr, y.ptr) == 0
        assert _close(y.to_host(), want.reshape(-1), scale.reshape(-1))
        y.free()
    for d in (da, db, dc):
        d.free()


@pytest.mark.parametrize("m,n", [(50_000_000, 2), (2, 50_000_000), (100_000_000, 1), (1, 100_000_000), (70000, 1000), (1, 1), (7, 3)])
def test_matrix_vector(hip, m, n):
    D = hip
    lib = __import__("numpower_amd._lib", fromlist=["load"]).load()
    a = synth.uniform((m, n), 15 + m % 7, -1.0, 1.0)
    x = synth.uniform((n,), 16, -1.0, 1.0)
    da, dx, dy = D.DeviceArray.from_host(a), D.DeviceArray.from_host(x), D.DeviceArray((m,))
    assert lib.np_sgemv(m, n, da.ptr, dx.ptr, dy.ptr) == 0
    want = a.astype(np.float64) @ x.astype(np.float64)
    assert _close(dy.to_host(), want, np.abs(a).astype(np.float64) @ np.abs(x).astype(np.float64))
    for d in (da, dx, dy):
        d.free()


@pytest.mark.parametrize("rows,cols", [(20_000_000, 4), (4, 20_000_000), (70000, 12), (1, 100), (100, 1), (66000, 1000), (3, 100_000_001)])
def test_broadcasts_and_chain_axis_sums(hip, rows, cols):
    D = hip
    from numpower_amd import _lib
    from numpower_amd._lib import BINARY_OPS, UNARY_OPS, FusedOp
    lib = _lib.load()
    x = synth.uniform((rows, cols), 21 + rows % 5, -1.0, 1.0)
    r = synth.uniform((cols,), 22, -1.0, 1.0)
    c = synth.uniform((rows,), 23, -1.0, 1.0)
    dx, dr, dc, out = D.DeviceArray.from_host(x), D.DeviceArray.from_host(r), D.DeviceArray.from_host(c), D.DeviceArray((rows, cols))
    D.binary("add", dx, "full", dr, "row", rows, cols, out=out)
    assert (out.to_host().view(np.uint32) == (x + r[None, :]).view(np.uint32)).all()
    D.binary("multiply", dc, "col", dx, "full", rows, cols, out=out)
    assert (out.to_host() == (c[:, None] * x)).all()
    # sum(|X| + row, axis) as one launch, both axes
    ops = (FusedOp * 2)(FusedOp(0, UNARY_OPS["abs"], 0, 0, 0, 0, 0, 0), FusedOp(1, BINARY_OPS["add"], 1, 0, 0, 0, 0, 0))
    ptrs = (C.c_void_p * 2)(dx.ptr, dr.ptr)
    kinds = (C.c_int * 2)(0, 2)
    want = np.abs(x).astype(np.float64) + r.astype(np.float64)[None, :]
    o1, o0 = D.DeviceArray((rows,)), D.DeviceArray((cols,))
    assert lib.np_fused_chain_reduce_axis(ptrs, kinds, 2, ops, 2, 0, rows, cols, 1, o1.ptr) == 0
    assert _close(o1.to_host(), want.sum(1), np.abs(want).sum(1))
    assert lib.np_fused_chain_reduce_axis(ptrs, kinds, 2, ops, 2, 0, rows, cols, 0, o0.ptr) == 0
    assert _close(o0.to_host(), want.sum(0), np.abs(want).sum(0))
    for d in (dx, dr, dc, out, o1, o0):
        d.free()
