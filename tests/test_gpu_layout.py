"""SURVEY.md §8(f) row 3 on the GPU: transpose / axis permutation (src/manipulation.c:68-130),
bit-exact against the oracle's strided copy and numpy, at sizes far beyond the 256 x 256 limit of
the reference's own GPU kernel (cuda_math.cu:1288-1294)."""
import numpy as np
import pytest

from numpower_amd import synth

pytestmark = pytest.mark.gpu


def _bits(x):
    return np.asarray(x, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("shape", [(2, 2), (2, 3), (64, 64), (257, 1001), (1000, 1000), (4096, 64), (1, 777), (5000, 3)])
def test_transpose_2d(shape, hip, oracle):
    from numpower_amd.ndarray import NDArray
    x = synth.uniform(shape, 71, -1.0, 1.0)
    got = NDArray.transpose(NDArray.array(x).gpu())
    assert got.shape() == [shape[1], shape[0]]
    g = got.cpu().numpy()
    assert (_bits(g) == _bits(x.T)).all()
    if x.size <= 300000:
        assert (_bits(g) == _bits(oracle.transpose(x))).all()


@pytest.mark.parametrize("shape,axes", [((3, 4, 5), None), ((3, 4, 5), (0, 2, 1)), ((3, 4, 5), (1, 0, 2)),
                                        ((6, 7, 8, 9), (3, 1, 0, 2)), ((2, 130, 70), (0, 2, 1)),
                                        ((4, 4), (0, 1)), ((7,), None), ((1, 1, 4), None), ((3, 4, 5), (-1, 0, 1))])
def test_permute_nd(shape, axes, hip, oracle):
    from numpower_amd.ndarray import NDArray
    x = synth.uniform(shape, 72, -1.0, 1.0)
    got = NDArray.transpose(NDArray.array(x).gpu(), axes)
    want = np.transpose(x, axes)
    g = got.cpu().numpy() if not isinstance(got, float) else np.float32(got)
    assert list(g.shape) == list(want.shape)
    assert (_bits(g) == _bits(np.ascontiguousarray(want))).all()
    assert (_bits(g) == _bits(oracle.transpose(x, axes))).all()


def test_transpose_errors_and_roundtrip(hip):
    from numpower_amd.ndarray import Error, NDArray
    x = synth.uniform((33, 65), 5, -1, 1)
    g = NDArray.array(x).gpu()
    with pytest.raises(Error, match="axes don't match array"):
        NDArray.transpose(g, (0,))
    with pytest.raises(Error, match="repeated axis in transpose"):
        NDArray.transpose(g, (1, 1))
    back = NDArray.transpose(NDArray.transpose(g)).cpu().numpy()
    assert (_bits(back) == _bits(x)).all()                      # involution
    # (A.B)^T == B^T.A^T through the GEMM, within GEMM tolerance
    A = synth.uniform((130, 70), 1, -1, 1)
    B = synth.uniform((70, 200), 2, -1, 1)
    gA, gB = NDArray.array(A).gpu(), NDArray.array(B).gpu()
    left = NDArray.transpose(NDArray.matmul(gA, gB)).cpu().numpy()
    right = NDArray.matmul(NDArray.transpose(gB), NDArray.transpose(gA)).cpu().numpy()
    scale = (np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)).T
    assert (np.abs(left.astype(np.float64) - right) <= 1e-6 * scale).all()


def test_transpose_16384x4096(hip):
    D = hip
    rows, cols = 16384, 4096
    x = synth.uniform((rows, cols), 73, 0.0, 1.0)
    import ctypes as C
    from numpower_amd._lib import check, load
    dx = D.DeviceArray.from_host(x)
    out = D.DeviceArray((cols, rows))
    check(load().np_transpose2d(dx.ptr, out.ptr, 1, rows, cols))
    assert (_bits(out.to_host()) == _bits(np.ascontiguousarray(x.T))).all()
    dx.free()
    out.free()
    del C


PERMS = [((30, 3, 64, 65), (0, 2, 3, 1)),        # NCHW -> NHWC: fuses to a batched 3 x 4160 transpose (skinny path)
         ((30, 64, 65, 3), (0, 3, 1, 2)),        # NHWC -> NCHW
         ((70, 130, 66), (2, 1, 0)),             # full reversal: LDS-tiled plane transpose, batch = middle axis
         ((9, 10, 11, 12), (3, 2, 1, 0)),
         ((5, 1, 7, 1, 9), (4, 3, 2, 1, 0)),     # extent-1 axes are dropped before dispatch
         ((2, 3, 4, 5, 6, 7), (5, 0, 1, 2, 3, 4)),   # rotate: fuses to (720, 7) -> (7, 720)
         ((2, 3, 4, 5, 6, 7), (1, 0, 3, 2, 5, 4)),
         ((129, 257, 3), (1, 0, 2)),             # innermost axis kept: gather path
         ((4, 5, 6, 7, 8, 3, 2, 2), (7, 6, 5, 4, 3, 2, 1, 0)),
         ((64, 64, 64), (1, 2, 0))]


@pytest.mark.parametrize("shape,axes", PERMS)
def test_permute_paths(shape, axes, hip):
    """np_permute after axis fusion: every dispatch (copy, batched 2-D transpose incl. the skinny
    kernel, LDS-tiled plane transpose, gather) against numpy, bit for bit."""
    from numpower_amd.ndarray import NDArray
    x = synth.uniform(shape, 73, -1.0, 1.0)
    got = NDArray.transpose(NDArray.array(x).gpu(), axes).cpu().numpy()
    want = np.ascontiguousarray(np.transpose(x, axes))
    assert got.shape == want.shape and (_bits(got) == _bits(want)).all()


@pytest.mark.parametrize("rc", [(100_003, 3), (3, 100_003), (70_000, 1), (1, 70_000), (5000, 16), (16, 5000), (1031, 1033),
                                (4099, 257), (2, 2), (17, 17),
                                (4099, 4099), (4100, 4243), (5000, 4099)])   # write-aligned tiles: 64 x 64 below 1200 tiles of 128 (round 5), 128 x 128 above
def test_transpose2d_skinny_and_odd(rc, hip):
    """Tall-skinny / short-wide matrices (one side <= 16: transpose_skinny_kernel; these also have
    more tile rows than gridDim.y allows) and odd sizes (dword-aligned float4 path)."""
    from numpower_amd.ndarray import NDArray
    x = synth.uniform(rc, 74, -1.0, 1.0)
    got = NDArray.transpose(NDArray.array(x).gpu()).cpu().numpy()
    assert got.shape == (rc[1], rc[0]) and (_bits(got) == _bits(np.ascontiguousarray(x.T))).all()


@pytest.mark.parametrize("batch,planes,n", [(1, 3, 100_004), (5, 3, 4096), (30, 3, 64 * 65 * 4), (2, 2, 2048), (3, 4, 2052), (1, 2, 1_000_000),
                                            (7, 4, 6148), (1, 3, 1024), (2, 3, 1028)])
def test_interleave_few_planes_both_ways(batch, planes, n, hip):
    """Two to four long rows / columns (NCHW <-> NHWC with C <= 4): the float4 interleave kernels of round 5, every tile
    boundary (2048 positions), ragged last tiles, batches — bit for bit against numpy, and against the skinny kernel they
    replace (np_layout_set_variant(5))."""
    import ctypes as C
    from numpower_amd._lib import check, load
    lib = load()
    x = synth.uniform((batch, planes, n), 75, -1.0, 1.0)
    dx = hip.DeviceArray.from_host(x)
    dy, dz, dw = hip.DeviceArray((batch, n, planes)), hip.DeviceArray((batch, planes, n)), hip.DeviceArray((batch, n, planes))
    check(lib.np_transpose2d(dx.ptr, dy.ptr, batch, planes, n))                 # planes -> interleaved
    y = dy.to_host()
    assert (_bits(y) == _bits(np.ascontiguousarray(x.transpose(0, 2, 1)))).all()
    check(lib.np_transpose2d(dy.ptr, dz.ptr, batch, n, planes))                 # and back
    assert (_bits(dz.to_host()) == _bits(x)).all()
    try:
        check(lib.np_layout_set_variant(5))
        check(lib.np_transpose2d(dx.ptr, dw.ptr, batch, planes, n))
    finally:
        check(lib.np_layout_set_variant(0))
    assert (_bits(dw.to_host()) == _bits(y)).all()
    # a view that starts 4 bytes into the buffer is not float4-loadable: the same call must still be right (skinny kernel)
    if n >= 8:
        m = n - 4
        sub = dx.view(1, (planes, m))
        out = hip.DeviceArray((m, planes))
        check(lib.np_transpose2d(sub.ptr, out.ptr, 1, planes, m))
        want = x.reshape(-1)[1:1 + planes * m].reshape(planes, m).T
        assert (_bits(out.to_host()) == _bits(np.ascontiguousarray(want))).all()
        out.free()
    for d in (dx, dy, dz, dw):
        d.free()
    del C


@pytest.mark.parametrize("shape,axes", [((6, 40, 300, 8), (0, 2, 1, 3)), ((3, 33, 65, 4), (0, 2, 1, 3)), ((5, 17, 19, 12), (2, 1, 0, 3)),
                                        ((2, 64, 64, 16), (2, 0, 1, 3)), ((4, 50, 70, 28), (0, 2, 1, 3)), ((7, 9, 100, 8), (1, 2, 0, 3))])
def test_permute_float4_elements(shape, axes, hip):
    """A kept innermost axis of 4, 8, 12 ... floats (NHWC-like): the plane permute moves whole float4s (round 5); against numpy
    and against the single-float form of the same kernel (np_layout_set_variant(6)), bit for bit."""
    import ctypes as C
    from numpower_amd._lib import check, load
    lib = load()
    x = synth.uniform(shape, 76, -1.0, 1.0)
    want = np.ascontiguousarray(np.transpose(x, axes))
    dx, dy = hip.DeviceArray.from_host(x), hip.DeviceArray(want.shape)
    sh, pm = (C.c_int * 4)(*shape), (C.c_int * 4)(*axes)
    for variant in (0, 6):
        hip.fill(dy, float("nan"))
        try:
            check(lib.np_layout_set_variant(variant))
            check(lib.np_permute(dx.ptr, dy.ptr, 4, sh, pm))
        finally:
            check(lib.np_layout_set_variant(0))
        assert (_bits(dy.to_host()) == _bits(want)).all(), variant
    dx.free()
    dy.free()
