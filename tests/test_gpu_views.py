"""SURVEY.md §8(f) rows 1 and 3 on the GPU, the plumbing either side of the kernels: ArrayEqual /
AllClose as one streaming reduction, Diagonal / Trace / Slice / ToContiguous as one strided gather,
Reshape / Flatten / ExpandDim / Append as metadata + device copies.  Index work is bit-exact
against the oracle restatement (oracle/oracle.py) and numpy."""
import ctypes as C

import numpy as np
import pytest

from numpower_amd import synth

pytestmark = pytest.mark.gpu


def _nd():
    from numpower_amd.ndarray import NDArray
    return NDArray


def _bits_equal(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return a.shape == b.shape and (a.view(np.uint32) == b.view(np.uint32)).all()


@pytest.mark.parametrize("n", [1, 7, 8, 1000, 1_000_003])
def test_array_equal_and_allclose(n, hip, oracle):
    nd = _nd()
    a = synth.uniform((n,), 61, -3.0, 3.0)
    ga = nd.array(a).gpu()
    assert nd.array_equal(ga, nd.array(a.copy()).gpu()) is True
    assert oracle.array_equal(a, a.copy()) == 1
    for pos in sorted({0, n // 2, n - 1}):
        b = a.copy()
        b[pos] = np.nextafter(b[pos], np.float32(10.0))            # one ulp off in one place
        gb = nd.array(b).gpu()
        assert nd.array_equal(ga, gb) is False and oracle.array_equal(a, b) == 0
        assert nd.allclose(ga, gb) is True and oracle.allclose(a, b) == 1
        b[pos] = a[pos] + np.float32(0.5)
        assert nd.allclose(ga, nd.array(b).gpu()) is False and oracle.allclose(a, b) == 0
    # shape mismatch: ArrayEqual -> 0, AllClose -> error
    assert nd.array_equal(ga, nd.array(np.zeros((n, 1), np.float32)).gpu()) is False
    from numpower_amd.ndarray import Error
    with pytest.raises(Error, match="Shape mismatch"):
        nd.allclose(ga, nd.array(np.zeros((n + 1,), np.float32)).gpu())


def test_equality_nan_and_tolerance_rules(hip, oracle):
    nd = _nd()
    a = np.float32([1.0, np.nan, 3.0, 0.0, -0.0, 1e-9, 100.0, 5.0, 6.0])
    ga = nd.array(a).gpu()
    # NaN != NaN (compare_ndarrays, logic.c:686-690); -0.0 == 0.0
    assert nd.array_equal(ga, nd.array(a.copy()).gpu()) is False and oracle.array_equal(a, a.copy()) == 0
    z = np.float32([0.0, -0.0, 5.0])
    assert nd.array_equal(nd.array(z).gpu(), nd.array(np.float32([-0.0, 0.0, 5.0])).gpu()) is True
    # allclose: a NaN on either side compares false, i.e. passes (float_allclose, logic.c:730-733)
    assert nd.allclose(ga, nd.array(a.copy()).gpu()) is True and oracle.allclose(a, a.copy()) == 1
    # |a-b| <= atol + rtol*|b| is asymmetric in b and inclusive
    x, y = np.float32([100.0]), np.float32([100.001])
    for rtol, atol in ((1e-5, 1e-8), (1e-6, 0.0), (0.0, 1e-3), (0.0, 9.9e-4)):
        want = oracle.allclose(x, y, rtol, atol)
        assert nd.allclose(nd.array(x).gpu(), nd.array(y).gpu(), rtol, atol) is bool(want)
    from numpower_amd.ndarray import Error
    with pytest.raises(Error, match="same device"):
        nd.allclose(ga, nd.array(a.copy()))


@pytest.mark.parametrize("shape", [(4, 4), (7, 3), (3, 7), (1000, 1000), (2049, 513)])
def test_diagonal_and_trace(shape, hip, oracle):
    nd = _nd()
    x = synth.uniform(shape, 62, -1.0, 1.0)
    g = nd.array(x).gpu()
    d = nd.diagonal(g)
    assert d.isGPU() and _bits_equal(d.cpu().numpy(), oracle.diagonal(x))
    assert _bits_equal(d.cpu().numpy(), np.diagonal(x))
    t = nd.trace(g)
    assert isinstance(t, float)
    want = float(np.diagonal(x).astype(np.float64).sum())
    assert abs(t - want) <= 1e-5 * max(1.0, np.abs(np.diagonal(x)).sum())
    from numpower_amd.ndarray import Error
    with pytest.raises(Error, match="Array must be 2-d"):
        nd.diagonal(nd.array(np.ones((4,), np.float32)).gpu())


def test_reshape_is_a_view_flatten_is_a_copy(hip, oracle):
    nd = _nd()
    x = synth.uniform((6, 8), 63, 0.0, 1.0)
    g = nd.array(x).gpu()
    before = nd.live_device_allocations()
    r = nd.reshape(g, [4, 12])
    assert r.shape() == [4, 12] and nd.live_device_allocations() == before      # no new buffer
    assert _bits_equal(r.cpu().numpy(), oracle.reshape(x, [4, 12]))
    g.fill(2.0)                                                                  # the view sees the write
    assert (r.cpu().numpy() == 2.0).all()
    f = nd.flatten(g)
    assert f.shape() == [48] and nd.live_device_allocations() == before + 1
    g.fill(3.0)
    assert (f.cpu().numpy() == 2.0).all()
    # the view keeps its base alive
    del g
    assert (r.cpu().numpy() == 3.0).all()
    from numpower_amd.ndarray import Error
    with pytest.raises(Error, match="incompatible shape in reshape call."):
        nd.reshape(r, [5, 5])
    # results of views feed the kernels like any other array
    assert _bits_equal((r + 1.0).cpu().numpy(), np.full((4, 12), 4.0, np.float32))


@pytest.mark.parametrize("axis", [0, -1, 1, [0, -1], [2, 1, 0], [0, -1, 1]])
def test_expand_dims(axis, hip, oracle):
    nd = _nd()
    x = synth.uniform((2, 3, 4), 64, 0.0, 1.0)
    got = nd.expand_dims(nd.array(x).gpu(), axis)
    want = oracle.expand_dims(x, axis)
    assert got.shape() == list(want.shape) and _bits_equal(got.cpu().numpy(), want)
    assert got.shape() == list(np.expand_dims(x, tuple(axis) if isinstance(axis, list) else axis).shape)


def test_expand_dims_errors(hip):
    nd = _nd()
    from numpower_amd.ndarray import Error
    g = nd.array(np.ones((2, 3), np.float32)).gpu()
    with pytest.raises(Error, match="invalid axis or axes provided."):
        nd.expand_dims(g, 5)
    with pytest.raises(Error, match="invalid axis or axes provided."):
        nd.expand_dims(g, [0, 0])


def test_append(hip, oracle):
    nd = _nd()
    a = synth.uniform((3, 5), 65, 0.0, 1.0)
    b = synth.uniform((1_000_001,), 66, 0.0, 1.0)
    got = nd.append(nd.array(a).gpu(), nd.array(b).gpu())
    assert got.shape() == [15 + b.size] and _bits_equal(got.cpu().numpy(), oracle.append(a, b))
    # a host scalar operand is copied H2D (manipulation.c:344-347)
    got = nd.append(nd.array(a).gpu(), 7.5)
    assert _bits_equal(got.cpu().numpy(), np.concatenate([a.reshape(-1), np.float32([7.5])]))
    from numpower_amd.ndarray import Error
    with pytest.raises(Error, match="Device mismatch"):
        nd.append(nd.array(a).gpu(), nd.array(b))


SLICES = [
    ([0, 2],),                      # leading range: stays a view
    ([1],),                         # integer index: row view
    ([0, 6, 2],),                   # stepped leading range: gathered
    ([5, 0, -1],),                  # negative step
    ([1, 5], [2, 7]),               # 2-D window
    ([2], [1, 9, 3]),               # row pick + stepped columns
    ([0, 6], [3]),                  # column pick
    ([-3, 100], [-4, -1]),          # negative / clamped bounds
    ([4, 2],),                      # empty
    ([1, 5], [0, 10], [1]),         # 3 indices on a 3-D array
]


@pytest.mark.parametrize("indices", SLICES)
def test_slice_matches_oracle(indices, hip, oracle):
    nd = _nd()
    shape = (6, 10, 3) if len(indices) == 3 else (6, 10)
    x = synth.uniform(shape, 67, 0.0, 1.0)
    g = nd.array(x).gpu()
    want = oracle.slice_(x, *indices)
    got = g.slice(*indices)
    assert got.shape() == list(want.shape)
    if want.size:
        assert _bits_equal(got.cpu().numpy(), want)
        # whatever came back is contiguous: the next op reads it correctly
        assert _bits_equal((got * 2.0).cpu().numpy(), want * np.float32(2.0))


def test_slice_view_vs_gather(hip):
    nd = _nd()
    x = synth.uniform((8, 16), 68, 0.0, 1.0)
    g = nd.array(x).gpu()
    before = nd.live_device_allocations()
    v = g.slice([2, 6])                       # contiguous rows: view, no allocation
    assert nd.live_device_allocations() == before
    c = g.slice([0, 8, 2])                    # strided: one gather into a new buffer
    assert nd.live_device_allocations() == before + 1
    g.fill(1.0)
    assert (v.cpu().numpy() == 1.0).all() and _bits_equal(c.cpu().numpy(), x[::2])
    from numpower_amd.ndarray import Error
    with pytest.raises(Error, match="too many indices for array."):
        g.slice([0], [0], [0])
    with pytest.raises(Error, match="slice step cannot be zero"):
        g.slice([0, 4, 0])


def test_strided_copy_abi(hip):
    """np_strided_copy directly: negative, zero (broadcast) and permuted strides, large gather."""
    from numpower_amd import _lib
    lib = _lib.load()
    x = synth.uniform((300, 500), 69, 0.0, 1.0)
    src = _lib.DeviceBuffer(x.nbytes)
    _lib.check(lib.np_memcpy_h2d(src.ptr, x.ctypes.data, x.nbytes))

    def gather(offset, shape, strides):
        n = int(np.prod(shape))
        out = _lib.DeviceBuffer(4 * n)
        sh = (C.c_int * len(shape))(*shape)
        st = (C.c_longlong * len(strides))(*strides)
        _lib.check(lib.np_strided_copy(src.ptr + 4 * offset, out.ptr, len(shape), sh, st))
        host = np.empty(n, np.float32)
        _lib.check(lib.np_memcpy_d2h(host.ctypes.data, out.ptr, 4 * n))
        return host.reshape(shape)

    assert _bits_equal(gather(0, (500, 300), (1, 500)), x.T)                       # transpose by strides
    assert _bits_equal(gather(299 * 500 + 499, (300, 500), (-500, -1)), x[::-1, ::-1])
    assert _bits_equal(gather(7, (300, 4), (500, 0)), np.repeat(x[:, 7:8], 4, axis=1))
    assert _bits_equal(gather(0, (150, 250), (1000, 2)), x[::2, ::2])
    assert _bits_equal(gather(0, (300, 500), (500, 1)), x)                         # contiguous -> memcpy path


@pytest.mark.parametrize("mn", [(1, 1), (3, 5), (257, 1001), (1024, 4096), (4001, 8)])
def test_outer(mn, hip, oracle):
    """NDArray_Outer (linalg.c:724-751): bit-exact vs the restated sger-on-zeros, incl. +0.0 for
    zero products of either sign."""
    nd = _nd()
    m, n = mn
    a = synth.uniform((m,), 71, -2.0, 2.0)
    b = synth.uniform((n,), 72, -2.0, 2.0)
    a[::3] = 0.0
    b[::4] = -0.0
    got = nd.outer(nd.array(a).gpu(), nd.array(b).gpu())
    want = oracle.outer(a, b)
    assert got.shape() == [m, n] and _bits_equal(got.cpu().numpy(), want)
    assert not np.signbit(want[want == 0.0]).any()
    from numpower_amd.ndarray import Error
    with pytest.raises(Error, match="1-dimensional vectors"):
        nd.outer(nd.array(np.ones((2, 2), np.float32)).gpu(), nd.array(b).gpu())
    with pytest.raises(Error, match="same device"):
        nd.outer(nd.array(a).gpu(), nd.array(b))
