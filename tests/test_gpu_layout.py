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
