"""Device-side initializers (the reference's own phpbench suite times full / zeros / ones / identity /
arange on the CPU, benchmarks/initializers): arrays are born on the GPU.  np_arange must reproduce
NDArray_Arange's float recurrence (initializers.c:836-839) bit for bit even though it evaluates
per-binade arithmetic segments in parallel."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _nd():
    from numpower_amd.ndarray import NDArray
    return NDArray


ARANGES = [  # (stop, start, step)
    (10, 0, 1), (20, 10, 1), (10, 1, 1), (1000, 0, 0.1), (1, 0, 1e-4), (100, -100, 0.37), (-50, 50, -0.25),
    (3.0e7, 0, 1), (1.7e7, 1.6e7, 0.5), (2.0e7, 1.67772e7, 1),                  # across 2^24: the sequence stops moving
    (1e6, 0, 1.0 / 3.0), (5, -5, 1e-5), (1e-30, 0, 1e-36), (1e-37, 0, 1e-44),   # tiny steps, denormals
    (3e38, 1e38, 1e32), (10, 0, 2.5), (7.5, 0.1, 0.7), (1e5, 0, 0.1 + 2 ** -30),
    (1, -1, 2 ** -24), (65536, 0, 0.5 + 2 ** -20), (1e4, 1.0, 1 + 2 ** -23),    # steps on / next to lattice half-points (ties)
    (2 ** 20 + 100, 2 ** 20 - 100, 2 ** -4 + 2 ** -5), (300, 0, 1e-3), (-1e6, 0, -7.3), (0.5, -0.5, 1e-6),
]


@pytest.mark.parametrize("stop,start,step", ARANGES)
def test_arange_is_the_reference_recurrence(stop, start, step, hip, oracle):
    nd = _nd()
    want = oracle.arange(stop, start, step)
    got = nd.arange(stop, start, step).cpu().numpy()
    assert got.shape == want.shape
    diff = got.view(np.uint32) != want.view(np.uint32)
    assert not diff.any(), "first mismatch at %d: got %r want %r" % (int(np.argmax(diff)), got[np.argmax(diff)], want[np.argmax(diff)])


def test_arange_random_sweep(hip, oracle):
    """400 random (start, step, n) incl. negative steps and magnitudes from 1e-20 to 1e20."""
    nd = _nd()
    rng = np.random.default_rng(11)
    for _ in range(400):
        mag = 10.0 ** rng.uniform(-20, 20)
        start = float(rng.uniform(-1, 1) * mag)
        step = float(rng.uniform(-1, 1) * mag * 10.0 ** rng.uniform(-8, 0))
        if step == 0.0:
            continue
        n = int(rng.integers(1, 200_000))
        stop = start + step * (n - 0.5)
        want = oracle.arange(stop, start, step)
        got = nd.arange(stop, start, step).cpu().numpy()
        assert got.shape == want.shape and (got.view(np.uint32) == want.view(np.uint32)).all(), (start, step, n)


def test_arange_errors(hip):
    nd = _nd()
    from numpower_amd.ndarray import Error
    with pytest.raises(Error, match="arange: zero length"):
        nd.arange(0, 10, 1)
    with pytest.raises(Error, match="arange: overflow while computing length"):
        nd.arange(1e30, 0, 1e-5)
    with pytest.raises(Error, match="only computes on the GPU"):
        nd.arange(10, 0, 1, 0)


def test_full_ones_zeros_identity_on_device(hip, oracle):
    nd = _nd()
    before = nd.live_device_allocations()
    f = nd.full([1000, 1003], 4.0, 1)
    assert f.isGPU() and f.shape() == [1000, 1003] and (f.cpu().numpy() == 4.0).all()
    o = nd.ones([7, 5, 3], 1)
    assert (o.cpu().numpy() == np.ones((7, 5, 3), np.float32)).all()
    z = nd.zeros([33, 65], 1)
    assert (z.cpu().numpy().view(np.uint32) == 0).all()
    for n in (0, 1, 4, 257, 2048):
        i = nd.identity(n, 1)
        want = oracle.identity(n)
        assert i.shape() == list(want.shape)
        if n:
            assert (i.cpu().numpy().view(np.uint32) == want.view(np.uint32)).all()
    # and they feed the hot path like any other array
    assert ((f + o.__class__.full([1000, 1003], 1.0, 1)).cpu().numpy() == 5.0).all()
    eye = nd.identity(64, 1)
    x = nd.array(np.arange(64 * 64, dtype=np.float32).reshape(64, 64)).gpu()
    assert (nd.matmul(eye, x).cpu().numpy() == x.cpu().numpy()).all()
    del f, o, z, i, eye, x
    assert nd.live_device_allocations() == before
    # CPU variants stay plain stores
    c = nd.full([3, 2], 2.5)
    assert not c.isGPU() and c.toArray() == [[2.5, 2.5], [2.5, 2.5], [2.5, 2.5]]
    assert nd.identity(3).toArray() == [[1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, 1.0]]
