"""SURVEY.md §8(f) row 2 (second half) on the GPU: argmax / argmin (src/ndmath/calculation.c:9-194),
exact index parity with the oracle's restatement of float_argmax / float_argmin, including ties
(first occurrence) and the two functions' different NaN rules."""
import numpy as np
import pytest

from numpower_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(7,), (1000,), (1000, 1000), (257, 1001), (3, 5, 64), (100003,)])
@pytest.mark.parametrize("is_max", [True, False])
def test_arg_flat_and_axes(shape, is_max, hip, oracle):
    from numpower_amd.ndarray import NDArray
    x = synth.uniform(shape, 81, -5.0, 5.0)
    flat = x.reshape(-1)
    flat[::37] = flat[3]                      # ties: the first occurrence must win
    g = NDArray.array(x).gpu()
    fn = NDArray.argmax if is_max else NDArray.argmin
    got = fn(g)
    assert isinstance(got, float)
    assert got == float(oracle.argreduce(x, None, is_max))
    assert got == float(np.argmax(flat) if is_max else np.argmin(flat))
    for axis in range(len(shape)):
        got = fn(g, axis)
        got = got.cpu().numpy() if not isinstance(got, float) else np.float32(got)
        want = oracle.argreduce(x, axis, is_max)
        assert got.shape == want.shape
        assert (got == want).all()
        assert (got == (np.argmax(x, axis) if is_max else np.argmin(x, axis))).all()
    if len(shape) == 2:
        kd = fn(g, 1, True)
        assert kd.shape() == [shape[0], 1]


def test_arg_nan_rules(hip, oracle):
    from numpower_amd.ndarray import NDArray
    nan = np.nan
    cases = [np.array([nan, 1, 5, 2], np.float32),       # argmax: NaN first is maximal -> 0
             np.array([1, nan, 5, 2], np.float32),       # argmax skips later NaNs -> 2; argmin -> 1
             np.array([1, 5, nan, nan, -7], np.float32), # argmin: first NaN wins -> 2
             np.array([-np.inf, nan, 3], np.float32),
             np.array([np.inf, 2, np.inf], np.float32),
             np.array([nan, nan], np.float32)]
    big = synth.uniform((50000,), 5, -1, 1)
    big[40000] = nan
    big[123] = 7.0
    cases.append(big)
    for x in cases:
        g = NDArray.array(x).gpu()
        assert NDArray.argmax(g) == float(oracle.argreduce(x, None, True)), x[:8]
        assert NDArray.argmin(g) == float(oracle.argreduce(x, None, False)), x[:8]
    m = np.stack([cases[1], cases[0]])
    g = NDArray.array(m).gpu()
    assert (NDArray.argmax(g, 1).cpu().numpy() == oracle.argreduce(m, 1, True)).all()
    assert (NDArray.argmin(g, 0).cpu().numpy() == oracle.argreduce(m, 0, False)).all()


def test_arg_errors(hip):
    from numpower_amd.ndarray import Error, NDArray
    g = NDArray.array(np.ones((3, 4), np.float32)).gpu()
    with pytest.raises(Error, match="Invalid axis parameter"):
        NDArray.argmax(g, 5)
    with pytest.raises(Error, match="only computes on the GPU"):
        NDArray.argmax(NDArray.array(np.ones((3, 4), np.float32)))


def test_argmax_1e8(hip):
    D = hip
    import ctypes as C
    from numpower_amd._lib import check, load
    n = 100_000_000
    x = synth.uniform((n,), 9, 0.0, 1.0)
    x[77_777_777] = 2.0
    x[12_345] = -1.0
    d = D.DeviceArray.from_host(x)
    out = D.DeviceArray((1,))
    check(load().np_argreduce(1, d.ptr, 1, n, 1, out.ptr))
    assert out.to_host()[0] == np.float32(77_777_777)
    check(load().np_argreduce(0, d.ptr, 1, n, 1, out.ptr))
    assert out.to_host()[0] == np.float32(12_345)
    d.free()
    del C


@pytest.mark.parametrize("shape,axis", [((300_000, 3), 0), ((3, 300_000), 1), ((1, 1_000_003), 1), ((1_000_003,), None),
                                        ((2000, 1003), 0), ((5, 40_000, 7), 1), ((70_000, 1), 0), ((2, 3, 50_000), 2)])
def test_argreduce_few_outputs_long_axis(shape, axis, hip, oracle):
    """Shapes where one thread per output would idle the chip: the axis is cut into chunks whose
    (value, index) partials are folded (argreduce_chunks_kernel / vectorised argreduce_rows_kernel).
    First-occurrence ties and the NaN rules must survive the chunk boundaries."""
    from numpower_amd.ndarray import NDArray
    x = synth.uniform(shape, 51, -1.0, 1.0)
    flat = x.reshape(-1)
    flat[::7] = np.float32(0.75)                    # many exact ties for the maximum region
    flat[3::11] = np.float32(-0.75)
    flat[flat.size // 2] = np.nan                   # a NaN that is not in position 0
    g = NDArray.array(x).gpu()
    for is_max in (True, False):
        got = NDArray.argmax(g, axis) if is_max else NDArray.argmin(g, axis)
        got = got.cpu().numpy() if not isinstance(got, float) else np.float32(got)
        want = oracle.argreduce(x, axis, is_max)
        assert np.array_equal(np.asarray(got, np.float32).reshape(-1), np.asarray(want, np.float32).reshape(-1))


@pytest.mark.parametrize("cols", [1, 2, 3, 4, 5, 8, 10, 16, 33, 100, 255, 256, 257])
def test_argreduce_short_rows(cols, hip, oracle):
    """argmax / argmin over the class scores of many samples ((N, 10), axis 1): lane groups per row
    (argreduce_rows_group), with ties and NaNs in every position."""
    from numpower_amd.ndarray import NDArray
    rows = 5003
    x = synth.uniform((rows, cols), 52, -1.0, 1.0)
    x[::7, :] = np.float32(0.5)                        # whole rows of ties: first index wins
    if cols > 1:
        x[1::11, cols // 2] = np.nan
        x[2::13, 0] = np.nan                           # NaN in position 0: maximal for argmax, first NaN for argmin
        x[3::17, cols - 1] = np.float32(2.0)
    g = NDArray.array(x).gpu()
    for is_max in (True, False):
        got = (NDArray.argmax(g, 1) if is_max else NDArray.argmin(g, 1)).cpu().numpy()
        want = oracle.argreduce(x, 1, is_max)
        assert np.array_equal(np.asarray(got, np.float32), np.asarray(want, np.float32)), (cols, is_max)


@pytest.mark.parametrize("shape,axis", [
    ((4200, 1024), 1), ((4100, 1000), 1), ((4100, 1027), 1), ((4099, 300), 1), ((4100, 8195), 1),   # one wave per row
    ((3, 1_000_001), 1), ((1, 5_000_003), 1), ((600, 40_001), 1),                                   # (row, chunk) workgroups
    ((200_003, 3), 0), ((100_001, 7), 0), ((33_000, 63), 0), ((2, 50_001, 5), 1),                   # a few columns, flat float4 walk
    ((50_001, 8), 0), ((30_001, 100), 0), ((7, 3001, 100), 1), ((9001, 256), 0), ((300, 77, 12), 1),  # whole float4 column groups
    ((2000, 1003), 0), ((4096, 1024), 0), ((1500, 9973), 0), ((4000, 20, 201), 1), ((3, 700, 260), 1), ((5, 17, 193), 1),  # wide inner
])
def test_argreduce_streaming_forms(shape, axis, hip, oracle):
    """Every streaming form of round 5 (per-component accumulators, strict compares inside a lane, the general tie / NaN
    combine only across lanes): exact indices against the oracle with whole regions of ties, +-inf, NaNs in position 0 of
    some rows / columns (maximal for argmax, first NaN for argmin) and NaNs elsewhere, ragged sizes off every vector width."""
    from numpower_amd._lib import check, load
    x = synth.uniform(shape, 53, -1.0, 1.0)
    flat = x.reshape(-1)
    flat[::7] = np.float32(0.75)
    flat[3::11] = np.float32(-0.75)
    flat[5::1013] = np.inf
    flat[6::1511] = -np.inf
    flat[flat.size // 2] = np.nan
    flat[9::4099] = np.nan
    xm = np.moveaxis(x, axis, 0)
    xm[0, ...].reshape(-1)[::5] = np.nan               # position 0 of every fifth output is a NaN
    outer = int(np.prod(shape[:axis], dtype=np.int64))
    inner = int(np.prod(shape[axis + 1:], dtype=np.int64))
    d = hip.DeviceArray.from_host(x)
    out = hip.DeviceArray((outer * inner,))
    for is_max in (True, False):
        check(load().np_argreduce(1 if is_max else 0, d.ptr, outer, shape[axis], inner, out.ptr))
        got = out.to_host()
        want = np.asarray(oracle.argreduce(x, axis, is_max), np.float32).reshape(-1)
        bad = np.flatnonzero(got != want)
        assert bad.size == 0, (shape, axis, is_max, bad[:5], got[bad[:5]], want[bad[:5]])
    # an all-equal and an all-(-inf) array: index 0 everywhere, whichever lane or chunk looked at what
    for fill in (np.float32(0.25), np.float32(-np.inf), np.float32(np.inf)):
        hip.fill(d, float(fill))
        for is_max in (True, False):
            check(load().np_argreduce(1 if is_max else 0, d.ptr, outer, shape[axis], inner, out.ptr))
            assert not out.to_host().any(), (shape, axis, is_max, fill)
    d.free()
    out.free()
