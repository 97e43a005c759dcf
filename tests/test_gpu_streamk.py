"""nd::matmul away from the shapes that tile the machine evenly: the stream-K form of the LDS-DMA GEMM
(sgemm_streamk_kernel: workgroups take equal contiguous ranges of K-TILES, tiles that straddle two ranges are
finished by the workgroup holding their k = 0 from the other's partial).  The reference path is one cblas_sgemm call
whatever the shape (linalg.c:44-82): the bar is the same as everywhere — 1e-5 of |A|.|B| against the oracle's OpenBLAS
product, 1e-6 against fp64 — plus what is specific to a scheme that splits K:
  * forced on (np_sgemm_set_variant(-4)) and forced off (-5) agree to 1e-6 |A|.|B| with each other,
  * the result is DETERMINISTIC (the partial sums are added in schedule order): two runs are bit-identical,
  * the flags it signals with end every launch as they began it (a second, different shape right behind works),
  * ragged M / N (EDGE), K % 16 != 0 (KTAIL) and unaligned leading dimensions (the padded path) all take it."""
import numpy as np
import pytest

from numpower_amd import synth
from numpower_amd._lib import check, load

pytestmark = pytest.mark.gpu

SHAPES = [(3000, 3000, 3000), (2048, 2048, 2048), (2000, 2000, 2000), (1536, 1536, 1536), (4097, 4097, 4097),
          (1000, 3000, 5000), (2050, 2060, 2048), (2048, 2048, 2056), (300, 4000, 8200), (256, 128, 4096 * 6),
          (257, 129, 12000), (512, 512, 512)]


@pytest.fixture()
def streamk_forced():
    lib = load()
    check(lib.np_sgemm_set_variant(-4))
    yield lib
    check(lib.np_sgemm_set_variant(-2))


@pytest.mark.parametrize("shape", SHAPES, ids=["%dx%dx%d" % s for s in SHAPES])
def test_streamk_matches_fp64_the_oracle_and_the_tile_form(shape, hip, oracle, streamk_forced):
    lib = streamk_forced
    m, n, k = shape
    A = synth.uniform((m, k), 31, -1.0, 1.0)
    B = synth.uniform((k, n), 32, -1.0, 1.0)
    dA, dB = hip.DeviceArray.from_host(A), hip.DeviceArray.from_host(B)
    c1 = hip.sgemm(dA, dB).to_host()
    c2 = hip.sgemm(dA, dB).to_host()
    assert (c1.view(np.uint32) == c2.view(np.uint32)).all(), "stream-K result differs from run to run"
    check(lib.np_sgemm_set_variant(-5))
    tile = hip.sgemm(dA, dB).to_host()
    check(lib.np_sgemm_set_variant(-4))
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    ref64 = A.astype(np.float64) @ B.astype(np.float64)
    assert (np.abs(c1 - ref64) / scale).max() <= 1e-6
    assert (np.abs(c1.astype(np.float64) - tile) / scale).max() <= 1e-6
    if m * n * k <= 3000 ** 3:
        ref = oracle.matmul(A, B)
        assert (np.abs(c1 - ref) <= 1e-5 * scale).all()
    for d in (dA, dB):
        d.free()


def test_streamk_back_to_back_shapes_and_default_planner(hip, streamk_forced):
    """Different shapes right behind each other (the flags are clean again after every launch), then the default
    planner on the same inputs: whichever form its model picks, the product is the same to 1e-6 |A|.|B|."""
    lib = streamk_forced
    outs = []
    for (m, n, k) in ((3000, 3000, 3000), (2048, 1024, 4096), (3000, 3000, 3000), (1111, 2222, 3332)):
        A = synth.uniform((m, k), 41, -1.0, 1.0)
        B = synth.uniform((k, n), 42, -1.0, 1.0)
        dA, dB = hip.DeviceArray.from_host(A), hip.DeviceArray.from_host(B)
        got = hip.sgemm(dA, dB).to_host()
        scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
        assert (np.abs(got - A.astype(np.float64) @ B.astype(np.float64)) / scale).max() <= 1e-6, (m, n, k)
        check(lib.np_sgemm_set_variant(-2))
        dflt = hip.sgemm(dA, dB).to_host()
        check(lib.np_sgemm_set_variant(-4))
        assert (np.abs(got.astype(np.float64) - dflt) / scale).max() <= 1e-6, (m, n, k)
        outs.append(got)
        for d in (dA, dB):
            d.free()
    assert (outs[0].view(np.uint32) == outs[2].view(np.uint32)).all()      # same inputs, same bits, with other launches in between


def test_streamk_random_shapes(hip, streamk_forced):
    """Random M, N, K (aligned and not, ragged tiles, K tails, shapes the stream-K model refuses and that fall back to
    the tile plans) with stream-K forced wherever the kernel can run it: every product within 1e-6 |A|.|B| of fp64 and
    bit-identical when repeated."""
    rng = np.random.default_rng(20260927)
    for case in range(24):
        m = int(rng.integers(1, 2600))
        n = int(rng.integers(1, 2600))
        k = int(rng.integers(16, 5200))
        if case % 3 == 0:                      # float4-loadable rows: the LDS-DMA kernels take them directly
            n, k = max(4, n // 4 * 4), max(16, k // 4 * 4)
        if case % 6 == 0:
            m, n = max(256, m // 256 * 256), max(128, n // 128 * 128)     # whole tiles
        A = synth.uniform((m, k), 5000 + case, -1.0, 1.0)
        B = synth.uniform((k, n), 6000 + case, -1.0, 1.0)
        dA, dB = hip.DeviceArray.from_host(A), hip.DeviceArray.from_host(B)
        c1 = hip.sgemm(dA, dB).to_host()
        c2 = hip.sgemm(dA, dB).to_host()
        assert (c1.view(np.uint32) == c2.view(np.uint32)).all(), (m, n, k)
        scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
        err = np.abs(c1 - A.astype(np.float64) @ B.astype(np.float64)) / np.maximum(scale, 1e-30)
        assert err.max() <= 1e-6, (m, n, k, float(err.max()))
        for d in (dA, dB):
            d.free()
