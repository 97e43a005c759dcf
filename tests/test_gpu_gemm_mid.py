"""sgemm_dmas_kernel — the LDS-DMA pipeline on 128x128 / 128x64 / 64x64 tiles with K split S ways inside one launch and
a distributed, deterministic fold (np_sgemm.hip) — forced through np_sgemm_set_variant(-(1000 + 100 * shape + S)) for
every tile shape and split on aligned, ragged and unaligned products, single and batched: within 1e-6 |A|.|B| of the fp64
product (the bar of tests/test_gpu_parity.py::test_matmul_*), within 1e-5 of the oracle (cblas_sgemm, linalg.c:75-79),
bit-identical from run to run (the fold order is fixed), and the tickets it borrows come back clean."""
import ctypes as C

import numpy as np
import pytest

from numpower_amd import synth
from numpower_amd._lib import check, load

pytestmark = pytest.mark.gpu

SHAPES = [(128, 128, 64), (256, 192, 1024), (96, 200, 33), (64, 64, 31), (200, 136, 97), (1024, 1024, 1024), (1000, 1000, 1000), (1001, 1003, 1002), (77, 65, 130),
          (130, 66, 4097), (64, 64, 20), (513, 259, 777), (300, 8, 515), (1, 128, 256), (129, 4, 2048), (2048, 2048, 128)]


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 5], ids=["128x128", "128x64 BK32", "64x64 BK32", "128x128 BK32", "64x64 BK32 on 8 waves"])
@pytest.mark.parametrize("S", [1, 2, 4, 8, 16])
def test_mid_tiles_every_split(tile, S, hip, oracle):
    lib = load()
    for (m, n, k) in SHAPES:
        A = synth.uniform((m, k), 41, -1.0, 1.0)
        B = synth.uniform((k, n), 42, -1.0, 1.0)
        a, b, c = hip.DeviceArray.from_host(A), hip.DeviceArray.from_host(B), hip.DeviceArray((m, n))
        check(lib.np_sgemm_set_variant(-(1000 + 100 * tile + S)))
        try:
            runs = []
            for _ in range(3):
                hip.fill(c, float("nan"))
                hip.sgemm(a, b, out=c)
                runs.append(c.to_host().copy())
        finally:
            check(lib.np_sgemm_set_variant(-999))
        want = A.astype(np.float64) @ B.astype(np.float64)
        scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
        assert not np.isnan(runs[0]).any(), (m, n, k)
        assert (np.abs(runs[0] - want) <= 1e-6 * scale).all(), (m, n, k, float((np.abs(runs[0] - want) / scale).max()))
        for r in runs[1:]:
            assert (r.view(np.uint32) == runs[0].view(np.uint32)).all(), (m, n, k, "not deterministic")
        ref = oracle.matmul(A, B)
        assert (np.abs(runs[0] - ref) <= 1e-5 * scale).all(), (m, n, k, "vs the oracle")
    assert lib.np_sync() == 0, lib.np_last_error()


@pytest.mark.parametrize("tile,S", [(0, 4), (1, 2), (2, 1), (2, 4), (3, 2), (5, 2), (5, 1), (5, 4)])
def test_mid_tiles_batched_and_strided(tile, S, hip):
    lib = load()
    batch, m, n, k = 3, 200, 136, 520
    A = synth.uniform((batch, m, k), 51, -1.0, 1.0)
    B = synth.uniform((batch, k, n), 52, -1.0, 1.0)
    a, b, c = hip.DeviceArray.from_host(A), hip.DeviceArray.from_host(B), hip.DeviceArray((batch, m, n))
    check(lib.np_sgemm_set_variant(-(1000 + 100 * tile + S)))
    try:
        hip.fill(c, float("nan"))
        check(lib.np_sgemm_strided_batched(batch, m, n, k, a.ptr, m * k, b.ptr, k * n, c.ptr, m * n))
        got = c.to_host()
    finally:
        check(lib.np_sgemm_set_variant(-999))
    want = A.astype(np.float64) @ B.astype(np.float64)
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    assert (np.abs(got - want) <= 1e-6 * scale).all()


KQ_SHAPES = [(48, 48, 16), (48, 48, 64), (50, 72, 80), (100, 200, 64), (16, 16, 16), (1, 4, 16), (97, 132, 208), (720, 720, 720), (333, 444, 176), (64, 64, 1024),
             (130, 68, 4096), (768, 768, 768), (512, 512, 512), (96, 200, 33), (77, 64, 130), (513, 260, 784), (60, 60, 4), (60, 60, 20), (100, 52, 36),
             (200, 200, 100), (700, 700, 700), (50, 48, 1000), (33, 36, 68), (48, 48, 60), (48, 48, 124),
             (97, 131, 67), (50, 50, 5), (64, 64, 4), (33, 35, 1001), (129, 66, 130), (200, 136, 97), (1, 1, 7), (65, 3, 63), (300, 8, 515)]


KQ_TILES = ["48x48", "32x32", "64x64", "48x32", "64x32", "64x48", "80x48"]


@pytest.mark.parametrize("shape", list(range(len(KQ_TILES))), ids=KQ_TILES)
def test_k_quartered_tiles(shape, hip, oracle):
    """sgemm_kq_kernel (one tile per workgroup, its four waves split every 64-deep K-tile and load their operands straight from
    memory into the v_mfma_f32_16x16x4 layouts, the partial tiles summed in LDS in wave order) forced through np_sgemm_set_variant(-(2000 + shape)): ragged M / N, K = 4 .. 4096 (last K-tile from one
    chunk to full, quarters that end inside, lanes that straddle K when K % 4 != 0), rows that are not float4-loadable (odd K / N:
    the dword-load form); K < 4 falls through to the planner.  Same bars as above; deterministic."""
    lib = load()
    for (m, n, k) in KQ_SHAPES:
        A = synth.uniform((m, k), 61, -1.0, 1.0)
        B = synth.uniform((k, n), 62, -1.0, 1.0)
        a, b, c = hip.DeviceArray.from_host(A), hip.DeviceArray.from_host(B), hip.DeviceArray((m, n))
        check(lib.np_sgemm_set_variant(-(2000 + shape)))
        try:
            runs = []
            for _ in range(2):
                hip.fill(c, float("nan"))
                hip.sgemm(a, b, out=c)
                runs.append(c.to_host().copy())
        finally:
            check(lib.np_sgemm_set_variant(-999))
        want = A.astype(np.float64) @ B.astype(np.float64)
        scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
        assert not np.isnan(runs[0]).any(), (m, n, k)
        assert (np.abs(runs[0] - want) <= 1e-6 * scale).all(), (m, n, k, float((np.abs(runs[0] - want) / scale).max()))
        assert (runs[1].view(np.uint32) == runs[0].view(np.uint32)).all(), (m, n, k, "not deterministic")
        assert (np.abs(runs[0] - oracle.matmul(A, B)) <= 1e-5 * scale).all(), (m, n, k, "vs the oracle")
    # a batch with strides, and non-finite values in the rows / columns a ragged last K-tile re-reads: they must not leak
    batch, m, n, k = 3, 100, 72, 80
    A = synth.uniform((batch, m, k), 63, -1.0, 1.0)
    B = synth.uniform((batch, k, n), 64, -1.0, 1.0)
    B[:, k - 1, 5] = np.inf          # the row a last K-tile's out-of-range rows are pulled back to
    A[:, 7, 64:] = np.nan            # row 7's last chunks
    a, b, c = hip.DeviceArray.from_host(A), hip.DeviceArray.from_host(B), hip.DeviceArray((batch, m, n))
    check(lib.np_sgemm_set_variant(-(2000 + shape)))
    try:
        check(lib.np_sgemm_strided_batched(batch, m, n, k, a.ptr, m * k, b.ptr, k * n, c.ptr, m * n))
        got = c.to_host()
    finally:
        check(lib.np_sgemm_set_variant(-999))
    with np.errstate(invalid="ignore", over="ignore"):
        want = np.matmul(A.astype(np.float64), B.astype(np.float64))
    assert (np.isnan(got) == np.isnan(want)).all() and (np.isinf(got) == np.isinf(want)).all()
    fin = np.isfinite(want)
    assert np.allclose(got[fin], want[fin], rtol=0, atol=1e-4)
    assert lib.np_sync() == 0, lib.np_last_error()


KQ_CHUNKED = [(100, 100, 100000), (64, 64, 30000), (100, 100, 100001), (96, 160, 7002), (20, 40, 50000), (130, 68, 4099), (200, 200, 5000), (33, 35, 2049),
              (64, 36, 16384), (48, 48, 2051), (100, 100, 512), (256, 256, 8192), (300, 300, 2500), (17, 17, 65537)]


@pytest.mark.parametrize("shape,S", [(2, 63), (2, 5), (1, 16), (1, 7), (0, 9), (3, 32), (4, 32), (5, 12), (6, 3), (2, 255)])
def test_k_quartered_tiles_over_k_chunks(shape, S, hip, oracle):
    """Deep-K products of a few tiles: K cut into chunks that run as ONE launch of sgemm_kq_kernel (chunk-major over the XCDs:
    GemmArgs::k_chunks) and are folded by np_reduce_axis — forced through np_sgemm_set_variant(-(30000 + 1000 * shape + S)) for
    every tile shape, chunk counts that are not multiples of eight, remainder chunks of every length (a remainder under four
    inner elements falls through to the older tiles), rows that are not float4-loadable.  Same bars as above; the chunk-major and
    the tile-major launch (np_sgemm_set_variant(-25)) compute the same chunks, so their results are bit-identical."""
    lib = load()
    for (m, n, k) in KQ_CHUNKED:
        A = synth.uniform((m, k), 71, -1.0, 1.0)
        B = synth.uniform((k, n), 72, -1.0, 1.0)
        a, b, c = hip.DeviceArray.from_host(A), hip.DeviceArray.from_host(B), hip.DeviceArray((m, n))
        check(lib.np_sgemm_set_variant(-(30000 + 1000 * shape + S)))
        try:
            runs = []
            for major in (-26, -26, -25):
                check(lib.np_sgemm_set_variant(major))
                hip.fill(c, float("nan"))
                hip.sgemm(a, b, out=c)
                runs.append(c.to_host().copy())
        finally:
            check(lib.np_sgemm_set_variant(-26))
            check(lib.np_sgemm_set_variant(-30000))
        want = A.astype(np.float64) @ B.astype(np.float64)
        scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
        assert not np.isnan(runs[0]).any(), (m, n, k)
        assert (np.abs(runs[0] - want) <= 1e-6 * scale).all(), (m, n, k, float((np.abs(runs[0] - want) / scale).max()))
        for r in runs[1:]:
            assert (r.view(np.uint32) == runs[0].view(np.uint32)).all(), (m, n, k, "not deterministic / launch order matters")
        assert (np.abs(runs[0] - oracle.matmul(A, B)) <= 1e-5 * scale).all(), (m, n, k, "vs the oracle")
    assert lib.np_sync() == 0, lib.np_last_error()


def test_deep_k_products_take_the_k_chunked_plan(hip, oracle):
    """What the planner does by itself with the deep-K class (profiles/r05/gemm_deep_k_ab.log): the K-chunked k-quartered plan,
    also for the 17..64-row shapes the thin K-chunk kernels leave underfilled; non-finite values in one chunk stay in the
    elements they belong to."""
    lib = load()
    out = (C.c_double * 11)()
    for (m, n, k) in [(100, 100, 100000), (128, 128, 65536), (64, 64, 100000), (200, 200, 50000), (32, 64, 100000), (100, 100, 100001), (160, 96, 25000)]:
        check(lib.np_sgemm_debug_plan(m, n, k, 1, 0, out))
        assert out[0] >= 6 and out[1] > 0 and out[2] >= 2, (m, n, k, list(out))
        A = synth.uniform((m, k), 73, -1.0, 1.0)
        B = synth.uniform((k, n), 74, -1.0, 1.0)
        A[3, k // 2] = np.inf
        B[k - 1, 1] = np.nan
        a, b, c = hip.DeviceArray.from_host(A), hip.DeviceArray.from_host(B), hip.DeviceArray((m, n))
        hip.fill(c, 5.0)
        hip.sgemm(a, b, out=c)
        got = c.to_host()
        with np.errstate(invalid="ignore", over="ignore"):
            want = A.astype(np.float64) @ B.astype(np.float64)
        assert (np.isnan(got) == np.isnan(want)).all() and (np.isinf(got) == np.isinf(want)).all(), (m, n, k)
        fin = np.isfinite(want)
        scale = np.abs(np.nan_to_num(A, posinf=0.0)).astype(np.float64) @ np.abs(np.nan_to_num(B, nan=0.0)).astype(np.float64)
        assert (np.abs(got[fin] - want[fin]) <= 1e-6 * scale[fin]).all(), (m, n, k)
    assert lib.np_sync() == 0, lib.np_last_error()
