"""Edge cases of the hot path on the GPU: empty and one-element arrays, 0-d operands on either
side, K = 0 products, and arrays past 2^31 elements (the reference's vmalloc(unsigned int),
gpu_alloc.c:11, stops at 4 GiB per buffer; the C ABI is size_t throughout and the kernels switch to
64-bit indices)."""
import ctypes as C

import numpy as np
import pytest

from numpower_amd import synth

pytestmark = pytest.mark.gpu


def _nd():
    from numpower_amd.ndarray import NDArray
    return NDArray


def test_empty_arrays_flow_through(hip):
    nd = _nd()
    e = nd.array(np.zeros((0,), np.float32)).gpu()
    e2 = nd.array(np.zeros((0, 5), np.float32)).gpu()
    assert (e + e).shape() == [0] and (e * 2.0).shape() == [0]
    assert nd.exp(e).shape() == [0] and nd.abs(e2).shape() == [0, 5]
    assert (e2 + e2).cpu().numpy().shape == (0, 5)
    assert nd.sum(e) == 0.0                       # empty sum; the loop of NDArray_Sum_Float never runs
    assert nd.prod(e) == 1.0
    assert nd.transpose(e2).shape() == [5, 0]
    assert nd.flatten(e2).shape() == [0]
    assert nd.array_equal(e, nd.array(np.zeros((0,), np.float32)).gpu()) is True
    # (0 x 5) . (5 x 3) -> 0 x 3; (3 x 0) . (0 x 4) -> zeros (beta = 0 with an empty inner dimension)
    z = nd.matmul(e2, nd.array(np.ones((5, 3), np.float32)).gpu())
    assert z.shape() == [0, 3]
    k0 = nd.matmul(nd.array(np.zeros((3, 0), np.float32)).gpu(), nd.array(np.zeros((0, 4), np.float32)).gpu())
    assert (k0.cpu().numpy() == np.zeros((3, 4), np.float32)).all()


def test_one_element_and_scalar_operands(hip, oracle):
    nd = _nd()
    one = nd.array(np.float32([[-0.0]])).gpu()
    # 1 x 1 array (op) host scalar, both orders; multiply keeps the reference's zero-sign rule for
    # a one-element (tail-only) loop: -0.0 products become +0.0 (arithmetics.c:410-412)
    got = (one * 3.0).cpu().numpy()
    want = oracle.binary("multiply", np.float32([[-0.0]]), np.float32(3.0))
    assert got.view(np.uint32).tolist() == want.view(np.uint32).tolist()
    got = (3.0 * one).cpu().numpy()
    want = oracle.binary("multiply", np.float32(3.0), np.float32([[-0.0]]))
    assert got.view(np.uint32).tolist() == want.view(np.uint32).tolist()
    x = nd.array(np.float32([2.5])).gpu()
    assert nd.sum(x) == 2.5 and nd.max(x) == 2.5 and nd.min(x) == 2.5 and nd.mean(x) == 2.5
    assert (10.0 - x).cpu().numpy().tolist() == [7.5]
    assert (x ** 2).cpu().numpy().tolist() == [6.25]
    assert nd.matmul(nd.array(np.float32([[3.0]])).gpu(), nd.array(np.float32([[4.0]])).gpu()).cpu().numpy().tolist() == [[12.0]]
    assert nd.argmax(x) == 0.0


def test_ragged_lengths_around_the_vector_width(hip, oracle):
    """Every length 1..40 and a few around 2^k: float4 body, scalar tail, AVX-body bound."""
    nd = _nd()
    for n in list(range(1, 41)) + [255, 256, 257, 1023, 1025, 4099]:
        a = synth.uniform((n,), 80 + n, -2.0, 2.0)
        b = synth.uniform((n,), 180 + n, 0.5, 2.0)
        a[::3] = 0.0
        ga, gb = nd.array(a).gpu(), nd.array(b).gpu()
        for op in ("add", "multiply", "mod", "equal"):
            got = nd._binary(op, ga, gb).cpu().numpy()
            want = oracle.binary(op, a, b)
            assert (got.view(np.uint32) == want.view(np.uint32)).all(), (op, n)
        assert (nd.floor(ga).cpu().numpy() == np.floor(a)).all()
        s = nd.sum(ga)
        assert abs(s - float(a.astype(np.float64).sum())) <= 1e-5 * max(1.0, np.abs(a).sum())


def test_past_2_to_31_elements(hip):
    """fill / unary / binary / fused chain on 2^31 + 4100 elements (8.6 GB per buffer): values are
    probed on both sides of the 2^31 boundary and at the very end (64-bit index kernels)."""
    from numpower_amd import _lib
    from numpower_amd._lib import BINARY_OPS, UNARY_OPS, FusedOp
    lib = _lib.load()
    n = (1 << 31) + 4100
    a, b, out = (_lib.DeviceBuffer(4 * n) for _ in range(3))
    _lib.check(lib.np_fill(a.ptr, 1.5, n))
    _lib.check(lib.np_fill(b.ptr, 2.0, n))
    # make the far end distinguishable: b[n-1] = 10
    ten = np.float32([10.0])
    _lib.check(lib.np_memcpy_h2d(b.ptr + 4 * (n - 1), ten.ctypes.data, 4))

    def probe(buf, index):
        v = C.c_float()
        _lib.check(lib.np_read_float(buf.ptr, index, C.byref(v)))
        return v.value

    probes = [0, 12345, (1 << 31) - 1, 1 << 31, (1 << 31) + 4096, n - 2, n - 1]
    _lib.check(lib.np_binary(BINARY_OPS["add"], a.ptr, 0, b.ptr, 0, out.ptr, 1, n, 0, 0))
    assert [probe(out, i) for i in probes] == [3.5] * 6 + [11.5]
    _lib.check(lib.np_unary(UNARY_OPS["negate"], b.ptr, out.ptr, n, 0.0, 0.0))
    assert [probe(out, i) for i in probes] == [-2.0] * 6 + [-10.0]
    ops = (FusedOp * 2)(FusedOp(1, BINARY_OPS["multiply"], 1, 0, 0, 0, 1, n // 8 * 8),
                        FusedOp(0, UNARY_OPS["sqrt"], 0, 0, 0, 0, 0, 0))
    ptrs = (C.c_void_p * 2)(a.ptr, b.ptr)
    kinds = (C.c_int * 2)(0, 0)
    _lib.check(lib.np_fused_chain(ptrs, kinds, 2, ops, 2, out.ptr, 1, n))
    want = float(np.sqrt(np.float32(3.0)))
    assert [probe(out, i) for i in probes[:-1]] == [want] * 6
    assert probe(out, n - 1) == float(np.sqrt(np.float32(15.0)))
    # a chain that ends in a full reduction (what nd::sum(nd::sqrt($a * $b)) is behind the binding), array_equal / allclose and
    # nd::all beyond 2^31 elements: refused as "array too large" until the end of round 6 — while the op-by-op forms worked
    host = C.c_float()
    _lib.check(lib.np_fused_chain_reduce(ptrs, kinds, 2, ops, 2, 0, 1, n, C.byref(host)))        # 0 = sum
    want_sum = want * (n - 1) + float(np.sqrt(np.float32(15.0)))
    # 2^31 copies of ONE value are the worst case of a per-lane fp32 accumulator: every add inside a binade rounds the same way
    # (2.7e-5 here; random data averages out, and the reference's single sequential accumulator stops growing at 2^24 * value)
    assert abs(host.value - want_sum) <= 1e-4 * want_sum, (host.value, want_sum)
    _lib.check(lib.np_fused_chain_reduce(ptrs, kinds, 2, ops, 2, 3, 1, n, C.byref(host)))        # 3 = max: the far end
    assert host.value == float(np.sqrt(np.float32(15.0)))
    any_ = C.c_int(-1)
    _lib.check(lib.np_count_mismatch(0, out.ptr, out.ptr, n, 0.0, 0.0, C.byref(any_)))
    assert any_.value == 0
    _lib.check(lib.np_count_mismatch(0, a.ptr, b.ptr, n, 0.0, 0.0, C.byref(any_)))
    assert any_.value == 1
    _lib.check(lib.np_unary(UNARY_OPS["negate"], b.ptr, out.ptr, n, 0.0, 0.0))
    _lib.check(lib.np_unary(UNARY_OPS["negate"], out.ptr, out.ptr, n, 0.0, 0.0))                  # out = b again
    _lib.check(lib.np_count_mismatch(0, out.ptr, b.ptr, n, 0.0, 0.0, C.byref(any_)))
    assert any_.value == 0
    _lib.check(lib.np_memcpy_h2d(out.ptr + 4 * (n - 1), np.float32([10.5]).ctypes.data, 4))     # only the LAST element differs
    _lib.check(lib.np_count_mismatch(0, out.ptr, b.ptr, n, 0.0, 0.0, C.byref(any_)))
    assert any_.value == 1
    _lib.check(lib.np_count_mismatch(1, out.ptr, b.ptr, n, 0.1, 0.0, C.byref(any_)))             # allclose, rtol 0.1: 10.5 vs 10
    assert any_.value == 0
    _lib.check(lib.np_count_mismatch(1, out.ptr, b.ptr, n, 0.01, 0.0, C.byref(any_)))
    assert any_.value == 1
    verdict = C.c_int(-1)
    _lib.check(lib.np_all(b.ptr, n, 0, C.byref(verdict)))
    assert verdict.value == 1
    _lib.check(lib.np_memcpy_h2d(out.ptr + 4 * (n - 1), np.float32([0.0]).ctypes.data, 4))      # one zero at the very end
    _lib.check(lib.np_all(out.ptr, n, 0, C.byref(verdict)))
    assert verdict.value == 0
    _lib.check(lib.np_all(out.ptr, n - 1, 0, C.byref(verdict)))
    assert verdict.value == 1
    # order statistics with 64-bit indices: b is 2.0 everywhere except b[n-1] = 10
    two = (C.c_float * 2)()
    _lib.check(lib.np_order_stat(b.ptr, n, n - 1, two))
    assert list(two) == [10.0, 10.0]
    _lib.check(lib.np_order_stat(b.ptr, n, n - 2, two))
    assert list(two) == [2.0, 10.0]
    _lib.check(lib.np_order_stat(b.ptr, n, 0, two))
    assert list(two) == [2.0, 2.0]
    for buf in (a, b, out):
        buf.free()
    freed = C.c_size_t()
    _lib.check(lib.np_pool_trim(C.byref(freed)))        # hand the 26 GB back to the driver


def test_device_copy_and_fill_are_bit_exact(hip):
    """Large np_memcpy_d2d goes through the library's copy kernel (np::device_copy), small / odd ones
    through hipMemcpyAsync: every bit pattern (NaN payloads, -0, denormals) must survive, at any
    word alignment; np_fill writes exactly n elements."""
    from numpower_amd import _lib
    lib = _lib.load()
    n = 5_000_003
    bits = (np.arange(n, dtype=np.uint64) * 2654435761 % (1 << 32)).astype(np.uint32)   # all kinds of patterns
    bits[:4] = [0x7fc00001, 0xffc12345, 0x80000000, 0x00000001]
    src = _lib.DeviceBuffer(4 * (n + 8)); dst = _lib.DeviceBuffer(4 * (n + 8))
    for off_s, off_d, cnt in ((0, 0, n), (1, 0, n), (0, 3, n), (2, 1, n - 5), (0, 0, 1000), (1, 1, 3)):
        _lib.check(lib.np_memset0(dst.ptr, 4 * (n + 8)))
        _lib.check(lib.np_memcpy_h2d(src.ptr + 4 * off_s, bits.ctypes.data, 4 * cnt))
        _lib.check(lib.np_memcpy_d2d(dst.ptr + 4 * off_d, src.ptr + 4 * off_s, 4 * cnt))
        back = np.empty(n + 8, np.uint32)
        _lib.check(lib.np_memcpy_d2h(back.ctypes.data, dst.ptr, 4 * (n + 8)))
        assert (back[off_d:off_d + cnt] == bits[:cnt]).all()
        assert (back[:off_d] == 0).all() and (back[off_d + cnt:] == 0).all()
    for off, cnt in ((0, n), (1, n), (3, n - 2), (2, 5)):
        _lib.check(lib.np_memset0(dst.ptr, 4 * (n + 8)))
        _lib.check(lib.np_fill(dst.ptr + 4 * off, -0.0, cnt))
        back = np.empty(n + 8, np.uint32)
        _lib.check(lib.np_memcpy_d2h(back.ctypes.data, dst.ptr, 4 * (n + 8)))
        assert (back[off:off + cnt] == 0x80000000).all() and (back[:off] == 0).all() and (back[off + cnt:] == 0).all()
    src.free(); dst.free()


def test_rows_of_2_gib_and_more(hip):
    """(3, 600 000 000): 7.2 GB, rows of 2.4 GB.  The reference's struct keeps byte strides as `int` (ndarray.h:52-74) and
    Generate_Strides wraps there (initializers.c:115-135: `shape[i + 1] * strides[i + 1]`): $a[1] would point outside the buffer.
    The hot path never reads strides (contiguous-only contract) and works; the host mirror marks the stride it cannot store,
    takes $a[i] from the extents, and refuses the strided views (slice, diagonal) out loud instead of reading somewhere else."""
    from numpower_amd.ndarray import GPU, Error, NDArray as nd
    rows, cols = 3, 600_000_000
    a = nd.full([rows, cols], 1.0, GPU)
    a[1].fill(2.0)                                   # through the view: row 1 only
    a[2].fill(-3.0)
    assert [nd.sum(a[i]) / cols for i in range(rows)] == [1.0, 2.0, -3.0]
    assert nd.max(a) == 2.0 and nd.min(a) == -3.0
    b = a * 2.0 + 1.0                                # elementwise: strides are never read
    assert [nd.max(b[i]) for i in range(rows)] == [3.0, 5.0, -5.0] and [nd.min(b[i]) for i in range(rows)] == [3.0, 5.0, -5.0]
    s0 = nd.sum(a, 0)                                # axis reductions take their extents from the shape
    assert s0.shape() == [cols] and nd.max(s0) == 0.0 and nd.min(s0) == 0.0
    s1 = nd.sum(a, 1).cpu().toArray()
    assert [v / cols for v in s1] == [1.0, 2.0, -3.0]
    t = nd.transpose(a)                              # (600 000 000, 3): its own strides fit
    assert t.shape() == [cols, rows] and t[cols - 1].cpu().toArray() == [1.0, 2.0, -3.0]
    with pytest.raises(Error, match="2 GiB"):
        a.slice([0, 2], [0, 10, 2])
    del a, b, s0, t


def test_allocation_failure_is_an_error(hip):
    """vmalloc's failure text (gpu_alloc.c:15 "device memory allocation failed") for a request no device can hold — as a status of
    np_malloc and as the exception of the array constructors — and the library usable afterwards.  (Requests a little above the
    288 GB are no test of this: the driver of this pool backs them with host memory.)"""
    import ctypes as C
    from numpower_amd import _lib
    from numpower_amd.ndarray import GPU, Error, NDArray as nd
    lib = _lib.load()
    p = C.c_void_p()
    rc = lib.np_malloc(C.byref(p), 1 << 44)                                   # 16 TiB
    assert rc == -2 and p.value is None                                       # NP_ERR_ALLOC
    assert lib.np_last_error().decode().startswith("device memory allocation failed")
    with pytest.raises(Error, match="device memory allocation failed"):
        nd.zeros([1 << 21, 1 << 21], GPU)                                     # 2^42 floats = 16 TiB
    with pytest.raises(Error, match="device memory allocation failed"):
        nd.full([1 << 21, 1 << 21], 1.0, GPU)
    x = nd.full([1000], 2.0, GPU)
    assert nd.sum(x * x) == 4000.0                                            # still in business
    assert lib.np_sync() == 0


def test_the_cache_goes_back_before_a_request_that_does_not_fit_next_to_it(hip):
    """The caching pool keeps freed blocks for reuse, and the driver of this pool does not refuse a request that no longer fits the device
    — it backs it with host memory.  A cache that is given back only when hipMalloc fails would let a long-lived process grow into host
    memory: np_malloc asks the driver for the device's free memory before a large cache miss and gives its cached blocks back first when
    they are in the way.  Nothing large is touched here (two 4 MB fills), and with the rule in place the process never holds more than
    160 GB at once."""
    import ctypes as C
    from numpower_amd import _lib
    lib = _lib.load()
    GB = 1 << 30
    trimmed = C.c_size_t()
    _lib.check(lib.np_pool_trim(C.byref(trimmed)))
    base = lib.np_pool_reserved_bytes()
    a = _lib.DeviceBuffer(160 * GB)
    a.free()                                                                  # cached: still reserved at the driver
    assert lib.np_pool_reserved_bytes() >= base + 160 * GB
    b = _lib.DeviceBuffer(150 * GB)                                           # 160 + 150 > 288: the cache must go back FIRST
    assert lib.np_pool_reserved_bytes() < base + 155 * GB, "the request was placed next to the cache (on this pool: in host memory)"
    n = 1 << 20
    _lib.check(lib.np_fill(b.ptr, 7.0, n))
    _lib.check(lib.np_fill(b.ptr + 150 * GB - 4 * n, 9.0, n))                 # both ends are there
    v = C.c_float()
    _lib.check(lib.np_read_float(b.ptr, n - 1, C.byref(v)))
    assert v.value == 7.0
    _lib.check(lib.np_read_float(b.ptr + 150 * GB - 4 * n, n - 1, C.byref(v)))
    assert v.value == 9.0
    # a request that DOES fit next to the cache leaves the cache alone
    b.free()
    c = _lib.DeviceBuffer(40 * GB)
    assert lib.np_pool_reserved_bytes() >= base + 190 * GB                    # 150 cached + 40 live
    c.free()
    _lib.check(lib.np_pool_trim(C.byref(trimmed)))
    assert trimmed.value >= 190 * GB and lib.np_pool_reserved_bytes() == base
