"""The --with-hip glue on the GPU, called BY THE REFERENCE'S NAMES through ctypes and checked against
the oracle:

  part 1  libnp_hipmath.so alone — vmalloc / vmemcpyh2d / cuda_* / NDArray_VFLOATF_I / vfree, the raw
          pointer ABI of src/gpu_alloc.h + src/ndmath/cuda/cuda_math.h, no host layer loaded;
  part 2  libnumpower_host.so — NDArrayMathGPU_ElementWise(nda, cuda_float_sin) with the function
          POINTER, as numpower.c:1651 calls it: recognised pointers run out of place in one pass,
          unknown pointers get the reference's copy + in-place call.

Bars as in test_gpu_parity.py: bit-exact for exact ops (incl. the CPU body/tail quirks the glue asks
for), <= 1e-5 relative for libm-class ops."""
import ctypes as C
from pathlib import Path

import numpy as np
import pytest

from numpower_amd import synth
from tests.test_ext_glue_cpu import UNARY
from tests.test_gpu_parity import DOMAIN, EXACT_UNARY, assert_bit_equal, assert_close

pytestmark = pytest.mark.gpu

LIBDIR = Path(__file__).resolve().parent.parent / "numpower_amd" / "lib"
fp = C.POINTER(C.c_float)


class Glue:
    """ctypes view of libnp_hipmath.so with device buffers managed through vmalloc / vfree."""

    def __init__(self):
        self.hip = C.CDLL(str(LIBDIR / "libnp_hip.so"), mode=C.RTLD_GLOBAL)
        self.lib = C.CDLL(str(LIBDIR / "libnp_hipmath.so"))
        self.lib.np_ext_last_error.restype = C.c_char_p
        self.lib.NDArray_VFLOATF_I.restype = C.c_float
        self.lib.NDArray_VFLOATF_I.argtypes = [C.c_void_p, C.c_int]
        self.lib.NDArray_VFLOAT.restype = C.c_float
        self.lib.NDArray_VFLOAT.argtypes = [C.c_void_p]
        self.lib.cuda_max_float.restype = C.c_float
        self.lib.cuda_min_float.restype = C.c_float
        self.hip.np_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        assert self.hip.np_init(0) == 0

    def ok(self):
        msg = self.lib.np_ext_last_error()
        assert not msg, msg
        return True

    def put(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        p = C.c_void_p()
        self.lib.vmalloc(C.byref(p), C.c_uint(max(arr.nbytes, 4)))
        self.ok()
        if arr.nbytes:
            self.lib.vmemcpyh2d(arr.ctypes.data_as(C.c_char_p), C.cast(p, C.c_char_p), C.c_uint(arr.nbytes))
        self.ok()
        return p

    def empty(self, n):
        p = C.c_void_p()
        self.lib.vmalloc(C.byref(p), C.c_uint(max(4 * n, 4)))
        self.ok()
        return p

    def get(self, p, shape):
        out = np.empty(shape, dtype=np.float32)
        if out.size:
            assert self.hip.np_memcpy_d2h(out.ctypes.data, p, out.nbytes) == 0
        return out

    def free(self, *ps):
        for p in ps:
            self.lib.vfree(p)
        self.ok()


@pytest.fixture(scope="module")
def glue():
    g = Glue()
    g.lib.np_ext_clear_error()
    yield g
    assert g.lib.np_ext_device_allocs() == 0, "vmalloc / vfree imbalance (vmemcheck would report a leak)"


def _unary_input(name, n, seed):
    lo, hi = DOMAIN.get(name, (-10.0, 10.0))
    return synth.uniform((n,), seed, lo, hi)


@pytest.mark.parametrize("name", UNARY)
def test_cuda_float_unary_in_place(name, glue, oracle):
    n = 100_003                                   # vector body + a ragged tail
    x = _unary_input(name, n, 40 + UNARY.index(name))
    d = glue.put(x)
    getattr(glue.lib, "cuda_float_" + name)(C.c_int(n), d)
    glue.ok()
    got = glue.get(d, (n,))
    ref = oracle.unary(name, x)
    if name in EXACT_UNARY:
        assert_bit_equal(got, ref, "cuda_float_" + name)
    else:
        assert_close(got, ref, "cuda_float_" + name)
    glue.free(d)


def test_cuda_float_clip_round_arctan2(glue, oracle):
    n = 77_777
    x = synth.uniform((n,), 3, -50.0, 50.0)
    y = synth.uniform((n,), 4, -50.0, 50.0)
    d = glue.put(x)
    glue.lib.cuda_float_clip(C.c_int(n), d, C.c_float(-7.25), C.c_float(11.5))
    assert_bit_equal(glue.get(d, (n,)), oracle.unary("clip", x, -7.25, 11.5), "cuda_float_clip")
    glue.free(d)
    d = glue.put(x)
    glue.lib.cuda_float_round(C.c_int(n), d, C.c_float(2.0))
    assert_bit_equal(glue.get(d, (n,)), oracle.unary("round", x, 2.0), "cuda_float_round")
    glue.free(d)
    d, dy = glue.put(x), glue.put(y)
    glue.lib.cuda_float_arctan2(C.c_int(n), d, dy)        # in place on d: d[i] = atan2f(d[i], y[i])
    assert_close(glue.get(d, (n,)), np.arctan2(x.astype(np.float64), y.astype(np.float64)), "cuda_float_arctan2")
    assert_bit_equal(glue.get(dy, (n,)), y, "arctan2 must not touch y")
    glue.free(d, dy)


def _operands(n, seed):
    a = synth.uniform((n,), seed, -4.0, 4.0)
    b = synth.uniform((n,), seed + 1, -4.0, 4.0)
    a[::7] = 0.0
    a[3::11] = -0.0
    b[5::13] = np.float32(-2.5)
    b[b == 0] = np.float32(1.0)
    return a, b


@pytest.mark.parametrize("n", [1, 7, 8, 1001, 262_147])
@pytest.mark.parametrize("op", ["add", "subtract", "multiply", "divide", "mod", "pow"])
def test_cuda_binary_float_matches_the_cpu_path(op, n, glue, oracle):
    """incl. multiply's zero-sign and mod's floor/fmodf body/tail split (arithmetics.c:397-412,788-800)."""
    a, b = _operands(n, 17)
    if op == "pow":
        a = np.abs(a) + np.float32(0.25)
    da, db, dr = glue.put(a), glue.put(b), glue.empty(n)
    getattr(glue.lib, "cuda_%s_float" % op)(C.c_int(n), da, db, dr, C.c_int(n))
    glue.ok()
    got = glue.get(dr, (n,))
    ref = oracle.binary(op, a, b)
    if op == "pow":
        assert_close(got, ref, "cuda_pow_float")
    else:
        assert_bit_equal(got, ref.reshape(-1), "cuda_%s_float n=%d" % (op, n))
    glue.free(da, db, dr)


@pytest.mark.parametrize("op,fn", [("equal", "equal"), ("not_equal", "not_equal"), ("greater", "greater"),
                                   ("greater_equal", "greater_equal"), ("less", "less"), ("less_equal", "less_equal")])
def test_cuda_float_compare(op, fn, glue, oracle):
    n = 50_005
    a, b = _operands(n, 23)
    b[::5] = a[::5]                                  # equal elements
    b[1::9] = a[1::9] + np.float32(5e-8)             # inside the scalar tail's 1e-7 tolerance
    da, db, dr = glue.put(a), glue.put(b), glue.empty(n)
    getattr(glue.lib, "cuda_float_compare_" + fn)(C.c_int(n), da, db, dr, C.c_int(n))
    glue.ok()
    assert_bit_equal(glue.get(dr, (n,)), oracle.binary(op, a, b).reshape(-1), "cuda_float_compare_" + fn)
    glue.free(da, db, dr)


def test_cuda_equal_sum_prod_min_max_fill_vfloat(glue, oracle):
    n = 300_001
    a = synth.uniform((n,), 31, -1.0, 1.0)
    da, db = glue.put(a), glue.put(a)
    assert glue.lib.cuda_equal_float(C.c_int(n), da, db, C.c_int(n)) == 1
    b = a.copy()
    b[n - 2] = np.nextafter(b[n - 2], np.float32(9))
    dc = glue.put(b)
    assert glue.lib.cuda_equal_float(C.c_int(n), da, dc, C.c_int(n)) == 0
    # reductions hand a HOST float back; sum accumulates onto the caller's initial value (0), prod onto 1
    v = C.c_float(0.0)
    glue.lib.cuda_sum_float(C.c_int(n), da, C.byref(v), C.c_int(n))
    want = float(a.astype(np.float64).sum())
    assert abs(v.value - want) <= 1e-5 * float(np.abs(a).astype(np.float64).sum())
    f = np.float32(1.0) + synth.uniform((4096,), 5, -1e-3, 1e-3)
    df = glue.put(f)
    v = C.c_float(1.0)
    glue.lib.cuda_prod_float(C.c_int(f.size), df, C.byref(v), C.c_int(f.size))
    assert abs(v.value - float(np.prod(f.astype(np.float64)))) <= 1e-5 * abs(float(np.prod(f.astype(np.float64))))
    assert glue.lib.cuda_max_float(da, C.c_int(n)) == a.max()
    assert glue.lib.cuda_min_float(da, C.c_int(n)) == a.min()
    assert glue.lib.NDArray_VFLOATF_I(da, 12345) == a[12345]
    assert glue.lib.NDArray_VFLOAT(da) == a[0]
    glue.lib.cuda_fill_float(dc, C.c_float(-3.5), C.c_int(n))
    assert (glue.get(dc, (n,)) == np.float32(-3.5)).all()
    # vmemcpyd2d(source, destination, bytes): the reference's argument order
    glue.lib.vmemcpyd2d(C.cast(da, C.c_char_p), C.cast(dc, C.c_char_p), C.c_uint(4 * n))
    assert_bit_equal(glue.get(dc, (n,)), a, "vmemcpyd2d")
    glue.free(da, db, dc, df)


def test_cuda_matvec_outer_transpose(glue, oracle):
    rows, cols = 300, 517
    A = synth.uniform((rows, cols), 61, -1.0, 1.0)
    x = synth.uniform((cols,), 62, -1.0, 1.0)
    dA, dx, dy = glue.put(A), glue.put(x), glue.empty(rows)
    glue.lib.cuda_float_multiply_matrix_vector(C.c_int(cols), dA, dx, dy, C.c_int(rows), C.c_int(cols))
    glue.ok()
    want = A.astype(np.float64) @ x.astype(np.float64)
    scale = np.abs(A).astype(np.float64) @ np.abs(x).astype(np.float64)
    assert (np.abs(glue.get(dy, (rows,)) - want) <= 1e-5 * scale).all()
    u, v = synth.uniform((211,), 63, -2.0, 2.0), synth.uniform((97,), 64, -2.0, 2.0)
    du, dv, dr = glue.put(u), glue.put(v), glue.empty(211 * 97)
    glue.lib.cuda_calculate_outer_product(C.c_int(211), C.c_int(97), du, dv, dr)
    assert_bit_equal(glue.get(dr, (211, 97)), oracle.outer(u, v), "cuda_calculate_outer_product")
    # transpose: out of place, and IN PLACE the way manipulation.c:124 calls it, well past 256 x 256
    dT = glue.empty(rows * cols)
    glue.lib.cuda_float_transpose(C.c_int(32), C.c_int(8), dA, dT, C.c_int(cols), C.c_int(rows))
    assert_bit_equal(glue.get(dT, (cols, rows)), A.T, "cuda_float_transpose")
    glue.lib.cuda_float_transpose(C.c_int(32), C.c_int(8), dA, dA, C.c_int(cols), C.c_int(rows))
    assert_bit_equal(glue.get(dA, (cols, rows)), A.T, "cuda_float_transpose in place")
    glue.ok()
    glue.free(dA, dx, dy, du, dv, dr, dT)


# ---------------------------------------------------------------------------------------------------
# part 2: the drivers with the reference's signature, through the host library
# ---------------------------------------------------------------------------------------------------

def _host():
    from numpower_amd.ndarray import NDArray, _fn, _load_host
    return NDArray, _fn, _load_host()


@pytest.mark.parametrize("name", UNARY)
def test_driver_takes_the_function_pointer(name, hip, oracle):
    """NDArrayMathGPU_ElementWise(nda, cuda_float_<name>) — numpower.c:1651-3348 — out of place: the input
    array keeps its values and the live-allocation count grows by exactly the result."""
    NDArray, _fn, h = _host()
    x = _unary_input(name, 257 * 255, 70 + UNARY.index(name)).reshape(257, 255)
    g = NDArray.array(x).gpu()
    before = h.NDArray_LiveDeviceAllocations()
    r = NDArray._wrap(h.NDArrayMathGPU_ElementWise(g._p, _fn(h, "cuda_float_" + name)))
    assert h.NDArray_LiveDeviceAllocations() == before + 1        # no hidden copy
    got = r.cpu().numpy()
    ref = oracle.unary(name, x)
    (assert_bit_equal if name in EXACT_UNARY else assert_close)(got, ref, name)
    assert_bit_equal(g.cpu().numpy(), x, "input untouched")
    assert r.shape() == [257, 255]


def test_driver_unknown_pointer_gets_copy_plus_in_place_call(hip, oracle):
    """A caller's own `void op(int, float*)` (here: a C callback that negates through cuda_float_negate and
    then takes exp through cuda_float_exp) is not recognised -> NDArray_Copy + op(n, data), cuda_math.cu:1532-1537."""
    NDArray, _fn, h = _host()
    x = synth.uniform((4099,), 9, -3.0, 3.0)
    g = NDArray.array(x).gpu()
    calls = []

    @C.CFUNCTYPE(None, C.c_int, C.c_void_p)
    def my_op(n, data):
        calls.append(n)
        h.cuda_float_negate(C.c_int(n), C.c_void_p(data))
        h.cuda_float_exp(C.c_int(n), C.c_void_p(data))

    r = NDArray._wrap(h.NDArrayMathGPU_ElementWise(g._p, C.cast(my_op, C.c_void_p)))
    assert calls == [4099]
    assert_close(r.cpu().numpy(), oracle.unary("exp", oracle.unary("negate", x)), "copy + in-place fallback")
    assert_bit_equal(g.cpu().numpy(), x, "input untouched")


def test_driver_1f_2f_1n(hip, oracle):
    NDArray, _fn, h = _host()
    x = synth.uniform((1000, 33), 11, -20.0, 20.0)
    y = synth.uniform((1000, 33), 12, -20.0, 20.0)
    g, gy = NDArray.array(x).gpu(), NDArray.array(y).gpu()
    r = NDArray._wrap(h.NDArrayMathGPU_ElementWise2F(g._p, _fn(h, "cuda_float_clip"), -2.5, 7.0))
    assert_bit_equal(r.cpu().numpy(), oracle.unary("clip", x, -2.5, 7.0), "clip")
    r = NDArray._wrap(h.NDArrayMathGPU_ElementWise1F(g._p, _fn(h, "cuda_float_round"), 1.0))
    assert_bit_equal(r.cpu().numpy(), oracle.unary("round", x, 1.0), "round")
    r = NDArray._wrap(h.NDArrayMathGPU_ElementWise1N(g._p, _fn(h, "cuda_float_arctan2"), gy._p))
    assert_close(r.cpu().numpy(), np.arctan2(x.astype(np.float64), y.astype(np.float64)), "arctan2")
    assert_bit_equal(g.cpu().numpy(), x, "input untouched")


def test_driver_refuses_cpu_arrays_with_an_error(hip):
    from numpower_amd.ndarray import Error
    NDArray, _fn, h = _host()
    c = NDArray.array(np.ones((3, 3), dtype=np.float32))          # on the CPU
    with pytest.raises(Error, match="operand is on the CPU"):
        NDArray._wrap(h.NDArrayMathGPU_ElementWise(c._p, _fn(h, "cuda_float_sin")))


def test_a_count_past_int_max_is_refused_not_treated_as_empty(glue):
    """The reference's cuda_* interface counts in `int` and its callers pass NDArray_NUMELEMENTS (a long): 2^31 elements and more
    arrive as a negative count.  The glue raises instead of filling / summing nothing: nd::sum() of such an array must not be 0."""
    x = glue.put(np.float32([1.0, 2.0, 3.0, 4.0]))
    wrapped = C.c_int(-(1 << 31) + 4)                         # what (int)(2^31 + 4) is
    glue.lib.cuda_fill_float(x, C.c_float(9.0), wrapped)
    assert b"does not fit the int" in glue.lib.np_ext_last_error()
    assert glue.get(x, (4,)).tolist() == [1.0, 2.0, 3.0, 4.0]           # untouched
    glue.lib.np_ext_clear_error()
    total = C.c_float(0.0)
    glue.lib.cuda_sum_float(C.c_int(1), x, C.byref(total), wrapped)
    assert b"does not fit the int" in glue.lib.np_ext_last_error()
    glue.lib.np_ext_clear_error()
    glue.lib.cuda_float_exp(wrapped, x)
    assert b"does not fit the int" in glue.lib.np_ext_last_error()
    glue.lib.np_ext_clear_error()
    glue.free(x)
