"""ctypes front end of oracle/np_oracle.c — the CPU restatement of the reference's hot path.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under numpower_amd/ imports this module.

The functions take and return numpy float32 arrays; shapes carry the reference's ndim semantics
(0-d arrays are the reference's 0-d scalars).
"""
from __future__ import annotations

import ctypes as C
import glob
import os
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB = HERE / "lib" / "libnp_oracle.so"

BINARY = {"add": 0, "subtract": 1, "multiply": 2, "divide": 3, "mod": 4, "pow": 5, "arctan2": 6,
          "equal": 7, "not_equal": 8, "greater": 9, "greater_equal": 10, "less": 11, "less_equal": 12, "maximum": 13, "minimum": 14}
UNARY = {name: i for i, name in enumerate([
    "abs", "sqrt", "exp", "exp2", "expm1", "log", "log2", "log10", "log1p", "logb",
    "sin", "cos", "tan", "arcsin", "arccos", "arctan", "degrees", "radians",
    "sinh", "cosh", "tanh", "arcsinh", "arccosh", "arctanh",
    "rint", "fix", "floor", "ceil", "trunc", "sinc", "negate", "sign",
    "clip", "round", "rsqrt", "positive", "reciprocal"])}
REDUCE = {"sum": 0, "prod": 1, "min": 2, "max": 3, "mean": 4}


class OracleError(RuntimeError):
    """The message the reference would pass to zend_throw_error."""


_lib = None
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)


def build():
    import sys
    sys.path.insert(0, str(HERE.parent))
    from numpower_amd.build import build_oracle
    return build_oracle()


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not LIB.exists():
        build()
    lib = C.CDLL(str(LIB))
    lib.oracle_last_error.restype = C.c_char_p
    lib.oracle_map.restype = C.c_long
    lib.oracle_map.argtypes = [C.c_int, _fp, _fp, C.c_long, C.c_float, C.c_float]
    lib.oracle_binary.restype = C.c_int
    lib.oracle_binary.argtypes = [C.c_int, _fp, _ip, C.c_int, _fp, _ip, C.c_int,
                                  C.POINTER(_fp), _ip, _ip]
    lib.oracle_free.argtypes = [C.c_void_p]
    lib.oracle_average_weighted.restype = C.c_float
    lib.oracle_average_weighted.argtypes = [_fp, _fp, C.c_long]
    for name in ("oracle_sum", "oracle_prod", "oracle_min", "oracle_max", "oracle_mean", "oracle_all",
                 "oracle_variance", "oracle_std"):
        fn = getattr(lib, name)
        fn.restype = C.c_float
        fn.argtypes = [_fp, C.c_long]
    lib.oracle_reduce_axis.restype = C.c_int
    lib.oracle_reduce_axis.argtypes = [C.c_int, _fp, _ip, C.c_int, C.c_int, _fp]
    lib.oracle_set_blas.restype = C.c_int
    lib.oracle_set_blas.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
    lib.oracle_blas_kind.restype = C.c_int
    lib.oracle_matmul.argtypes = [C.c_int, C.c_int, C.c_int, _fp, _fp, _fp]
    lib.oracle_matvec.argtypes = [C.c_int, C.c_int, _fp, _fp, _fp]
    lib.oracle_matmul_check.restype = C.c_int
    lib.oracle_matmul_check.argtypes = [_ip, C.c_int, _ip, C.c_int]
    lib.oracle_broadcast.restype = C.c_int
    lib.oracle_broadcast.argtypes = [_fp, _ip, C.c_int, _ip, C.c_int, _fp]
    _lib = lib
    return lib


def _f(a):
    return np.asarray(a, dtype=np.float32, order="C")   # (ascontiguousarray would promote 0-d to 1-d)


def _ptr(a):
    return a.ctypes.data_as(_fp)


def _shape(a):
    s = (C.c_int * max(a.ndim, 1))(*a.shape)
    return s


def _err():
    return OracleError(load().oracle_last_error().decode())


# ---------------------------------------------------------------------------------------------

def binary(op: str, a, b) -> np.ndarray:
    """NDArray_{Add,Subtract,Multiply,Divide,Mod,Pow}_Float(a, b) incl. broadcast + AVX quirks."""
    lib = load()
    a, b = _f(a), _f(b)
    out = _fp()
    oshape = (C.c_int * 32)()
    ondim = C.c_int()
    rc = lib.oracle_binary(BINARY[op], _ptr(a), _shape(a), a.ndim, _ptr(b), _shape(b), b.ndim,
                           C.byref(out), oshape, C.byref(ondim))
    if rc != 0:
        raise _err()
    shape = tuple(oshape[i] for i in range(ondim.value))
    n = int(np.prod(shape, dtype=np.int64)) if shape else 1
    res = np.ctypeslib.as_array(out, shape=(n,)).copy().reshape(shape)
    lib.oracle_free(out)
    return res


def unary(op: str, x, p0: float = 0.0, p1: float = 0.0, strict_domain: bool = False) -> np.ndarray:
    """NDArray_Map / Map1F / Map2F with the float_* kernel `op`."""
    x = _f(x)
    out = np.empty_like(x)
    bad = load().oracle_map(UNARY[op], _ptr(x), _ptr(out), x.size, p0, p1)
    if bad and strict_domain:
        raise OracleError("RuntimeError: Invalid argument provided for %s (reference calls exit(1))" % op)
    return out


def reduce_all(op: str, x) -> np.float32:
    x = _f(x)
    fn = getattr(load(), "oracle_" + op)
    return np.float32(fn(_ptr(x), x.size))


def argreduce(x, axis=None, is_max=True) -> np.ndarray:
    """NDArray_ArgMinMaxCommon: float indices; axis None = flattened (axis 128 in the reference)."""
    lib = load()
    lib.oracle_argreduce.argtypes = [C.c_int, _fp, C.c_long, C.c_long, C.c_long, _fp]
    lib.oracle_argreduce.restype = None
    x = _f(x)
    if axis is None:
        outer, length, inner, oshape = 1, x.size, 1, ()
    else:
        axis = axis % x.ndim
        outer = int(np.prod(x.shape[:axis], dtype=np.int64))
        inner = int(np.prod(x.shape[axis + 1:], dtype=np.int64))
        length, oshape = x.shape[axis], x.shape[:axis] + x.shape[axis + 1:]
    out = np.empty(outer * inner, dtype=np.float32)
    lib.oracle_argreduce(1 if is_max else 0, _ptr(x), outer, length, inner, _ptr(out))
    return out.reshape(oshape)


def transpose(x, axes=None) -> np.ndarray:
    """NDArray_Transpose + NDArray_ToContiguous."""
    lib = load()
    lib.oracle_transpose.restype = C.c_int
    lib.oracle_transpose.argtypes = [_fp, _ip, C.c_int, _ip, _fp, _ip]
    x = _f(x)
    out = np.empty(x.size, dtype=np.float32)
    oshape = (C.c_int * max(x.ndim, 1))()
    perm = None if axes is None else (C.c_int * max(len(axes), 1))(*[int(a) for a in axes])
    if axes is not None and len(axes) != x.ndim:
        raise OracleError("axes don't match array")
    if lib.oracle_transpose(_ptr(x), _shape(x), x.ndim, perm, _ptr(out), oshape) != 0:
        raise _err()
    return out.reshape(tuple(oshape[i] for i in range(x.ndim)))


def arange(stop, start=0.0, step=1.0) -> np.ndarray:
    """NDArray::arange(stop, start, step) -> NDArray_Arange(start, stop, step) (numpower.c:1284-1302,
    initializers.c:818-841): length = ceil((stop - start) / step), then the float recurrence."""
    import math
    length = math.ceil((stop - start) / step)
    if not (-2 ** 31 <= length <= 2 ** 31 - 1):
        raise OracleError("arange: overflow while computing length")
    if length <= 0:
        raise OracleError("arange: zero length")
    lib = load()
    lib.oracle_arange.restype = None
    lib.oracle_arange.argtypes = [_fp, C.c_double, C.c_double, C.c_long]
    out = np.empty(int(length), dtype=np.float32)
    lib.oracle_arange(_ptr(out), float(start), float(step), int(length))
    return out


def identity(size: int) -> np.ndarray:
    """NDArray_Identity (initializers.c:479-510)."""
    if size < 0:
        raise OracleError("negative dimensions are not allowed")
    return np.eye(size, dtype=np.float32) if size else np.empty((0,), np.float32)


def full(shape, value) -> np.ndarray:
    """NDArray_Full / NDArray_Ones / NDArray_Zeros (initializers.c:379-470,655-660)."""
    return np.full(tuple(int(v) for v in shape), np.float32(value), dtype=np.float32)


def array_equal(a, b) -> int:
    """NDArray_ArrayEqual (logic.c:703-716)."""
    a, b = _f(a), _f(b)
    if a.shape != b.shape:
        return 0
    lib = load()
    lib.oracle_array_equal.restype = C.c_int
    lib.oracle_array_equal.argtypes = [_fp, _fp, C.c_long]
    return int(lib.oracle_array_equal(_ptr(a), _ptr(b), a.size))


def allclose(a, b, rtol: float = 1e-05, atol: float = 1e-08) -> int:
    """NDArray_AllClose (logic.c:750-772) / float_allclose, per-element meaning (see np_oracle.c)."""
    a, b = _f(a), _f(b)
    if a.shape != b.shape:
        raise OracleError("Shape mismatch")
    lib = load()
    lib.oracle_allclose.restype = C.c_int
    lib.oracle_allclose.argtypes = [_fp, _fp, C.c_long, C.c_float, C.c_float]
    return int(lib.oracle_allclose(_ptr(a), _ptr(b), a.size, rtol, atol))


# ---- manipulation wrappers: index bookkeeping only, numpy restatements of manipulation.c:554-1073 ----
def atleast_3d(a):
    """NDArray_AtLeast3D (manipulation.c:592-615): (1, n, 1) below 2-d as the reference; numpy's (r, c, 1) for a
    2-d input, where the reference overflows its two-int shape buffer."""
    return np.atleast_3d(_f(a))


def column_stack(arrays):
    """NDArray_ColumnStack (manipulation.c:1055-1073): atleast_2d + transpose of EVERY input (2-d inputs too,
    unlike numpy), concatenated along axis 1."""
    return np.concatenate([np.atleast_2d(_f(a)).T for a in arrays], axis=1)


def diag(a):
    """NDArray_Diag (initializers.c:597-625)."""
    a = _f(a)
    if a.ndim not in (1, 2):
        raise OracleError("Input array must be a vector or 2-dimensional")
    return np.diag(a).astype(np.float32) if a.ndim == 1 else np.ascontiguousarray(np.diagonal(a)[:min(a.shape)])


def median(a, with_stats: bool = False):
    """calculate_median (arithmetics.c:111-138) on the flattened array."""
    a = _f(a).reshape(-1)
    lib = load()
    lib.oracle_median.restype = C.c_float
    lib.oracle_median.argtypes = [_fp, C.c_long, _fp]
    stats = np.zeros(2, np.float32)
    v = np.float32(lib.oracle_median(_ptr(a), a.size, _ptr(stats)))
    return (v, stats) if with_stats else v


def quantile(a, q: float, with_stats: bool = False):
    """NDArray_Quantile (statistics.c:60-79): `q` a scalar in [0, 1], the array flattened."""
    a = _f(a).reshape(-1)
    if q < 0 or q > 1:
        raise OracleError("Q must be between 0 and 1")
    lib = load()
    lib.oracle_quantile.restype = C.c_float
    lib.oracle_quantile.argtypes = [_fp, C.c_long, C.c_float, _fp]
    stats = np.zeros(2, np.float32)
    v = np.float32(lib.oracle_quantile(_ptr(a), a.size, float(q), _ptr(stats)))
    return (v, stats) if with_stats else v


# ---- views / layout: index bookkeeping only (bit-exact by construction), numpy restatements ----

def reshape(x, shape) -> np.ndarray:
    """NDArray_Reshape (manipulation.c:138-162): same buffer, new shape."""
    x = _f(x)
    total = 1
    for v in shape:
        total *= int(v)
    if total != x.size:
        raise OracleError("incompatible shape in reshape call.")
    return x.reshape(tuple(int(v) for v in shape))


def flatten(x) -> np.ndarray:
    """NDArray_Flatten (manipulation.c:169-184): 1-D copy (a 0-d input becomes [x])."""
    return _f(x).reshape(-1).copy()


def expand_dims(x, axis) -> np.ndarray:
    """NDArray_ExpandDim (manipulation.c:453-513): axes normalised against the OUTPUT rank
    (check_and_adjust_axis, manipulation.c:41-54), remaining slots take x's dimensions in order."""
    x = _f(x)
    axes = [int(a) for a in (axis if isinstance(axis, (list, tuple)) else [axis])]
    out_ndim = len(axes) + x.ndim
    norm = []
    for a in axes:
        if a < -out_ndim or a >= out_ndim:
            raise OracleError("invalid axis or axes provided.")
        norm.append(a + out_ndim if a < 0 else a)
    shape, it = [], 0
    for ax in range(out_ndim):
        if ax in norm:
            shape.append(1)
        else:
            if it >= x.ndim:
                raise OracleError("invalid axis or axes provided.")
            shape.append(x.shape[it])
            it += 1
    return reshape(x, shape)


def append(a, b) -> np.ndarray:
    """NDArray_Append(axis = -1) -> NDArray_ConcatenateFlat (manipulation.c:293-374)."""
    return np.concatenate([_f(a).reshape(-1), _f(b).reshape(-1)])


def diagonal(x) -> np.ndarray:
    """NDArray_Diagonal (indexing.c:21-48): element i at byte offset i*(strides[0]+strides[1]);
    the reference takes shape[1] of them (reads past the buffer when rows < cols) — min(rows, cols)
    is the same wherever that is defined."""
    x = _f(x)
    if x.ndim != 2:
        raise OracleError("NDArray_Diagonal: Array must be 2-d.")
    n = min(x.shape)
    flat = x.reshape(-1)
    return flat[np.arange(n) * (x.shape[1] + 1)].copy()


def trace(x) -> np.float32:
    """NDArray_Trace (linalg.c:758-767): NDArray_Sum_Float of the diagonal."""
    return reduce_all("sum", diagonal(x))


def slice_indices(length: int, index):
    """Slice_GetIndices (indexing.c:59-108) -> (start, step, count)."""
    index = [int(v) for v in index]
    step = index[2] if len(index) == 3 else 1
    if step == 0:
        raise OracleError("slice step cannot be zero")
    if len(index) >= 1:
        start = index[0]
        if start < 0:
            start += length
        if start < 0:
            start = -1 if step < 0 else 0
        if start >= length:
            start = length - 1 if step < 0 else length
    else:
        start = length - 1 if step < 0 else 0
    if len(index) >= 2:
        stop = index[1]
        if stop < 0:
            stop += length
        if stop < 0:
            stop = -1
        if stop > length:
            stop = length
    else:
        stop = -1 if step < 0 else length
    if (step < 0 and stop >= start) or (step > 0 and start >= stop):
        count = 0
    elif step < 0:
        count = int((stop - start + 1) / step) + 1     # C integer division truncates
    else:
        count = int((stop - start - 1) / step) + 1
    if count <= 0:
        count, step, start = 0, 1, 0
    return start, step, count


def slice_(x, *indices) -> np.ndarray:
    """NDArray_Slice (manipulation.c:193-283) + NDArray_ToContiguous: one-element indexes pick and
    drop the axis, [start, stop(, step)] select a range."""
    x = _f(x)
    if len(indices) > x.ndim:
        raise OracleError("too many indices for array.")
    sel = []
    for i, index in enumerate(indices):
        index = list(index) if isinstance(index, (list, tuple)) else [index]
        start, step, count = slice_indices(x.shape[i], index)
        if len(index) == 1:
            sel.append(start)
        else:
            sel.append(start + step * np.arange(count))
    out = x
    for axis in range(len(sel) - 1, -1, -1):       # last axis first so earlier axes keep their position
        out = np.take(out, sel[axis], axis=axis)
    return np.ascontiguousarray(out, dtype=np.float32)


def average_weighted(a, w) -> np.float32:
    a, w = _f(a), _f(w)
    return np.float32(load().oracle_average_weighted(_ptr(a), _ptr(w), a.size))


def reduce_axis(op: str, x, axis: int) -> np.ndarray:
    """reduce(x, &axis, Add|Multiply) (+ Divide for mean)."""
    x = _f(x)
    out = np.empty(x.shape[:axis] + x.shape[axis + 1:], dtype=np.float32) if 0 <= axis < x.ndim \
        else np.empty((), dtype=np.float32)
    rc = load().oracle_reduce_axis(REDUCE[op], _ptr(x), _shape(x), x.ndim, axis, _ptr(out))
    if rc != 0:
        raise _err()
    return out


def find_openblas():
    """Locate an OpenBLAS with the CBLAS interface: (path, symbol_prefix) or None.
    The reference links whatever -lcblas/-lopenblas the host has (config.m4:67-87)."""
    cands = []
    for pat, prefix in (("/usr/lib/x86_64-linux-gnu/libopenblas*.so*", ""),
                        ("/usr/lib64/libopenblas*.so*", ""),
                        ("/usr/local/lib/python3*/dist-packages/scipy.libs/libscipy_openblas-*.so", "scipy_"),
                        ("/usr/lib/python3*/site-packages/scipy.libs/libscipy_openblas-*.so", "scipy_")):
        for p in sorted(glob.glob(pat)):
            cands.append((p, prefix))
    try:
        import scipy
        base = Path(scipy.__file__).resolve().parent.parent / "scipy.libs"
        for p in sorted(base.glob("libscipy_openblas-*.so")):
            cands.append((str(p), "scipy_"))
    except Exception:
        pass
    return cands[0] if cands else None


_blas_info = None


def use_openblas(threads: int | None = None):
    """Attach the oracle's matmul to an OpenBLAS found on this host; returns a description."""
    global _blas_info
    if _blas_info is not None:
        return _blas_info
    found = find_openblas()
    if not found:
        _blas_info = {"kind": "builtin-plain-sgemm", "threads": 1}
        return _blas_info
    path, prefix = found
    t = threads or os.cpu_count() or 1
    if load().oracle_set_blas(path.encode(), prefix.encode(), t) != 0:
        _blas_info = {"kind": "builtin-plain-sgemm", "threads": 1, "error": _err().args[0]}
        return _blas_info
    _blas_info = {"kind": "openblas", "path": path, "threads": t}
    return _blas_info


def matmul(a, b) -> np.ndarray:
    """NDArray_Matmul for 2-D operands (cblas_sgemm row-major)."""
    use_openblas()
    a, b = _f(a), _f(b)
    if load().oracle_matmul_check(_shape(a), a.ndim, _shape(b), b.ndim) != 0:
        raise _err()
    if a.ndim == 0:
        return binary("multiply", a, b)
    if a.ndim == 1:
        # NDArray_Dot -> NDArray_Inner: sum of products (cblas_sdot), not on the GPU hot path
        return np.float32(np.dot(a.astype(np.float64), b.astype(np.float64)))
    m, k = a.shape
    n = b.shape[1]
    c = np.zeros((m, n), dtype=np.float32)
    load().oracle_matmul(m, n, k, _ptr(a), _ptr(b), _ptr(c))
    return c


def outer(a, b) -> np.ndarray:
    """NDArray_Outer (linalg.c:724-751): cblas_sger(alpha = 1) into a zeroed matrix, i.e.
    out[i][j] = 0 + a[i]*b[j] — one exact product, one rounding, whatever BLAS computes it."""
    a, b = _f(a), _f(b)
    if a.ndim != 1 or b.ndim != 1:
        raise OracleError("Invalid operation: NDArray::outer() requires both arrays to be 1-dimensional vectors.")
    return (a[:, None] * b[None, :] + np.float32(0.0)).astype(np.float32)


def matvec(a, x) -> np.ndarray:
    use_openblas()
    a, x = _f(a), _f(x)
    y = np.zeros(a.shape[0], dtype=np.float32)
    load().oracle_matvec(a.shape[0], a.shape[1], _ptr(a), _ptr(x), _ptr(y))
    return y
