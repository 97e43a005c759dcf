"""Python stand-in for the PHP `NDArray` class surface on the hot path.

PHP is not available where this was built, so the Zend glue (numpower.c) cannot be compiled;
this module plays its role for tests and benchmarks: it marshals Python values the way
ZVAL_TO_NDARRAY does (numpower.c:89-117), calls the SAME host entry points the PHP_METHODs call
(libnumpower_host.so: NDArray_Add_Float, reduce, NDArray_Matmul, NDArray_ToGPU ...), frees
temporaries like CHECK_INPUT_AND_FREE (numpower.c:119-135) and converts results like
RETURN_NDARRAY (numpower.c:137-150: ndim > 0 -> NDArray object, ndim == 0 -> float).

    a = NDArray.array([[1, 2], [3, 4]]).gpu()
    c = (a + 2) * a[0]
    NDArray.sum(c, axis=0).cpu().toArray()

All arithmetic runs in the HIP library.  Operating on CPU-resident arrays raises `Error` (the
reference's CPU path is not re-implemented here and nothing falls back to numpy).
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

from . import _lib
from ._lib import BINARY_OPS, UNARY_OPS

CPU, GPU = 0, 1

# NP_LAZY_BINDING=1 (or NDArray.set_lazy_binding(True)): this stand-in then behaves like a `--with-hip` tree with INTEGRATION.md 2c
# applied — the arithmetic operators / static arithmetic methods / unary methods call the appenders of ext/hip_lazy.c
# (NPH_LazyBinary, NPH_LazyElementWise*), the full reductions NPH_ReduceAll, and EVERY other use of an array's pointer goes through
# the marshalling point (`_p` = buffer_get: NPH_OnBufferGet computes pending values first).  Off by default: the mirror is then the
# 2a + 2b tree (one launch per op) and `$a->lazy()` (numpower_amd/lazy.py) is the explicit form.  Running the whole GPU suite with
# the switch on is the flush-set test of the mirror: every method that is not an appender must see finished values.
import os as _os
_LAZY_BINDING = _os.environ.get("NP_LAZY_BINDING") == "1"
_LAZY_ARITH = ("add", "subtract", "multiply", "divide", "mod", "pow")


class Error(RuntimeError):
    """PHP `Error` thrown by zend_throw_error in the reference."""


class _CNDArrayDescriptor(C.Structure):
    _fields_ = [("type", C.c_char_p), ("elsize", C.c_int), ("numElements", C.c_long)]


class _CNDArray(C.Structure):
    pass


_CNDArray._fields_ = [
    ("uuid", C.c_int), ("strides", C.POINTER(C.c_int)), ("dimensions", C.POINTER(C.c_int)),
    ("ndim", C.c_int), ("data", C.c_void_p), ("base", C.POINTER(_CNDArray)), ("flags", C.c_int),
    ("descriptor", C.POINTER(_CNDArrayDescriptor)), ("iterator", C.c_void_p),
    ("php_iterator", C.c_void_p), ("refcount", C.c_int), ("device", C.c_int)]

_P = C.POINTER(_CNDArray)
_host = None


class _CDims(C.Structure):   # NDArray_Dims, src/ndarray.h:40-43
    _fields_ = [("ptr", C.POINTER(C.c_int)), ("len", C.c_int)]


def host_lib_path() -> Path:
    return _lib.LIBDIR / "libnumpower_host.so"


def _load_host():
    global _host
    if _host is not None:
        return _host
    _lib.load()   # libnp_hip.so first (RTLD_GLOBAL)
    path = host_lib_path()
    if not path.exists():
        raise Error(f"{path} is missing: run `python -m numpower_amd.build` (no CPU fallback)")
    h = C.CDLL(str(path))
    ip = C.POINTER(C.c_int)
    fp = C.POINTER(C.c_float)
    sig = {
        "numpower_host_last_error": (C.c_char_p, []),
        "numpower_host_clear_error": (None, []),
        "NDArray_Zeros": (_P, [ip, C.c_int, C.c_char_p, C.c_int]),
        "NDArray_Empty": (_P, [ip, C.c_int, C.c_char_p, C.c_int]),
        "NDArray_Copy": (_P, [_P, C.c_int]),
        "NDArray_Fill": (_P, [_P, C.c_float]),
        "NDArray_CreateFromDoubleScalar": (_P, [C.c_double]),
        "NDArray_CreateFromLongScalar": (_P, [C.c_long]),
        "NDArray_FromHostBuffer": (_P, [fp, ip, C.c_int]),
        "NDArray_LeadingSlice": (_P, [_P, C.c_int]),
        "NDArray_FREE": (None, [_P]),
        "NDArray_ToGPU": (_P, [_P]),
        "NDArray_ToCPU": (_P, [_P]),
        "NDArray_GetFloatScalar": (C.c_float, [_P]),
        "NDArray_CopyToHostBuffer": (C.c_int, [_P, fp]),
        "NDArray_LiveDeviceAllocations": (C.c_long, []),
        "NDArray_IsBroadcastable": (C.c_int, [_P, _P]),
        # reference signatures (cuda_math.h:10-15,75-76): the op is a cuda_float_* FUNCTION POINTER
        "NDArrayMathGPU_ElementWise": (_P, [_P, C.c_void_p]),
        "NDArrayMathGPU_ElementWise1F": (_P, [_P, C.c_void_p, C.c_float]),
        "NDArrayMathGPU_ElementWise2F": (_P, [_P, C.c_void_p, C.c_float, C.c_float]),
        "NDArrayMathGPU_ElementWise1N": (_P, [_P, C.c_void_p, _P]),
        "NDArray_Abs": (_P, [_P]),
        "NDArray_Rsqrt": (_P, [_P]),
        "NDArray_Exp2": (_P, [_P]),
        "NDArray_Sum_Float": (C.c_float, [_P]),
        "NDArray_Float_Prod": (C.c_float, [_P]),
        "NDArray_Mean_Float": (C.c_float, [_P]),
        "NDArray_Median_Float": (C.c_float, [_P]),
        "NDArray_Quantile": (_P, [_P, _P]),
        "NDArray_Min": (C.c_float, [_P]),
        "NDArray_Max": (C.c_float, [_P]),
        "NDArray_MinAxis": (_P, [_P, C.c_int]),
        "NDArray_MaxAxis": (_P, [_P, C.c_int]),
        "NDArray_Matmul": (_P, [_P, _P]),
        "NDArray_Dot": (_P, [_P, _P]),
        "NDArray_Outer": (_P, [_P, _P]),
        "NDArray_Inner": (_P, [_P, _P]),
        "NDArray_BatchedMatmul": (_P, [_P, _P]),
        "NDArray_ShardedBatchedMatmul": (_P, [_P, _P, C.c_int, C.c_int]),
        "NDArray_CommInit": (C.c_int, [C.c_int, C.c_int, C.c_char_p]),
        "NDArray_CommDestroy": (C.c_int, []),
        "NDArray_CommRank": (C.c_int, []),
        "NDArray_CommWorld": (C.c_int, []),
    }
    for name in ("Add", "Subtract", "Multiply", "Divide", "Mod", "Pow"):
        sig[f"NDArray_{name}_Float"] = (_P, [_P, _P])
    for name in ("Equal", "NotEqual", "Greater", "GreaterEqual", "Less", "LessEqual", "Maximum", "Minimum"):
        sig[f"NDArray_{name}"] = (_P, [_P, _P])
    sig["NDArray_All"] = (C.c_float, [_P])
    sig["NDArray_Transpose"] = (_P, [_P, C.POINTER(_CDims)])
    sig["NDArray_ArgMinMaxCommon"] = (_P, [_P, C.c_int, C.c_bool, C.c_bool])
    sig["NDArray_Variance"] = (_P, [_P])
    sig["NDArray_Std"] = (_P, [_P])
    sig["NDArray_Average"] = (_P, [_P, _P])
    sig["NDArray_FullOn"] = (_P, [ip, C.c_int, C.c_double, C.c_int])
    sig["NDArray_IdentityOn"] = (_P, [C.c_int, C.c_int])
    sig["NDArray_ArangeOn"] = (_P, [C.c_double, C.c_double, C.c_double, C.c_int])
    sig["NDArray_ArrayEqual"] = (C.c_int, [_P, _P])
    sig["NDArray_AllClose"] = (C.c_int, [_P, _P, C.c_float, C.c_float])
    sig["NDArray_ToContiguous"] = (_P, [_P])
    sig["NDArray_Diagonal"] = (_P, [_P, C.c_int])
    sig["NDArray_Trace"] = (_P, [_P])
    sig["NDArray_Reshape"] = (_P, [_P, ip, C.c_int])
    sig["NDArray_Flatten"] = (_P, [_P])
    sig["NDArray_ExpandDim"] = (_P, [_P, _P])
    sig["NDArray_Append"] = (_P, [C.POINTER(_P), C.c_int, C.c_int])
    for name in ("NDArray_AtLeast1D", "NDArray_AtLeast2D", "NDArray_AtLeast3D", "NDArray_Diag"):
        sig[name] = (_P, [_P])
    sig["NDArray_Squeeze"] = (_P, [_P, _P])
    sig["NDArray_SwapAxes"] = (_P, [_P, C.c_int, C.c_int])
    sig["NDArray_Rollaxis"] = (_P, [_P, C.c_int, C.c_int])
    sig["NDArray_Moveaxis"] = (_P, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int])
    sig["NDArray_Concatenate"] = (_P, [C.POINTER(_P), C.c_int, C.c_int])
    for name in ("NDArray_VSTACK", "NDArray_HSTACK", "NDArray_DSTACK", "NDArray_ColumnStack"):
        sig[name] = (_P, [C.POINTER(_P), C.c_int])
    sig["NDArray_Slice"] = (_P, [_P, C.POINTER(_P), C.c_int])
    # ext/hip_lazy.c
    sig["NPH_LazyBinary"] = (_P, [C.c_int, C.c_void_p, _P, _P])
    sig["NPH_LazyElementWise"] = (_P, [_P, C.c_void_p])
    sig["NPH_LazyElementWise1F"] = (_P, [_P, C.c_void_p, C.c_float])
    sig["NPH_LazyElementWise2F"] = (_P, [_P, C.c_void_p, C.c_float, C.c_float])
    sig["NPH_ReduceAll"] = (C.c_float, [C.c_int, C.c_void_p, _P])
    sig["NPH_OnBufferGet"] = (None, [_P])
    sig["NPH_Flush"] = (C.c_int, [_P])
    sig["NPH_IsPending"] = (C.c_int, [_P])
    sig["NPH_PendingCount"] = (C.c_int, [])
    for name, (res, args) in sig.items():
        fn = getattr(h, name)
        fn.restype = res
        fn.argtypes = args
    # reduce(array, int *axis, operation): the operation is passed as the address of the
    # library's own NDArray_Add_Float / NDArray_Multiply_Float, exactly as numpower.c does.
    h.reduce.restype = _P
    h.reduce.argtypes = [_P, C.POINTER(C.c_int), C.c_void_p]
    _host = h
    return h


def _fn(h, symbol):
    """Address of one of the library's own functions, passed where numpower.c passes the function name
    (`NDArrayMathGPU_ElementWise(nda, cuda_float_sin)`, numpower.c:1651)."""
    return C.cast(getattr(h, symbol), C.c_void_p)


def _raise_pending(h, what="call"):
    msg = h.numpower_host_last_error()
    text = msg.decode() if msg else ""
    h.numpower_host_clear_error()
    raise Error(text or f"{what} failed")


_BINARY_FN = {"add": "NDArray_Add_Float", "subtract": "NDArray_Subtract_Float",
              "multiply": "NDArray_Multiply_Float", "divide": "NDArray_Divide_Float",
              "mod": "NDArray_Mod_Float", "pow": "NDArray_Pow_Float",
              # comparison family (src/logic.c), PHP methods equal / not_equal / greater / ...
              "equal": "NDArray_Equal", "not_equal": "NDArray_NotEqual", "greater": "NDArray_Greater",
              "greater_equal": "NDArray_GreaterEqual", "less": "NDArray_Less",
              "less_equal": "NDArray_LessEqual",
              "maximum": "NDArray_Maximum", "minimum": "NDArray_Minimum"}

# PHP method name -> np_unary_op for the plain NDArrayMathGPU_ElementWise family
# (method table numpower.c:5136-5174)
_UNARY_METHODS = [n for n in UNARY_OPS if n not in ("clip", "round")]


class NDArray:
    """Handle on a C `NDArray*` owned by libnumpower_host.so (PHP objects hold a uuid into
    MAIN_MEM_STACK instead, buffer.c:91-120; the ownership rules are the same: the object's
    destructor calls NDArray_FREE, views keep their base alive through the refcount)."""

    __slots__ = ("_ptr", "_h")

    def __init__(self, ptr):
        self._h = _load_host()
        if not ptr:
            _raise_pending(self._h)
        self._ptr = ptr

    @property
    def _p(self):
        """The marshalling point (ZVAL_TO_NDARRAY -> buffer_get, numpower.c:105, src/buffer.c:80): with the lazy binding on, a pending
        value is computed here, and so are the chains that read this array (the consumer may write it).  Appenders use _ptr."""
        p = self._ptr
        if _LAZY_BINDING and p:
            self._h.NPH_OnBufferGet(p)
            if self._h.numpower_host_last_error():
                _raise_pending(self._h)
        return p

    @_p.setter
    def _p(self, value):
        self._ptr = value

    @staticmethod
    def set_lazy_binding(on: bool):
        global _LAZY_BINDING
        _LAZY_BINDING = bool(on)

    def __del__(self):
        try:
            if self._ptr:
                self._h.NDArray_FREE(self._ptr)      # ndarray_destructor -> buffer_ndarray_free: no flush on the way out
                self._ptr = None
        except Exception:
            pass

    # ---- marshalling (ZVAL_TO_NDARRAY, numpower.c:89-117) ---------------------------------
    @staticmethod
    def _coerce(value):
        """-> (NDArray, is_temporary)"""
        if isinstance(value, NDArray):
            return value, False
        h = _load_host()
        if isinstance(value, bool):
            raise Error("argument must be an array, long, double, gdimage or ndarray.")
        if isinstance(value, int):
            return NDArray(h.NDArray_CreateFromLongScalar(value)), True
        if isinstance(value, float):
            return NDArray(h.NDArray_CreateFromDoubleScalar(value)), True
        if isinstance(value, (list, tuple, np.ndarray)):
            return NDArray.array(value), True
        raise Error("argument must be an array, long, double, gdimage or ndarray.")

    @staticmethod
    def _wrap(ptr):
        """RETURN_NDARRAY (numpower.c:137-150)."""
        h = _load_host()
        if not ptr:
            _raise_pending(h)
        if ptr.contents.ndim > 0:
            return NDArray(ptr)
        v = h.NDArray_GetFloatScalar(ptr)
        h.NDArray_FREE(ptr)
        return float(v)

    # ---- construction -----------------------------------------------------------------------
    @staticmethod
    def array(values) -> "NDArray":
        """NDArray::array — CPU array from a nested list (fp32)."""
        a = np.asarray(values, dtype=np.float32, order="C")
        h = _load_host()
        shape = (C.c_int * max(a.ndim, 1))(*a.shape)
        return NDArray(h.NDArray_FromHostBuffer(a.ctypes.data_as(C.POINTER(C.c_float)), shape, a.ndim))

    @staticmethod
    def zeros(shape, device=CPU) -> "NDArray":
        h = _load_host()
        s = (C.c_int * max(len(shape), 1))(*shape)
        return NDArray(h.NDArray_Zeros(s, len(shape), b"float32", device))

    # ---- placement ---------------------------------------------------------------------------
    @staticmethod
    def full(shape, fill_value, device=CPU) -> "NDArray":      # PHP_METHOD full, numpower.c:1214
        h = _load_host()
        arr = (C.c_int * max(len(shape), 1))(*[int(v) for v in shape])
        return NDArray(h.NDArray_FullOn(arr, len(shape), float(fill_value), device))

    @staticmethod
    def ones(shape, device=CPU) -> "NDArray":                  # numpower.c:1261
        return NDArray.full(shape, 1.0, device)

    @staticmethod
    def identity(size: int, device=CPU) -> "NDArray":          # numpower.c:940
        return NDArray(_load_host().NDArray_IdentityOn(int(size), device))

    @staticmethod
    def arange(stop, start=0.0, step=1.0, device=GPU) -> "NDArray":   # numpower.c:1284-1302: (stop, start, step)
        return NDArray(_load_host().NDArray_ArangeOn(float(start), float(stop), float(step), device))

    def gpu(self) -> "NDArray":
        """$a->gpu(): always a new array (NDArray_ToGPU, ndarray.c:1037-1068)."""
        return NDArray(self._h.NDArray_ToGPU(self._p))

    def cpu(self) -> "NDArray":
        return NDArray(self._h.NDArray_ToCPU(self._p))

    @staticmethod
    def setDevice(device_id: int) -> None:
        """NDArray::setDevice -> cudaSetDevice in the reference (numpower.c:615-635)."""
        _lib.check(_lib.load().np_set_device(int(device_id)))

    def isGPU(self) -> bool:
        return self._p.contents.device == GPU

    # ---- introspection -------------------------------------------------------------------------
    def shape(self):
        c = self._p.contents
        return [c.dimensions[i] for i in range(c.ndim)]

    def size(self) -> int:
        return int(self._p.contents.descriptor.contents.numElements)

    def ndim(self) -> int:
        return int(self._p.contents.ndim)

    def toArray(self):
        """toArray(): nested list of floats; throws for GPU arrays (numpower.c:456-477)."""
        return self.numpy().astype(np.float64).tolist()

    def numpy(self) -> np.ndarray:
        out = np.empty(self.shape(), dtype=np.float32)
        if self._h.NDArray_CopyToHostBuffer(self._p, out.ctypes.data_as(C.POINTER(C.c_float))) != 0:
            _raise_pending(self._h)
        return out

    def fill(self, value: float) -> "NDArray":
        if not self._h.NDArray_Fill(self._p, value):
            _raise_pending(self._h)
        return self

    def __getitem__(self, index: int) -> "NDArray":
        """$a[i]: view of slice i of the leading axis."""
        return NDArray._wrap(self._h.NDArray_LeadingSlice(self._p, int(index)))

    def __len__(self):
        return self.shape()[0]

    # ---- binary ops ------------------------------------------------------------------------------
    @staticmethod
    def _binary(name, a, b):
        h = _load_host()
        x, _tx = NDArray._coerce(a)
        y, _ty = NDArray._coerce(b)
        if _LAZY_BINDING and name in _LAZY_ARITH:     # ndarray_do_operation_ex / PHP_METHOD(add ...) as section 2c edits them
            return NDArray._wrap(h.NPH_LazyBinary(BINARY_OPS[name], _fn(h, _BINARY_FN[name]), x._ptr, y._ptr))
        return NDArray._wrap(getattr(h, _BINARY_FN[name])(x._p, y._p))

    add = staticmethod(lambda a, b: NDArray._binary("add", a, b))
    subtract = staticmethod(lambda a, b: NDArray._binary("subtract", a, b))
    multiply = staticmethod(lambda a, b: NDArray._binary("multiply", a, b))
    divide = staticmethod(lambda a, b: NDArray._binary("divide", a, b))
    mod = staticmethod(lambda a, b: NDArray._binary("mod", a, b))
    pow = staticmethod(lambda a, b: NDArray._binary("pow", a, b))
    equal = staticmethod(lambda a, b: NDArray._binary("equal", a, b))
    not_equal = staticmethod(lambda a, b: NDArray._binary("not_equal", a, b))
    greater = staticmethod(lambda a, b: NDArray._binary("greater", a, b))
    greater_equal = staticmethod(lambda a, b: NDArray._binary("greater_equal", a, b))
    less = staticmethod(lambda a, b: NDArray._binary("less", a, b))
    less_equal = staticmethod(lambda a, b: NDArray._binary("less_equal", a, b))
    maximum = staticmethod(lambda a, b: NDArray._binary("maximum", a, b))
    minimum = staticmethod(lambda a, b: NDArray._binary("minimum", a, b))

    @staticmethod
    def all(a) -> int:
        """PHP_METHOD(NDArray, all) without axis: RETURN_LONG(NDArray_All(nda)) (numpower.c:1330)."""
        h = _load_host()
        x, _ = NDArray._coerce(a)
        h.numpower_host_clear_error()
        v = h.NDArray_All(x._p)
        if h.numpower_host_last_error():
            _raise_pending(h)
        return int(v)

    def __add__(self, o): return NDArray._binary("add", self, o)
    def __radd__(self, o): return NDArray._binary("add", o, self)
    def __sub__(self, o): return NDArray._binary("subtract", self, o)
    def __rsub__(self, o): return NDArray._binary("subtract", o, self)
    def __mul__(self, o): return NDArray._binary("multiply", self, o)
    def __rmul__(self, o): return NDArray._binary("multiply", o, self)
    def __truediv__(self, o): return NDArray._binary("divide", self, o)
    def __rtruediv__(self, o): return NDArray._binary("divide", o, self)
    def __mod__(self, o): return NDArray._binary("mod", self, o)
    def __pow__(self, o): return NDArray._binary("pow", self, o)

    @staticmethod
    def square(a):   # PHP_METHOD(NDArray, square): Multiply_Float(nda, nda), numpower.c:3093
        x, _ = NDArray._coerce(a)
        return NDArray._binary("multiply", x, x)

    @staticmethod
    def arctan2(x, y):
        h = _load_host()
        a, _ = NDArray._coerce(x)
        b, _ = NDArray._coerce(y)
        return NDArray._wrap(h.NDArrayMathGPU_ElementWise1N(a._p, _fn(h, "cuda_float_arctan2"), b._p))

    # ---- unary ops ---------------------------------------------------------------------------------
    @staticmethod
    def _unary(name, a):
        h = _load_host()
        x, _ = NDArray._coerce(a)
        if name == "rsqrt":
            # numpower.c:1791 passes cuda_float_arccos for rsqrt on the GPU (a copy-paste slip: the reference has
            # no cuda_float_rsqrt); the stand-in asks for the CPU definition (float_rsqrt, double_math.c:111-126)
            # through NDArray_Rsqrt instead of reproducing the slip.
            return NDArray._wrap(h.NDArray_Rsqrt(x._p))
        if name == "exp2":    # no cuda_float_exp2 in the reference (numpower.c:3153 is CPU-only): own entry point
            return NDArray._wrap(h.NDArray_Exp2(x._p))
        if name == "abs":     # PHP_METHOD(NDArray, abs) calls NDArray_Abs (numpower.c:1619)
            return NDArray._wrap(h.NDArray_Abs(x._p))
        if _LAZY_BINDING and x._ptr.contents.device == GPU:      # the method's device branch, as section 2c edits it
            return NDArray._wrap(h.NPH_LazyElementWise(x._ptr, _fn(h, "cuda_float_" + name)))
        return NDArray._wrap(h.NDArrayMathGPU_ElementWise(x._p, _fn(h, "cuda_float_" + name)))

    @staticmethod
    def clip(a, min, max):
        h = _load_host()
        x, _ = NDArray._coerce(a)
        if _LAZY_BINDING and x._ptr.contents.device == GPU:
            return NDArray._wrap(h.NPH_LazyElementWise2F(x._ptr, _fn(h, "cuda_float_clip"), float(min), float(max)))
        return NDArray._wrap(h.NDArrayMathGPU_ElementWise2F(x._p, _fn(h, "cuda_float_clip"), float(min), float(max)))

    @staticmethod
    def round(a, precision=0):
        h = _load_host()
        x, _ = NDArray._coerce(a)
        if _LAZY_BINDING and x._ptr.contents.device == GPU:
            return NDArray._wrap(h.NPH_LazyElementWise1F(x._ptr, _fn(h, "cuda_float_round"), float(precision)))
        return NDArray._wrap(h.NDArrayMathGPU_ElementWise1F(x._p, _fn(h, "cuda_float_round"), float(precision)))

    # ---- reductions -----------------------------------------------------------------------------------
    @staticmethod
    def _reduce(op, a, axis):
        h = _load_host()
        x, _ = NDArray._coerce(a)
        if axis is None:
            fn = {"sum": h.NDArray_Sum_Float, "prod": h.NDArray_Float_Prod, "min": h.NDArray_Min,
                  "max": h.NDArray_Max}[op]
            h.numpower_host_clear_error()
            if _LAZY_BINDING:     # PHP_METHOD(sum ...) as section 2c edits it: a pending operand is reduced inside its chain's kernel
                from ._lib import REDUCE_OPS
                v = h.NPH_ReduceAll(REDUCE_OPS[op], C.cast(fn, C.c_void_p), x._ptr)
            else:
                v = fn(x._p)
            if h.numpower_host_last_error():
                _raise_pending(h)
            return float(v)
        ax = C.c_int(int(axis))
        if op == "sum":    # numpower.c:4637
            return NDArray._wrap(h.reduce(x._p, C.byref(ax), C.cast(h.NDArray_Add_Float, C.c_void_p)))
        if op == "prod":   # numpower.c:4742
            return NDArray._wrap(h.reduce(x._p, C.byref(ax), C.cast(h.NDArray_Multiply_Float, C.c_void_p)))
        if op == "min":
            return NDArray._wrap(h.NDArray_MinAxis(x._p, int(axis)))
        return NDArray._wrap(h.NDArray_MaxAxis(x._p, int(axis)))

    sum = staticmethod(lambda a, axis=None: NDArray._reduce("sum", a, axis))
    prod = staticmethod(lambda a, axis=None: NDArray._reduce("prod", a, axis))
    min = staticmethod(lambda a, axis=None: NDArray._reduce("min", a, axis))
    max = staticmethod(lambda a, axis=None: NDArray._reduce("max", a, axis))

    @staticmethod
    def mean(a, axis=None):
        """PHP_METHOD(NDArray, mean) (numpower.c:2642-2688)."""
        h = _load_host()
        x, _ = NDArray._coerce(a)
        if axis is None:
            # NDArray_Sum_Float(nda) / NDArray_NUMELEMENTS(nda): float / long in C
            s = NDArray._reduce("sum", x, None)
            return float(np.float32(s) / np.float32(x.size()))
        total = NDArray._reduce("sum", x, axis)
        count = NDArray(h.NDArray_CreateFromLongScalar(x.shape()[int(axis)]))
        if isinstance(total, float):   # 1-D input: 0-d sum
            return float(np.float32(total) / np.float32(x.shape()[int(axis)]))
        return NDArray._binary("divide", total, count)

    @staticmethod
    def median(a):
        """PHP_METHOD(NDArray, median) (numpower.c:2700-2731) -> float; on the device (radix select)."""
        h = _load_host()
        x, _ = NDArray._coerce(a)
        h.numpower_host_clear_error()
        v = h.NDArray_Median_Float(x._p)
        if h.numpower_host_last_error():
            _raise_pending(h)
        return float(v)

    @staticmethod
    def quantile(a, q):
        """PHP_METHOD(NDArray, quantile) (numpower.c:2788): q a scalar in [0, 1] -> float."""
        h = _load_host()
        x, _ = NDArray._coerce(a)
        qq, _ = NDArray._coerce(q)
        return NDArray._wrap(h.NDArray_Quantile(x._p, qq._p))

    # ---- layout (PHP_METHOD transpose, numpower.c:1404-1450) ----
    @staticmethod
    def transpose(a, axes=None):
        h = _load_host()
        x, _ = NDArray._coerce(a)
        if axes is None:
            return NDArray._wrap(h.NDArray_Transpose(x._p, None))
        arr = (C.c_int * max(len(axes), 1))(*[int(v) for v in axes])
        dims = _CDims(arr, len(axes))
        return NDArray._wrap(h.NDArray_Transpose(x._p, C.byref(dims)))

    # ---- views / layout / equality around the path ----
    @staticmethod
    def reshape(a, shape):   # PHP_METHOD reshape, numpower.c:644: a view sharing the buffer
        h = _load_host()
        x, _ = NDArray._coerce(a)
        arr = (C.c_int * max(len(shape), 1))(*[int(v) for v in shape])
        return NDArray._wrap(h.NDArray_Reshape(x._p, arr, len(shape)))

    @staticmethod
    def flatten(a):          # numpower.c:1584
        h = _load_host()
        x, _ = NDArray._coerce(a)
        return NDArray._wrap(h.NDArray_Flatten(x._p))

    @staticmethod
    def expand_dims(a, axis):   # numpower.c:3563: axis int or list of ints
        h = _load_host()
        x, _ = NDArray._coerce(a)
        ax, _ = NDArray._coerce(axis)
        return NDArray._wrap(h.NDArray_ExpandDim(x._p, ax._p))

    @staticmethod
    def append(a, b):        # numpower.c:3908: flat concatenation (axis = -1)
        h = _load_host()
        x, _ = NDArray._coerce(a)
        y, _ = NDArray._coerce(b)
        arr = (_P * 2)(x._p, y._p)
        return NDArray._wrap(h.NDArray_Append(arr, -1, 2))

    # ---- manipulation wrappers (PHP_METHODs atleast_1d … column_stack, numpower.c:1480-1570, 3590-3900) ----
    @staticmethod
    def _unary_host(name, a):
        h = _load_host()
        x, _ = NDArray._coerce(a)
        return NDArray._wrap(getattr(h, name)(x._p))

    atleast_1d = staticmethod(lambda a: NDArray._unary_host("NDArray_AtLeast1D", a))
    atleast_2d = staticmethod(lambda a: NDArray._unary_host("NDArray_AtLeast2D", a))
    atleast_3d = staticmethod(lambda a: NDArray._unary_host("NDArray_AtLeast3D", a))
    diag = staticmethod(lambda a: NDArray._unary_host("NDArray_Diag", a))

    @staticmethod
    def squeeze(a, axis=None):
        h = _load_host()
        x, _ = NDArray._coerce(a)
        if axis is None:
            return NDArray._wrap(h.NDArray_Squeeze(x._p, None))
        ax, _ = NDArray._coerce(axis)
        return NDArray._wrap(h.NDArray_Squeeze(x._p, ax._p))

    @staticmethod
    def swapaxes(a, axis1, axis2):
        h = _load_host()
        x, _ = NDArray._coerce(a)
        return NDArray._wrap(h.NDArray_SwapAxes(x._p, int(axis1), int(axis2)))

    @staticmethod
    def rollaxis(a, axis, start=0):
        h = _load_host()
        x, _ = NDArray._coerce(a)
        return NDArray._wrap(h.NDArray_Rollaxis(x._p, int(axis), int(start)))

    @staticmethod
    def moveaxis(a, source, destination):
        h = _load_host()
        x, _ = NDArray._coerce(a)
        src = [int(v) for v in (source if isinstance(source, (list, tuple)) else [source])]
        dst = [int(v) for v in (destination if isinstance(destination, (list, tuple)) else [destination])]
        s = (C.c_int * max(len(src), 1))(*src)
        d = (C.c_int * max(len(dst), 1))(*dst)
        return NDArray._wrap(h.NDArray_Moveaxis(x._p, s, d, len(src), len(dst)))

    @staticmethod
    def _stack(name, arrays, *extra):
        h = _load_host()
        xs = [NDArray._coerce(a)[0] for a in arrays]
        arr = (_P * max(len(xs), 1))(*[x._p for x in xs])
        return NDArray._wrap(getattr(h, name)(arr, len(xs), *extra))

    concatenate = staticmethod(lambda arrays, axis=0: NDArray._stack("NDArray_Concatenate", arrays, int(axis)))
    vstack = staticmethod(lambda arrays: NDArray._stack("NDArray_VSTACK", arrays))
    hstack = staticmethod(lambda arrays: NDArray._stack("NDArray_HSTACK", arrays))
    dstack = staticmethod(lambda arrays: NDArray._stack("NDArray_DSTACK", arrays))
    column_stack = staticmethod(lambda arrays: NDArray._stack("NDArray_ColumnStack", arrays))

    @staticmethod
    def diagonal(a):         # numpower.c:1179
        h = _load_host()
        x, _ = NDArray._coerce(a)
        return NDArray._wrap(h.NDArray_Diagonal(x._p, 0))

    @staticmethod
    def trace(a):            # numpower.c:4067
        h = _load_host()
        x, _ = NDArray._coerce(a)
        return NDArray._wrap(h.NDArray_Trace(x._p))

    @staticmethod
    def array_equal(a, b) -> bool:   # the `==` handler of the PHP object, numpower.c:175-186
        h = _load_host()
        x, _ = NDArray._coerce(a)
        y, _ = NDArray._coerce(b)
        h.numpower_host_clear_error()
        r = h.NDArray_ArrayEqual(x._p, y._p)
        if h.numpower_host_last_error():
            _raise_pending(h)
        return bool(r)

    @staticmethod
    def allclose(a, b, rtol=1e-05, atol=1e-08) -> bool:   # numpower.c:1358-1391
        h = _load_host()
        x, _ = NDArray._coerce(a)
        y, _ = NDArray._coerce(b)
        if x is y:
            return True
        r = h.NDArray_AllClose(x._p, y._p, float(rtol), float(atol))
        if r == -1:
            _raise_pending(h)
        return bool(r)

    def slice(self, *indices):   # PHP_METHOD slice, numpower.c:4773: each index [start(,stop(,step))]
        h = _load_host()
        idx = [NDArray._coerce(list(i) if isinstance(i, (list, tuple)) else [i])[0] for i in indices]
        arr = (_P * max(len(idx), 1))(*[i._p for i in idx])
        return NDArray._wrap(h.NDArray_Slice(self._p, arr, len(idx)))

    def contiguous(self):        # NDArray_ToContiguous
        return NDArray._wrap(_load_host().NDArray_ToContiguous(self._p))

    # ---- argmax / argmin (PHP_METHOD argmax / argmin, numpower.c:2570-2630) ----
    @staticmethod
    def _arg(a, axis, keepdims, is_max):
        h = _load_host()
        x, _ = NDArray._coerce(a)
        ax = 128 if axis is None else int(axis)   # ZEND_NUM_ARGS() == 1 -> axis = 128 (flattened)
        return NDArray._wrap(h.NDArray_ArgMinMaxCommon(x._p, ax, bool(keepdims), bool(is_max)))

    argmax = staticmethod(lambda a, axis=None, keepdims=False: NDArray._arg(a, axis, keepdims, True))
    argmin = staticmethod(lambda a, axis=None, keepdims=False: NDArray._arg(a, axis, keepdims, False))

    # ---- statistics (PHP_METHOD variance / std / average, numpower.c:2743-2900) ----
    @staticmethod
    def variance(a):
        h = _load_host()
        x, _ = NDArray._coerce(a)
        return NDArray._wrap(h.NDArray_Variance(x._p))

    @staticmethod
    def std(a):
        h = _load_host()
        x, _ = NDArray._coerce(a)
        return NDArray._wrap(h.NDArray_Std(x._p))

    @staticmethod
    def average(a, weights=None):
        h = _load_host()
        x, _ = NDArray._coerce(a)
        if weights is None:
            return NDArray._wrap(h.NDArray_Average(x._p, None))
        w, _ = NDArray._coerce(weights)
        return NDArray._wrap(h.NDArray_Average(x._p, w._p))

    # ---- linear algebra -----------------------------------------------------------------------------------
    @staticmethod
    def matmul(a, b):
        h = _load_host()
        x, _ = NDArray._coerce(a)
        y, _ = NDArray._coerce(b)
        return NDArray._wrap(h.NDArray_Matmul(x._p, y._p))

    @staticmethod
    def dot(a, b):
        h = _load_host()
        x, _ = NDArray._coerce(a)
        y, _ = NDArray._coerce(b)
        return NDArray._wrap(h.NDArray_Dot(x._p, y._p))

    @staticmethod
    def inner(a, b):         # PHP_METHOD inner
        h = _load_host()
        x, _ = NDArray._coerce(a)
        y, _ = NDArray._coerce(b)
        return NDArray._wrap(h.NDArray_Inner(x._p, y._p))

    def copy(self):          # PHP_METHOD copy: NDArray_Copy on the array's own device
        return NDArray._wrap(_load_host().NDArray_Copy(self._p, C.cast(self._p, _P).contents.device))

    @staticmethod
    def outer(a, b):
        h = _load_host()
        x, _ = NDArray._coerce(a)
        y, _ = NDArray._coerce(b)
        return NDArray._wrap(h.NDArray_Outer(x._p, y._p))

    @staticmethod
    def batched_matmul(a, b):
        h = _load_host()
        return NDArray._wrap(h.NDArray_BatchedMatmul(a._p, b._p))

    @staticmethod
    def comm_init(rank: int, world: int, endpoint: str):
        """Join the node's ranks (one process per GPU); endpoint "tcp://127.0.0.1:port" or a file path."""
        if _load_host().NDArray_CommInit(int(rank), int(world), endpoint.encode()) != 0:
            _raise_pending(_load_host())

    @staticmethod
    def comm_destroy():
        if _load_host().NDArray_CommDestroy() != 0:
            _raise_pending(_load_host())

    @staticmethod
    def sharded_batched_matmul(a_slab, b_slab, batch: int, gather_mode: int = 1):
        """This rank's slab of a batch of `batch` products; gather_mode 0 keeps the result sharded, 1 gathers it
        with one all-gather, k >= 2 overlaps the gather of k pieces with the GEMM (include/numpower_host.h)."""
        h = _load_host()
        return NDArray._wrap(h.NDArray_ShardedBatchedMatmul(a_slab._p, b_slab._p, int(batch), int(gather_mode)))

    @staticmethod
    def live_device_allocations() -> int:
        return int(_load_host().NDArray_LiveDeviceAllocations())


def _install_unary_methods():
    for name in _UNARY_METHODS:
        if hasattr(NDArray, name):
            continue
        setattr(NDArray, name, staticmethod(lambda a, _n=name: NDArray._unary(_n, a)))


_install_unary_methods()
NDArray.negative = NDArray.negate   # PHP method name (numpower.c method table); float_negate underneath
# `abs` goes through NDArray_Abs in the reference (arithmetics.c:934-947); same kernel
nd = NDArray
