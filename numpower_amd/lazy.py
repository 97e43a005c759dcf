"""Caller-side fusion of elementwise chains.

In the reference every PHP-level op — `nd::exp($a)`, `* $b`, `+ 2` — allocates a result and makes
a full round trip through memory (ndarray_do_operation_ex, numpower.c:193-229).  `Lazy` is what a
Zend glue would keep behind an `NDArray` object instead: the chain of pending elementwise ops,
flushed as ONE kernel (NDArray_FusedChain -> np_fused_chain) when a value is needed.

    y = (a.lazy().exp() * b + 2.0).eval()        # one pass over HBM, bit-identical to
    y = (NDArray.exp(a) * b) + 2.0               # three passes and two temporaries

Chains are linear: acc = f_k(... f_1(a)) and keep the shape of `a` throughout (eagerly, `$vector + $a` with
a 1 x C `$a` is a flat op whose result takes the vector's shape, arithmetics.c:194-197; inside a chain it stays
1 x C); binary steps take another GPU array of the same size, a
smaller GPU array that broadcasts onto the chain's shape (row vector, column, 0-d) or a Python
number.  Anything else (reductions, matmul) is evaluated eagerly by NDArray as before.
"""
from __future__ import annotations

import ctypes as C

from . import _lib
from ._lib import BINARY_OPS, NP_FUSED_BINARY, NP_FUSED_UNARY, UNARY_OPS, FusedOp
from .ndarray import Error, NDArray, _load_host, _P

MAX_OPS = 12
MAX_INPUTS = 6


class Lazy:
    """A pending chain.  Chains are PERSISTENT values: every step returns a NEW Lazy (inputs / ops lists
    copied), so one chain can be continued in two directions —

        base = a.lazy().exp();  y1 = base + 1;  y2 = base * 2

    leaves `base`, `y1`, `y2` three different expressions (exp(a), exp(a)+1, exp(a)*2)."""

    def __init__(self, first: NDArray, _inputs=None, _ops=None):
        if _inputs is None:
            if not isinstance(first, NDArray):
                raise Error("Lazy chains start from an NDArray")
            _inputs, _ops = [first], []
        self.inputs = _inputs
        self.ops = _ops

    # ---- building -----------------------------------------------------------------------------
    def _room_for(self, extra_inputs=0):
        """-> (inputs, ops) copies to extend; a chain that is full is evaluated first and the copy
        starts from its value (this object keeps its own pending ops)."""
        if len(self.ops) >= MAX_OPS or len(self.inputs) + extra_inputs > MAX_INPUTS:
            return [self.eval()], []
        return list(self.inputs), list(self.ops)

    def _unary(self, name, p0=0.0, p1=0.0):
        inputs, ops = self._room_for()
        ops.append(FusedOp(NP_FUSED_UNARY, UNARY_OPS[name], 0, 0, float(p0), float(p1), 0, 0))
        return Lazy(None, inputs, ops)

    def _binary(self, name, other, swap):
        if isinstance(other, Lazy):
            other = other.eval()
        inputs, ops = self._room_for(1)
        if isinstance(other, NDArray):
            operand = None
            for i, x in enumerate(inputs):      # reuse an input that is already bound
                if x is other:
                    operand = i
            if operand is None:
                inputs.append(other)
                operand = len(inputs) - 1
        elif isinstance(other, (int, float)) and not isinstance(other, bool):
            scalar, _ = NDArray._coerce(other)        # 0-d CPU scalar, as ZVAL_TO_NDARRAY makes it
            inputs.append(scalar)
            operand = len(inputs) - 1
        else:
            raise Error("argument must be an array, long, double, gdimage or ndarray.")
        ops.append(FusedOp(NP_FUSED_BINARY, BINARY_OPS[name], operand, 1 if swap else 0, 0.0, 0.0, 0, 0))
        return Lazy(None, inputs, ops)

    def __add__(self, o): return self._binary("add", o, False)
    def __radd__(self, o): return self._binary("add", o, True)
    def __sub__(self, o): return self._binary("subtract", o, False)
    def __rsub__(self, o): return self._binary("subtract", o, True)
    def __mul__(self, o): return self._binary("multiply", o, False)
    def __rmul__(self, o): return self._binary("multiply", o, True)
    def __truediv__(self, o): return self._binary("divide", o, False)
    def __rtruediv__(self, o): return self._binary("divide", o, True)
    def __mod__(self, o): return self._binary("mod", o, False)
    def __rmod__(self, o): return self._binary("mod", o, True)
    def __pow__(self, o): return self._binary("pow", o, False)
    def __rpow__(self, o): return self._binary("pow", o, True)

    def clip(self, min, max): return self._unary("clip", min, max)
    def round(self, precision=0): return self._unary("round", precision)

    def __getattr__(self, name):
        if name in UNARY_OPS:
            return lambda: self._unary(name)
        if name in BINARY_OPS:
            return lambda other: self._binary(name, other, False)
        raise AttributeError(name)

    # ---- flushing -------------------------------------------------------------------------------
    def eval(self) -> NDArray:
        """Run the pending chain as one kernel and return the resulting GPU array."""
        h = _load_host()
        h.NDArray_FusedChain.restype = _P
        h.NDArray_FusedChain.argtypes = [C.POINTER(_P), C.c_int, C.POINTER(FusedOp), C.c_int]
        arr = (_P * len(self.inputs))(*[x._p for x in self.inputs])
        ops = (FusedOp * max(len(self.ops), 1))(*self.ops)
        return NDArray._wrap(h.NDArray_FusedChain(arr, len(self.inputs), ops, len(self.ops)))


def _reduce_method(name):
    def method(self, axis=None):
        """Flush the chain INTO a reduction — of everything (-> float) or over one axis (-> NDArray): the
        expression's values are never stored."""
        from ._lib import REDUCE_OPS
        h = _load_host()
        if axis is not None:
            h.NDArray_FusedChainReduceAxis.restype = _P
            h.NDArray_FusedChainReduceAxis.argtypes = [C.POINTER(_P), C.c_int, C.POINTER(FusedOp), C.c_int, C.c_int, C.c_int]
            arr = (_P * len(self.inputs))(*[x._p for x in self.inputs])
            ops = (FusedOp * max(len(self.ops), 1))(*self.ops)
            return NDArray._wrap(h.NDArray_FusedChainReduceAxis(arr, len(self.inputs), ops, len(self.ops),
                                                                REDUCE_OPS[name], int(axis)))
        h.NDArray_FusedChainReduce.restype = C.c_float
        h.NDArray_FusedChainReduce.argtypes = [C.POINTER(_P), C.c_int, C.POINTER(FusedOp), C.c_int, C.c_int]
        arr = (_P * len(self.inputs))(*[x._p for x in self.inputs])
        ops = (FusedOp * max(len(self.ops), 1))(*self.ops)
        h.numpower_host_clear_error()
        v = h.NDArray_FusedChainReduce(arr, len(self.inputs), ops, len(self.ops), REDUCE_OPS[name])
        msg = h.numpower_host_last_error()
        if msg:
            h.numpower_host_clear_error()
            raise Error(msg.decode())
        return float(v)
    return method


for _r in ("sum", "prod", "min", "max", "mean"):
    setattr(Lazy, _r, _reduce_method(_r))


def lazy(a: NDArray) -> Lazy:
    return Lazy(a)


NDArray.lazy = lambda self: Lazy(self)   # $a->lazy() in the PHP surface this stands in for


def _defer_to_lazy(name):
    """`$array (op) $lazy`: let the pending chain absorb the op (Lazy.__r<op>__) instead of
    NDArray's eager operator trying to coerce a Lazy."""
    eager = getattr(NDArray, name)

    def op(self, o):
        if isinstance(o, Lazy):
            return NotImplemented
        return eager(self, o)
    setattr(NDArray, name, op)


for _n in ("__add__", "__sub__", "__mul__", "__truediv__", "__mod__", "__pow__"):
    _defer_to_lazy(_n)
del _lib
