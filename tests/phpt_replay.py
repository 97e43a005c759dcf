"""Replay the reference's PHPT known-answer vectors (tests/golden/phpt_vectors.json) through a
back end and render the results exactly as PHP's print_r would, so the text can be compared with
the --EXPECT-- section byte for byte.

A back end provides the PHP-visible operations of class NDArray for the hot path:
    array(nested_list) -> handle          NDArray::array
    scalar(number) -> handle              int/float operand (0-d CPU scalar, numpower.c:93-98)
    row(handle, i) -> handle              $a[i]  (view of row i)
    binary(name, x, y) -> handle|float    operators / NDArray_*_Float
    unary(name, x, p0, p1) -> handle      NDArray::exp ... (Map family)
    reduce(name, x, axis|None) -> handle|float
    matmul(x, y) -> handle
    to_list(handle) -> nested list of Python floats (toArray())
Scalars (0-d results, full reductions) come back as Python floats (RETURN_NDARRAY,
numpower.c:137-150).
"""
from __future__ import annotations

import json
import math
from pathlib import Path

import numpy as np

VECTORS = Path(__file__).resolve().parent / "golden" / "phpt_vectors.json"

OPERATORS = {"+": "add", "-": "subtract", "*": "multiply", "/": "divide", "**": "pow", "%": "mod"}
REDUCTIONS = {"sum", "prod", "min", "max", "mean"}
COMPARISONS = {"equal", "not_equal", "greater", "greater_equal", "less", "less_equal"}


def load_vectors():
    return json.loads(VECTORS.read_text())["tests"]


# ---- PHP text rendering ---------------------------------------------------------------------

def php_float(v: float) -> str:
    """PHP's (string)$float with precision=14 (zend_gcvt): what print_r shows for a double."""
    v = float(v)
    if math.isnan(v):
        return "NAN"
    if math.isinf(v):
        return "INF" if v > 0 else "-INF"
    if v == 0.0:
        return "-0" if math.copysign(1.0, v) < 0 else "0"
    s = "%.14G" % v
    if "E" in s:
        mant, exp = s.split("E")
        if "." not in mant:
            mant += ".0"
        sign = exp[0]
        digits = exp[1:].lstrip("0") or "0"
        return "%sE%s%s" % (mant, sign, digits)
    return s


def print_r(value, indent: int = 0) -> str:
    """PHP print_r() of a float or a nested array of floats."""
    if not isinstance(value, list):
        return php_float(value)
    pad = " " * indent
    out = "Array\n" + pad + "(\n"
    for i, v in enumerate(value):
        out += pad + "    [%d] => " % i
        out += print_r(v, indent + 8)
        out += "\n"
    out += pad + ")\n"
    return out


# ---- replay -----------------------------------------------------------------------------------

def _operand(backend, env, spec):
    if "call" in spec:                      # nested static call
        return _call(backend, env, spec["call"])
    if "var" in spec:
        h = env[spec["var"]]
        if "index" in spec:
            h = backend.row(h, spec["index"])
        return h
    lit = spec["lit"]
    if isinstance(lit, list):
        return backend.array(lit)
    return backend.scalar(lit)


LAYOUT_OPS = {"reshape", "flatten", "expand_dims", "append", "trace", "allclose"}


def _call(backend, env, call):
    if call["kind"] == "var":
        return env[call["var"]]
    op = call["op"]
    kw = call.get("kwargs", {})
    if op in ("arange", "identity", "ones", "zeros"):     # initializers: plain numbers / a shape list
        lits = [a["lit"] for a in call["args"]]
        return getattr(backend, op)(*lits)
    if call["kind"] == "operator":
        args = [_operand(backend, env, a) for a in call["args"]]
        return backend.binary(OPERATORS[op], args[0], args[1])
    if op in ("reshape", "expand_dims"):    # second argument is a shape / axis list, not an array
        return getattr(backend, op)(_operand(backend, env, call["args"][0]), call["args"][1]["lit"])
    args = [_operand(backend, env, a) for a in call["args"]]
    if op in REDUCTIONS:
        return backend.reduce(op, args[0], kw.get("axis"))
    if op == "matmul":
        return backend.matmul(args[0], args[1])
    if op in COMPARISONS:
        return backend.binary(op, args[0], args[1])
    if op == "all":
        return backend.all(args[0])
    if op == "transpose":
        return backend.transpose(args[0])
    if op in ("flatten", "trace"):
        return getattr(backend, op)(args[0])
    if op in ("append", "allclose"):
        return getattr(backend, op)(args[0], args[1])
    if op == "square":     # PHP_METHOD(NDArray, square): Multiply_Float(nda, nda), numpower.c:3093
        return backend.binary("multiply", args[0], args[0])
    if op == "clip":
        return backend.unary("clip", args[0], float(kw["min"]), float(kw["max"]))
    if op == "round":
        return backend.unary("round", args[0], float(kw["precision"]), 0.0)
    return backend.unary(op, args[0], 0.0, 0.0)


def replay(backend, test) -> str:
    """Run one PHPT record through `backend`; returns the text PHP would have printed."""
    env = {name: backend.array(val) for name, val in test["vars"].items()}
    text = ""
    for call in test["calls"]:
        res = _call(backend, env, call)
        if "assign" in call:                    # $a = \\NDArray::f(...): no output
            env[call["assign"]] = res
            continue
        if call.get("printer") == "var_dump":
            text += "bool(%s)\n" % ("true" if res else "false")
        elif call["to_array"]:
            text += print_r(backend.to_list(res))
        else:
            text += print_r(float(res))
    return text


class GpuBackend:
    """Back end = the product path: numpower_amd.ndarray.NDArray (host library -> C ABI -> HIP
    kernels).  Every array operand is moved to the GPU first, exactly what a PHP script does with
    ->gpu(); results come back with ->cpu()->toArray()."""

    def __init__(self):
        from numpower_amd.ndarray import NDArray
        self.nd = NDArray

    def array(self, nested):
        return self.nd.array(nested).gpu()

    def scalar(self, v):
        return v   # int/float operands stay host scalars (ZVAL_TO_NDARRAY, numpower.c:93-98)

    def row(self, h, i):
        return h[i]

    def binary(self, name, x, y):
        return self.nd._binary(name, x, y)

    def unary(self, name, x, p0, p1):
        if name == "clip":
            return self.nd.clip(x, p0, p1)
        if name == "round":
            return self.nd.round(x, p0)
        return self.nd._unary(name, x)

    def reduce(self, name, x, axis):
        return self.nd._reduce(name, x, axis)

    def matmul(self, x, y):
        return self.nd.matmul(x, y)

    def all(self, x):
        return self.nd.all(x)

    def transpose(self, x):
        return self.nd.transpose(x)

    def reshape(self, x, shape):
        return self.nd.reshape(x, shape)

    def flatten(self, x):
        return self.nd.flatten(x)

    def expand_dims(self, x, axis):
        return self.nd.expand_dims(x, axis)

    def append(self, x, y):
        return self.nd.append(x, y)

    def trace(self, x):
        return self.nd.trace(x)

    def allclose(self, x, y):
        return self.nd.allclose(x, y)

    # initializers: born on the GPU
    def arange(self, stop, start=0.0, step=1.0):
        return self.nd.arange(stop, start, step, 1)

    def identity(self, size):
        return self.nd.identity(size, 1)

    def ones(self, shape):
        return self.nd.ones(shape, 1)

    def zeros(self, shape):
        return self.nd.zeros(shape, 1)

    def to_list(self, h):
        return (h.cpu() if h.isGPU() else h).toArray()


class OracleBackend:
    """Back end = the CPU restatement in oracle/ (numpy arrays stand for CPU NDArrays)."""

    def __init__(self):
        from oracle import oracle
        self.o = oracle

    def array(self, nested):
        return np.array(nested, dtype=np.float32)

    def scalar(self, v):
        return np.array(v, dtype=np.float32)

    def row(self, h, i):
        return h[i]

    def _ret(self, a):
        a = np.asarray(a, dtype=np.float32)
        return float(a) if a.ndim == 0 else a

    def binary(self, name, x, y):
        return self._ret(self.o.binary(name, x, y))

    def unary(self, name, x, p0, p1):
        return self._ret(self.o.unary(name, x, p0, p1))

    def reduce(self, name, x, axis):
        if axis is None:
            return float(self.o.reduce_all(name, x))
        return self._ret(self.o.reduce_axis(name, x, int(axis)))

    def matmul(self, x, y):
        return self._ret(self.o.matmul(x, y))

    def all(self, x):
        return float(self.o.reduce_all("all", x))

    def transpose(self, x):
        return self._ret(self.o.transpose(x))

    def reshape(self, x, shape):
        return self.o.reshape(x, shape)

    def flatten(self, x):
        return self.o.flatten(x)

    def expand_dims(self, x, axis):
        return self.o.expand_dims(x, axis)

    def append(self, x, y):
        return self.o.append(x, y)

    def trace(self, x):
        return float(self.o.trace(x))

    def allclose(self, x, y):
        return True if x is y else bool(self.o.allclose(x, y))   # identical handles: numpower.c:1372-1375

    def arange(self, stop, start=0.0, step=1.0):
        return self.o.arange(stop, start, step)

    def identity(self, size):
        return self.o.identity(size)

    def ones(self, shape):
        return self.o.full(shape, 1.0)

    def zeros(self, shape):
        return self.o.full(shape, 0.0)

    def to_list(self, h):
        return np.asarray(h, dtype=np.float64).tolist()
