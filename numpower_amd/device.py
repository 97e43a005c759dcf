"""Thin Python handle on device buffers + direct calls into the C ABI (include/np_hip.h).

Used by the parity tests and bench.py to drive the kernels exactly the way a C caller would:
raw device pointers, sizes, op codes.  numpy is used only to move host data in and out.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import (BINARY_OPS, NP_COL, NP_FULL, NP_QUIRK_AVX_BODY, NP_ROW, NP_SCALAR,
                   REDUCE_OPS, UNARY_OPS, check, load)

KINDS = {"full": NP_FULL, "scalar": NP_SCALAR, "row": NP_ROW, "col": NP_COL}


class DeviceArray:
    """Contiguous fp32 device buffer with a shape (no strides: same contract as the reference)."""

    def __init__(self, shape, buf: _lib.DeviceBuffer | None = None, offset_elems: int = 0,
                 base: "DeviceArray | None" = None):
        self.shape = tuple(int(s) for s in shape)
        self.size = int(np.prod(self.shape, dtype=np.int64)) if self.shape else 1
        if buf is None and base is None:
            buf = _lib.DeviceBuffer(max(self.size, 1) * 4)
        self._buf = buf if base is None else base._buf
        self._base = base
        self._offset = offset_elems + (base._offset if base is not None else 0)

    @property
    def ptr(self) -> int:
        return self._buf.ptr + 4 * self._offset

    @classmethod
    def from_host(cls, arr) -> "DeviceArray":
        a = np.ascontiguousarray(arr, dtype=np.float32)
        d = cls(a.shape)
        if a.size:
            check(load().np_memcpy_h2d(d.ptr, a.ctypes.data, a.nbytes))
        return d

    def to_host(self) -> np.ndarray:
        out = np.empty(self.shape, dtype=np.float32)
        if out.size:
            check(load().np_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes))
        return out

    def view(self, offset_elems: int, shape) -> "DeviceArray":
        """Sub-buffer view (e.g. one row of a matrix), shares storage."""
        return DeviceArray(shape, offset_elems=offset_elems, base=self)

    def free(self):
        if self._base is None and self._buf is not None:
            self._buf.free()


def init(device: int = 0):
    check(load().np_init(device))


def sync():
    check(load().np_sync())


def binary(op: str, a: DeviceArray, a_kind: str, b: DeviceArray, b_kind: str, rows: int,
           cols: int, quirk_numel_a: int | None = None, out: DeviceArray | None = None):
    """np_binary.  quirk_numel_a = numel of operand `a` as the reference's AVX loop bound sees it
    (enables NP_QUIRK_AVX_BODY); None = plain IEEE semantics."""
    if out is None:
        out = DeviceArray((rows, cols))
    flags, body_end = 0, 0
    if quirk_numel_a is not None:
        flags = NP_QUIRK_AVX_BODY
        body_end = load().np_avx_body_end(quirk_numel_a)
    check(load().np_binary(BINARY_OPS[op], a.ptr, KINDS[a_kind], b.ptr, KINDS[b_kind], out.ptr,
                           rows, cols, flags, body_end))
    return out


def unary(op: str, x: DeviceArray, p0: float = 0.0, p1: float = 0.0,
          out: DeviceArray | None = None):
    if out is None:
        out = DeviceArray(x.shape)
    check(load().np_unary(UNARY_OPS[op], x.ptr, out.ptr, x.size, p0, p1))
    return out


def reduce_all(op: str, x: DeviceArray) -> float:
    v = C.c_float()
    check(load().np_reduce_all(REDUCE_OPS[op], x.ptr, x.size, C.byref(v)))
    return v.value


def reduce_axis(op: str, x: DeviceArray, axis: int, quirk: bool = False,
                out: DeviceArray | None = None):
    shape = x.shape
    outer = int(np.prod(shape[:axis], dtype=np.int64)) if axis > 0 else 1
    inner = int(np.prod(shape[axis + 1:], dtype=np.int64)) if axis + 1 < len(shape) else 1
    if out is None:
        out = DeviceArray(shape[:axis] + shape[axis + 1:])
    check(load().np_reduce_axis(REDUCE_OPS[op], x.ptr, outer, shape[axis], inner, out.ptr,
                                NP_QUIRK_AVX_BODY if quirk else 0))
    return out


def sgemm(a: DeviceArray, b: DeviceArray, out: DeviceArray | None = None):
    m, k = a.shape
    k2, n = b.shape
    assert k == k2
    if out is None:
        out = DeviceArray((m, n))
    check(load().np_sgemm(m, n, k, a.ptr, b.ptr, out.ptr))
    return out


def sgemm_batched(a: DeviceArray, b: DeviceArray, out: DeviceArray | None = None):
    bt, m, k = a.shape
    bt2, k2, n = b.shape
    assert bt == bt2 and k == k2
    if out is None:
        out = DeviceArray((bt, m, n))
    check(load().np_sgemm_strided_batched(bt, m, n, k, a.ptr, m * k, b.ptr, k * n, out.ptr, m * n))
    return out


def sgemv(a: DeviceArray, x: DeviceArray, out: DeviceArray | None = None):
    m, n = a.shape
    if out is None:
        out = DeviceArray((m,))
    check(load().np_sgemv(m, n, a.ptr, x.ptr, out.ptr))
    return out


def transpose2d(x: DeviceArray, out: DeviceArray | None = None):
    rows, cols = x.shape
    if out is None:
        out = DeviceArray((cols, rows))
    check(load().np_transpose2d(x.ptr, out.ptr, 1, rows, cols))
    return out


def fill(x: DeviceArray, value: float):
    check(load().np_fill(x.ptr, value, x.size))
    return x
