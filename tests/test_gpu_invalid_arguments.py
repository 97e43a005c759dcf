"""The C ABI called wrongly, on the GPU box: null pointers, unknown op / kind / reduction codes, impossible shapes.  Every call must
come back with NP_ERR_INVALID and a message — not crash, not launch, not poison the device — and the library must go on working
(the reference ignores every error at this boundary: cuda_math.cu launches whatever it is given)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NP_ERR_INVALID = -1


def test_wrong_calls_are_refused_and_nothing_breaks(hip):
    from numpower_amd import _lib
    from numpower_amd._lib import BINARY_OPS, UNARY_OPS, FusedOp
    lib = _lib.load()
    D = hip
    n = 1000
    a, b, out = D.DeviceArray((n,)), D.DeviceArray((n,)), D.DeviceArray((n,))
    D.fill(a, 1.5)
    D.fill(b, 2.0)
    host = C.c_float()
    flag = C.c_int()
    two = (C.c_float * 2)()
    ops1 = (FusedOp * 1)(FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0))
    ops99 = (FusedOp * 99)(*[FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0) for _ in range(99)])
    i99 = (C.c_int * 99)(*([2] * 99))
    p99 = (C.c_int * 99)(*range(99))
    bad_kind = (FusedOp * 1)(FusedOp(7, 0, 0, 0, 0, 0, 0, 0))
    bad_unary = (FusedOp * 1)(FusedOp(0, 999, 0, 0, 0, 0, 0, 0))
    bad_operand = (FusedOp * 1)(FusedOp(1, BINARY_OPS["add"], 5, 0, 0, 0, 0, 0))
    ptrs = (C.c_void_p * 2)(a.ptr, b.ptr)
    null_ptrs = (C.c_void_p * 2)(None, b.ptr)
    kinds = (C.c_int * 2)(0, 0)
    bad_kinds = (C.c_int * 2)(0, 9)
    i3 = lambda *v: (C.c_int * 3)(*v)        # noqa: E731
    ll2 = lambda *v: (C.c_longlong * 2)(*v)  # noqa: E731
    calls = {
        "binary: unknown op": lambda: lib.np_binary(999, a.ptr, 0, b.ptr, 0, out.ptr, 1, n, 0, 0),
        "binary: negative op": lambda: lib.np_binary(-1, a.ptr, 0, b.ptr, 0, out.ptr, 1, n, 0, 0),
        "binary: unknown kind": lambda: lib.np_binary(0, a.ptr, 9, b.ptr, 0, out.ptr, 1, n, 0, 0),
        "binary: null a": lambda: lib.np_binary(0, None, 0, b.ptr, 0, out.ptr, 1, n, 0, 0),
        "binary: null out": lambda: lib.np_binary(0, a.ptr, 0, b.ptr, 0, None, 1, n, 0, 0),
        "unary: unknown op": lambda: lib.np_unary(999, a.ptr, out.ptr, n, 0.0, 0.0),
        "unary: null in": lambda: lib.np_unary(0, None, out.ptr, n, 0.0, 0.0),
        "unary: null out": lambda: lib.np_unary(0, a.ptr, None, n, 0.0, 0.0),
        "chain: no inputs": lambda: lib.np_fused_chain(ptrs, kinds, 0, ops1, 1, out.ptr, 1, n),
        "chain: too many ops": lambda: lib.np_fused_chain(ptrs, kinds, 2, ops99, 99, out.ptr, 1, n),
        "chain: unknown step kind": lambda: lib.np_fused_chain(ptrs, kinds, 2, bad_kind, 1, out.ptr, 1, n),
        "chain: unknown unary op": lambda: lib.np_fused_chain(ptrs, kinds, 2, bad_unary, 1, out.ptr, 1, n),
        "chain: operand out of range": lambda: lib.np_fused_chain(ptrs, kinds, 2, bad_operand, 1, out.ptr, 1, n),
        "chain: null input": lambda: lib.np_fused_chain(null_ptrs, kinds, 2, ops1, 1, out.ptr, 1, n),
        "chain: unknown input kind": lambda: lib.np_fused_chain(ptrs, bad_kinds, 2, ops1, 1, out.ptr, 1, n),
        "chain: null out": lambda: lib.np_fused_chain(ptrs, kinds, 2, ops1, 1, None, 1, n),
        "chain reduce: unknown reduction": lambda: lib.np_fused_chain_reduce(ptrs, kinds, 2, ops1, 1, 42, 1, n, C.byref(host)),
        "chain reduce: null result": lambda: lib.np_fused_chain_reduce(ptrs, kinds, 2, ops1, 1, 0, 1, n, None),
        "chain reduce axis: bad axis": lambda: lib.np_fused_chain_reduce_axis(ptrs, kinds, 2, ops1, 1, 0, 10, 100, 2, out.ptr),
        "reduce_all: unknown op": lambda: lib.np_reduce_all(42, a.ptr, n, C.byref(host)),
        "reduce_all: null input": lambda: lib.np_reduce_all(0, None, n, C.byref(host)),
        "reduce_all: null result": lambda: lib.np_reduce_all(0, a.ptr, n, None),
        "reduce_axis: unknown op": lambda: lib.np_reduce_axis(42, a.ptr, 10, 10, 10, out.ptr, 0),
        "reduce_axis: null out": lambda: lib.np_reduce_axis(0, a.ptr, 10, 10, 10, None, 0),
        "argreduce: null input": lambda: lib.np_argreduce(1, None, 10, 10, 10, out.ptr),
        "moments: empty": lambda: lib.np_moments(a.ptr, 0, C.byref(host), C.byref(host)),
        "moments: null result": lambda: lib.np_moments(a.ptr, n, None, C.byref(host)),
        "weighted sums: null weights": lambda: lib.np_weighted_sums(a.ptr, None, n, C.byref(host), C.byref(host)),
        "order_stat: rank past the end": lambda: lib.np_order_stat(a.ptr, n, n, two),
        "order_stat: empty": lambda: lib.np_order_stat(a.ptr, 0, 0, two),
        "count_mismatch: unknown mode": lambda: lib.np_count_mismatch(7, a.ptr, b.ptr, n, 0.0, 0.0, C.byref(flag)),
        "count_mismatch: null result": lambda: lib.np_count_mismatch(0, a.ptr, b.ptr, n, 0.0, 0.0, None),
        "all: null result": lambda: lib.np_all(a.ptr, n, 0, None),
        "sgemm: null operand": lambda: lib.np_sgemm(10, 10, 10, None, b.ptr, out.ptr),
        "sgemm: null result": lambda: lib.np_sgemm(10, 10, 10, a.ptr, b.ptr, None),
        "sgemv: null vector": lambda: lib.np_sgemv(10, 10, a.ptr, None, out.ptr),
        "outer: null operand": lambda: lib.np_outer(None, 10, b.ptr, 10, out.ptr),
        "transpose: in place": lambda: lib.np_transpose2d(a.ptr, a.ptr, 1, 10, 10),
        "transpose: null": lambda: lib.np_transpose2d(None, out.ptr, 1, 10, 10),
        "permute: repeated axis": lambda: lib.np_permute(a.ptr, out.ptr, 3, i3(10, 10, 10), i3(0, 0, 1)),
        "permute: axis out of range": lambda: lib.np_permute(a.ptr, out.ptr, 3, i3(10, 10, 10), i3(0, 1, 3)),
        "permute: too many dimensions": lambda: lib.np_permute(a.ptr, out.ptr, 99, i99, p99),
        "permute: negative extent": lambda: lib.np_permute(a.ptr, out.ptr, 3, i3(10, -1, 10), i3(0, 1, 2)),
        "strided_copy: null strides": lambda: lib.np_strided_copy(a.ptr, out.ptr, 2, (C.c_int * 2)(10, 10), None),
        "copy2d: pitch below width": lambda: lib.np_copy2d(out.ptr, 5, a.ptr, 10, 10, 10),
        "fill: null": lambda: lib.np_fill(None, 1.0, n),
        "arange: null": lambda: lib.np_arange(None, 0.0, 1.0, n),
        "identity: null": lambda: lib.np_identity(None, 10),
        "read_float: null result": lambda: lib.np_read_float(a.ptr, 0, None),
        "memcpy_h2d: null host": lambda: lib.np_memcpy_h2d(a.ptr, None, 16),
        "memcpy_d2h: null device": lambda: lib.np_memcpy_d2h(two, None, 8),
        "malloc: null result": lambda: lib.np_malloc(None, 16),
        "free: a pointer that is not a block": lambda: lib.np_free(a.ptr + 64),
        "set_device: no such device": lambda: lib.np_set_device(4096),
    }
    wrong = {}
    for what, call in calls.items():
        rc = call()
        msg = lib.np_last_error()
        if rc != NP_ERR_INVALID or not msg:
            wrong[what] = (rc, msg)
    assert not wrong, wrong
    # nothing was launched, nothing is pending, the device is healthy and the operands are untouched
    assert lib.np_sync() == 0
    bits = C.c_uint(99)
    assert lib.np_clear_device_error(C.byref(bits)) == 0 and bits.value == 0
    assert (a.to_host() == np.float32(1.5)).all() and (b.to_host() == np.float32(2.0)).all()
    D.binary("add", a, "full", b, "full", 1, n, out=out)
    assert (out.to_host() == np.float32(3.5)).all()
    for d in (a, b, out):
        d.free()
