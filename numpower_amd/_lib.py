"""ctypes binding of the C ABI in include/np_hip.h (libnp_hip.so).

This is the only place Python touches the device library.  There is no fallback: if the shared
library is missing, or a call fails, an exception is raised (the product path never routes
through numpy or the oracle).
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

LIBDIR = Path(__file__).resolve().parent / "lib"

# enums of include/np_hip.h -----------------------------------------------------------------
NP_OK = 0
BINARY_OPS = {"add": 0, "subtract": 1, "multiply": 2, "divide": 3, "mod": 4, "pow": 5,
              "arctan2": 6, "equal": 7, "not_equal": 8, "greater": 9, "greater_equal": 10,
              "less": 11, "less_equal": 12, "maximum": 13, "minimum": 14}
NP_FULL, NP_SCALAR, NP_ROW, NP_COL = 0, 1, 2, 3
NP_QUIRK_AVX_BODY = 1
UNARY_OPS = {name: i for i, name in enumerate([
    "abs", "sqrt", "exp", "exp2", "expm1", "log", "log2", "log10", "log1p", "logb",
    "sin", "cos", "tan", "arcsin", "arccos", "arctan", "degrees", "radians",
    "sinh", "cosh", "tanh", "arcsinh", "arccosh", "arctanh",
    "rint", "fix", "floor", "ceil", "trunc", "sinc", "negate", "sign",
    "clip", "round", "rsqrt", "positive", "reciprocal"])}
REDUCE_OPS = {"sum": 0, "prod": 1, "min": 2, "max": 3, "mean": 4}

_f32p = C.c_void_p   # device pointers travel as integers

NP_FUSED_UNARY, NP_FUSED_BINARY = 0, 1


class FusedOp(C.Structure):
    """np_fused_op of include/np_hip.h."""
    _fields_ = [("kind", C.c_int), ("op", C.c_int), ("operand", C.c_int), ("swap", C.c_int),
                ("p0", C.c_float), ("p1", C.c_float), ("flags", C.c_uint), ("body_end", C.c_size_t)]

# name -> (restype, argtypes).  Every symbol include/np_hip.h declares is listed here; the CPU
# test-suite checks that the built library exports each of them.
PROTOTYPES = {
    "np_init": (C.c_int, [C.c_int]),
    "np_set_device": (C.c_int, [C.c_int]),
    "np_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "np_sync": (C.c_int, []),
    "np_last_error": (C.c_char_p, []),
    "np_version": (C.c_char_p, []),
    "np_set_stream": (C.c_int, [C.c_void_p]),
    "np_get_stream": (C.c_void_p, []),
    "np_timer_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "np_timer_start": (C.c_int, [C.c_void_p]),
    "np_timer_stop": (C.c_int, [C.c_void_p]),
    "np_timer_elapsed_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "np_timer_destroy": (C.c_int, [C.c_void_p]),
    "np_graph_begin": (C.c_int, []),
    "np_graph_end": (C.c_int, [C.POINTER(C.c_void_p)]),
    "np_graph_launch": (C.c_int, [C.c_void_p]),
    "np_graph_destroy": (C.c_int, [C.c_void_p]),
    "np_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "np_free": (C.c_int, [C.c_void_p]),
    "np_live_allocs": (C.c_long, []),
    "np_pool_trim": (C.c_int, [C.POINTER(C.c_size_t)]),
    "np_pool_reserved_bytes": (C.c_size_t, []),
    "np_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "np_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "np_memcpy_d2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "np_memset0": (C.c_int, [C.c_void_p, C.c_size_t]),
    "np_fill": (C.c_int, [_f32p, C.c_float, C.c_size_t]),
    "np_read_float": (C.c_int, [_f32p, C.c_size_t, C.POINTER(C.c_float)]),
    "np_avx_body_end": (C.c_size_t, [C.c_size_t]),
    "np_binary": (C.c_int, [C.c_int, _f32p, C.c_int, _f32p, C.c_int, _f32p, C.c_size_t,
                            C.c_size_t, C.c_uint, C.c_size_t]),
    "np_unary": (C.c_int, [C.c_int, _f32p, _f32p, C.c_size_t, C.c_float, C.c_float]),
    "np_fused_chain": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.POINTER(FusedOp),
                                 C.c_int, _f32p, C.c_size_t, C.c_size_t]),
    "np_fused_chain_reduce": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.POINTER(FusedOp),
                                        C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.POINTER(C.c_float)]),
    "np_fused_chain_reduce_dev": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.POINTER(FusedOp),
                                            C.c_int, C.c_int, C.c_size_t, C.c_size_t, _f32p]),
    "np_fused_chain_reduce_axis": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int, C.POINTER(FusedOp),
                                             C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_int, _f32p]),
    "np_comm_init": (C.c_int, [C.c_int, C.c_int, C.c_char_p]),
    "np_comm_rank": (C.c_int, []),
    "np_comm_world": (C.c_int, []),
    "np_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "np_comm_max": (C.c_int, [C.c_float, C.POINTER(C.c_float)]),
    "np_comm_barrier": (C.c_int, []),
    "np_comm_destroy": (C.c_int, []),
    "np_comm_debug_exchange": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_void_p, C.c_double]),
    "np_allgather_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]),
    "np_comm_wait": (C.c_int, []),
    "np_comm_set_variant": (C.c_int, [C.c_int]),
    "np_comm_set_wait_limit": (C.c_int, [C.c_double]),
    "np_debug_hw_ids": (C.c_int, [C.POINTER(C.c_uint), C.c_size_t]),
    "np_debug_set_cus": (C.c_int, [C.c_int]),
    "np_comm_debug_model": (C.c_int, [C.c_int, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "np_comm_sync_mode": (C.c_int, []),
    "np_comm_piece": (C.c_int, [C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "np_comm_stream": (C.c_void_p, []),
    "np_sgemm_strided_batched_allgather": (C.c_int, [C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t,
                                                     C.c_void_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int]),
    "np_comm_rccl_version": (C.c_int, [C.POINTER(C.c_int)]),
    "np_comm_debug_sendrecv_self": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "np_comm_debug_loopback_timed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_float)]),
    "np_comm_debug_loopback": (C.c_int, [C.c_void_p, C.c_size_t]),
    "np_comm_debug_plan": (C.c_int, [C.c_int, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_ulonglong), C.c_size_t,
                                     C.POINTER(C.c_size_t)]),
    "np_reduce_all": (C.c_int, [C.c_int, _f32p, C.c_size_t, C.POINTER(C.c_float)]),
    "np_reduce_all_dev": (C.c_int, [C.c_int, _f32p, C.c_size_t, _f32p]),
    "np_all": (C.c_int, [_f32p, C.c_size_t, C.c_uint, C.POINTER(C.c_int)]),
    "np_copy2d": (C.c_int, [_f32p, C.c_size_t, _f32p, C.c_size_t, C.c_size_t, C.c_size_t]),
    "np_order_stat": (C.c_int, [_f32p, C.c_size_t, C.c_size_t, C.POINTER(C.c_float)]),
    "np_order_stat_dev": (C.c_int, [_f32p, C.c_size_t, C.c_size_t, _f32p]),
    "np_count_mismatch": (C.c_int, [C.c_int, _f32p, _f32p, C.c_size_t, C.c_float, C.c_float, C.POINTER(C.c_int)]),
    "np_argreduce": (C.c_int, [C.c_int, _f32p, C.c_size_t, C.c_size_t, C.c_size_t, _f32p]),
    "np_moments": (C.c_int, [_f32p, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "np_moments_dev": (C.c_int, [_f32p, C.c_size_t, _f32p]),
    "np_weighted_sums": (C.c_int, [_f32p, _f32p, C.c_size_t, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "np_reduce_axis": (C.c_int, [C.c_int, _f32p, C.c_size_t, C.c_size_t, C.c_size_t, _f32p,
                                 C.c_uint]),
    "np_reduce_axis_workspace": (C.c_size_t, [C.c_size_t, C.c_size_t, C.c_size_t]),
    "np_sgemm": (C.c_int, [C.c_size_t, C.c_size_t, C.c_size_t, _f32p, _f32p, _f32p]),
    "np_sgemm_strided_batched": (C.c_int, [C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                           _f32p, C.c_size_t, _f32p, C.c_size_t, _f32p,
                                           C.c_size_t]),
    "np_sgemm_strided_batched_piece": (C.c_int, [C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                                 _f32p, C.c_size_t, _f32p, C.c_size_t, _f32p, C.c_size_t]),
    "np_sgemv": (C.c_int, [C.c_size_t, C.c_size_t, _f32p, _f32p, _f32p]),
    "np_outer": (C.c_int, [_f32p, C.c_size_t, _f32p, C.c_size_t, _f32p]),
    "np_transpose2d": (C.c_int, [_f32p, _f32p, C.c_size_t, C.c_size_t, C.c_size_t]),
    "np_permute": (C.c_int, [_f32p, _f32p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "np_strided_copy": (C.c_int, [_f32p, _f32p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_longlong)]),
    "np_identity": (C.c_int, [_f32p, C.c_size_t]),
    "np_arange": (C.c_int, [_f32p, C.c_double, C.c_double, C.c_size_t]),
    "np_sgemm_set_variant": (C.c_int, [C.c_int]),
    "np_sgemm_debug_plan": (C.c_int, [C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    "np_elementwise_set_variant": (C.c_int, [C.c_int]),
    "np_layout_set_variant": (C.c_int, [C.c_int]),
    "np_reduce_set_variant": (C.c_int, [C.c_int]),
    "np_select_set_variant": (C.c_int, [C.c_int]),
    "np_runtime_set_variant": (C.c_int, [C.c_int]),
    "np_select_last_path": (C.c_int, [C.POINTER(C.c_int)]),
    "np_debug_raise_device_error": (C.c_int, [C.c_uint]),
    "np_debug_launch_count": (C.c_int, [C.POINTER(C.c_ulonglong)]),
    "np_clear_device_error": (C.c_int, [C.POINTER(C.c_uint)]),
    "np_debug_clock_mhz": (C.c_int, [C.POINTER(C.c_float)]),
}


class NumPowerError(RuntimeError):
    """Raised for every failed C-ABI call (mirrors the PHP `Error` the reference throws)."""


_lib = None


def lib_path() -> Path:
    # dev tools that force GEMM plans / time ablations ask for the -DNP_TUNING build explicitly
    # (python -m numpower_amd.build --tuning); nothing in the package or the tests does.
    import os
    if os.environ.get("NP_HIP_USE_TUNING_BUILD") == "1":
        return LIBDIR / "libnp_hip_tuning.so"
    if os.environ.get("NP_HIP_LIB"):            # dev A/B of two builds on one box (tools/ only)
        return Path(os.environ["NP_HIP_LIB"])
    return LIBDIR / "libnp_hip.so"


def load():
    """Load libnp_hip.so (built by numpower_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not path.exists():
        raise NumPowerError(
            f"{path} is missing: the HIP extension has not been built "
            "(run `python -m numpower_amd.build`); there is no CPU fallback")
    lib = C.CDLL(str(path), mode=C.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int):
    if rc != NP_OK:
        msg = load().np_last_error()
        raise NumPowerError(msg.decode() if msg else f"np_hip error {rc}")


class DeviceBuffer:
    """One np_malloc block (vmalloc/vfree pair of the reference, gpu_alloc.c:11-33)."""

    __slots__ = ("ptr", "nbytes")

    def __init__(self, nbytes: int):
        p = C.c_void_p()
        check(load().np_malloc(C.byref(p), nbytes))
        self.ptr = p.value or 0
        self.nbytes = nbytes

    def free(self):
        if self.ptr:
            check(load().np_free(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Timer:
    """hipEvent pair on the library stream."""

    def __init__(self):
        self._t = C.c_void_p()
        check(load().np_timer_create(C.byref(self._t)))

    def start(self):
        check(load().np_timer_start(self._t))

    def stop(self):
        check(load().np_timer_stop(self._t))

    def elapsed_ms(self) -> float:
        ms = C.c_float()
        check(load().np_timer_elapsed_ms(self._t, C.byref(ms)))
        return ms.value

    def __del__(self):
        try:
            load().np_timer_destroy(self._t)
        except Exception:
            pass
