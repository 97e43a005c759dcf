#!/usr/bin/env python3
"""apply_with_hip.py — turn a NumPower checkout into a `--with-hip` tree (INTEGRATION.md sections 2a and 2b, as code).

    python tools/apply_with_hip.py [--keep-cuda] <NumPower checkout> <output directory>

Copies the checkout to <output directory>, applies every edit of INTEGRATION.md section 2a by ANCHORED regex and
copies the glue (ext/*.c, ext/*.h, include/np_hip.h) to <output>/src/hip/.  Every edit names its file, the exact
text it expects to find and how many times; a missing or ambiguous anchor is a hard error (exit status 2) — the
table cannot silently rot when the reference moves.

**The default output is a HIP-only tree**: each CUDA-runtime / cuBLAS statement is REPLACED by its np_* / v* call — no
`#ifdef HAVE_NP_HIP ... #else <CUDA statement> #endif` pairs, no CUDA / HIP dual code path (BASELINE north star).  What
is left of CUDA in the tree is the reference's own src/gpu_alloc.c and src/ndmath/cuda/, which a `--with-hip` build does
not compile (config.m4: the glue takes their place), and the `--with-cuda` option text of config.m4.  check_tree() then
greps the RAW text of every C source the build compiles — comments and inactive preprocessor branches included — for
cuda[A-Z]*, cublas[A-Z]*, CUBLAS_*, <cuda_runtime.h>, <cublas_v2.h>: none may be left.

`--keep-cuda` keeps each replaced statement on an #else side instead:

    #ifdef HAVE_NP_HIP
        <the np_* / v* call>
    #else
        <the reference's CUDA-runtime statement, untouched>
    #endif

so that ONE tree still builds `--with-cuda` (a maintainer who wants to carry both back ends upstream); there the check
is the weaker one: no preprocessor-VISIBLE line of a --with-hip build may name CUDA.  Either way `--with-hip` (config.m4,
added by the last edits) defines HAVE_CUBLAS — the name the reference's C files gate every NDARRAY_DEVICE_GPU branch
on — and HAVE_NP_HIP.

Section 2b's edits INSERT (Edit.after) instead of replacing: the L2 functions that pick the device inside the function
(NDArray_{Add...Pow}_Float, the six comparisons, reduce()) gain `#ifdef HAVE_NP_HIP if (NPH_TAKES(a, b)) return
NPH_Binary_Float(...); #endif` behind their own device-mismatch check, so GPU operands cost one launch and CPU operands
fall through to the reference's code, untouched.  fast_path_program_source() wraps that inserted text into a C program
(numpower_amd/lib/fast_path_bodies) that the CPU and GPU test tiers run.

Section 2b's inserts (and section 2c's, the pending chains) stay inside `#ifdef HAVE_NP_HIP` in both modes: that guard
is not a second back end, it is what lets the same tree still configure WITHOUT a GPU (CPU-only build: no src/hip/ on
the include path).  src/gpu_alloc.c and src/ndmath/cuda/ are replaced wholesale by the glue and are not part of a
--with-hip build.

All file:line remarks refer to NumPower/numpower @ 2024_08_07.  Nothing of the reference is stored in this
repository: the tool holds anchors (regexes) and replacement text only, and tests/test_apply_with_hip_cpu.py runs it
on a scratch copy of /root/reference in the build container.
"""
from __future__ import annotations

import re
import shutil
import sys
from dataclasses import dataclass
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


class PatchError(Exception):
    pass


@dataclass
class Edit:
    file: str          # path inside the checkout
    what: str          # INTEGRATION.md row / reference line
    anchor: str        # regex (re.M); group "old" = the text that is wrapped / replaced
    new: str           # the HAVE_NP_HIP side (same indentation as the old text is applied automatically)
    expect: int = 1    # how many times the anchor must match
    wrap: bool = True  # True: a CUDA statement with a HIP counterpart — replaced by `new` (default), or with --keep-cuda
                       # `#ifdef HAVE_NP_HIP new #else old #endif`;  False: plain substitution by `new` in both modes
    guard: bool = False  # True: `old` is NOT a CUDA statement but device-independent reference code that a build without
                         # --with-hip must keep (reduce()'s slice loop, exp2's NDArray_Map): BOTH modes emit
                         # `#ifdef HAVE_NP_HIP new #else old #endif` — a feature guard, not a second GPU back end
    after: bool = False  # True: `old` stays where it is and `#ifdef HAVE_NP_HIP new #endif` is INSERTED behind it (the
                         # fast-path early-outs of section 2b: nothing of the reference is replaced)
    template: bool = False  # True: `new` holds \g<name> references to named groups of the anchor (expanded per match)
    context: str = ""  # C declarations of the locals of the surrounding reference function that `new` uses (CONTEXTS
                       # below): with them the new text is compiled on its own — snippet_check_source() — against
                       # include/np_hip.h, include/numpower_host.h and ext/hip_math.h; "" = nothing to compile


def _cuda_includes(file: str, line: str, more: str = "") -> Edit:
    return Edit(file, "%s: CUDA headers -> <np_hip.h>" % line,
                r"^(?P<old>#include <cuda_runtime\.h>\n#include <cublas_v2\.h>)$",
                "#include <np_hip.h>" + more)


# ---- section 2b: GPU early-outs at the top of the L2 functions that pick the device inside (ext/hip_fast.h) ----
_FAST_NOTE = ("/* numpower_amd: an array operand on the GPU leaves here through ONE np_binary launch — no Zeros + Fill scalar\n"
              " * temporary, no materialised NDArray_Broadcast; CPU operands fall through to the code below, untouched */\n")


def _fast_arith(name: str, op: str, line: str) -> Edit:
    return Edit("src/ndmath/arithmetics.c", "arithmetics.c:%s NDArray_%s_Float: GPU early-out" % (line, name),
                r"^(?P<old>NDArray_%s_Float\(NDArray\* a, NDArray\* b\) \{\n" % name +
                r"(?:[^\n]*\n){1,3}?"
                r"[ \t]*if \(NDArray_DEVICE\(a\) != NDArray_DEVICE\(b\) && NDArray_NDIM\(a\) != 0 && NDArray_NDIM\(b\) != 0\) \{\n"
                r"[ \t]*zend_throw_error\(NULL, \"Device mismatch, both NDArray MUST be in the same device\.\"\);\n"
                r"[ \t]*return NULL;\n"
                r"[ \t]*\})$",
                _FAST_NOTE +
                "if (NPH_TAKES(a, b)) {\n"
                "    return NPH_Binary_Float(%s, a, b);\n"
                "}" % op, after=True)


def _fast_compare(name: str, op: str, line: str) -> Edit:
    return Edit("src/logic.c", "logic.c:%s NDArray_%s: GPU early-out" % (line, name),
                r"^(?P<old>NDArray_%s\(NDArray\* nda, NDArray\* ndb\) \{\n" % name +
                r"(?:[^\n]*\n){1,3}?"
                r"[ \t]*if \(\(NDArray_DEVICE\(nda\) != NDArray_DEVICE\(ndb\)\) && NDArray_NDIM\(nda\) != 0 && NDArray_NDIM\(ndb\) != 0\) \{\n"
                r"[ \t]*zend_throw_error\(NULL, \"Devices mismatch in `equal` function\"\);\n"
                r"[ \t]*return NULL;\n"
                r"[ \t]*\})$",
                _FAST_NOTE +
                "if (NPH_TAKES(nda, ndb)) {\n"
                "    return NPH_Binary_Float(%s, nda, ndb);\n"
                "}" % op, after=True)


_SYNC = (r"^(?P<old>[ \t]*cudaDeviceSynchronize\(\);)$")

# ---- section 2c: pending chains (ext/hip_lazy.h) ----
_LAZY_FLUSH_NOTE = ("/* numpower_amd 2c: every consumer obtains its NDArray* here (ZVAL_TO_NDARRAY, ARRAY_OF_NDARRAYS, print_r_ ...): an\n"
                    " * array whose values are still a pending chain gets them now, in ONE fused launch, and so do the chains that read\n"
                    " * this array's buffer (the consumer may write it); appenders switch this off while they look their operands up */\n")
_LAZY_MARSHAL_NOTE = "/* numpower_amd 2c: an appender — its operands may stay pending chains (hip_lazy.h) */\n"
_LAZY_REDUCE_NOTE = ("/* numpower_amd 2c: a consumer that knows chains — a pending operand is reduced inside its chain's kernel (NPH_ReduceAll);\n"
                     " * the axis forms flush it (reduce() / single_reduce()) */\n")
# unary PHP_METHODs that call NDArrayMathGPU_ElementWise{,1F,2F}(nda, ...): 33 + clip + round in the reference, + exp2's
# new device branch (section 2a)
N_UNARY_APPENDERS = 36

EDITS = [
    # ---- headers ----
    _cuda_includes("numpower.c", "numpower.c:31-32", "\n#include <hip_lazy.h>"),
    # (found by the raw-text check of round 6: numpower.c:15 includes this header, and HAVE_CUBLAS — which --with-hip
    #  defines — pulled <cuda_runtime.h> in through it; rounds 4-5 only scanned *.c and src/)
    _cuda_includes("php_numpower.h", "php_numpower.h:9-10"),
    _cuda_includes("src/initializers.c", "initializers.c:15-16"),
    _cuda_includes("src/ndarray.c", "ndarray.c:18-19", "\n#include <hip_fast.h>\n#include <hip_lazy.h>\n#include \"ndmath/arithmetics.h\""),
    _cuda_includes("src/ndmath/arithmetics.c", "arithmetics.c:14-15", "\n#include <hip_fast.h>"),
    _cuda_includes("src/ndmath/linalg.c", "linalg.c:27-28"),
    _cuda_includes("src/manipulation.c", "manipulation.c:13-14"),
    _cuda_includes("src/debug.c", "debug.c:9-10"),
    # ---- NDArray_ToGPU: cudaMemcpy(H2D) + cudaDeviceSynchronize + error check (ndarray.c:1055-1060) ----
    Edit("src/ndarray.c", "ndarray.c:1055-1060 NDArray_ToGPU: H2D copy",
         r"^(?P<old>[ \t]*cudaMemcpy\(tmp_gpu, NDArray_FDATA\(target\), NDArray_NUMELEMENTS\(target\) \* sizeof\(float\), cudaMemcpyHostToDevice\);\n"
         r"[ \t]*cudaError_t err = cudaDeviceSynchronize\(\);\n"
         r"[ \t]*if \(err != cudaSuccess\) \{\n"
         r"[ \t]*zend_throw_error\(NULL, \"Error synchronizing: %s\\n\", cudaGetErrorString\(err\)\);\n"
         r"[ \t]*return NULL;\n"
         r"[ \t]*\})$",
         "if (np_memcpy_h2d(tmp_gpu, NDArray_FDATA(target), NDArray_NUMELEMENTS(target) * sizeof(float)) != NP_OK) {\n"
         "    zend_throw_error(NULL, \"%s\", np_last_error());\n"
         "    return NULL;\n"
         "}"),
    # ---- device -> host copies ----
    Edit("src/ndarray.c", "ndarray.c:1090 NDArray_ToCPU: D2H copy",
         r"^(?P<old>[ \t]*cudaMemcpy\(rtn->data, NDArray_FDATA\(target\), NDArray_NUMELEMENTS\(target\) \* sizeof\(float\), cudaMemcpyDeviceToHost\);)$",
         "np_memcpy_d2h(rtn->data, NDArray_FDATA(target), NDArray_NUMELEMENTS(target) * sizeof(float));"),
    Edit("src/ndarray.c", "ndarray.c:1021 NDArray_ToIntVector: one float back",
         r"^(?P<old>[ \t]*cudaMemcpy\(tmp_val, &NDArray_FDATA\(nda\)\[i\], sizeof\(float\), cudaMemcpyDeviceToHost\);)$",
         "np_memcpy_d2h(tmp_val, &NDArray_FDATA(nda)[i], sizeof(float));"),
    Edit("src/debug.c", "debug.c:201 print_matrix_float: D2H copy",
         r"^(?P<old>[ \t]*cudaMemcpy\(tmp_buffer, buffer, num_elements \* sizeof\(float\), cudaMemcpyDeviceToHost\);)$",
         "np_memcpy_d2h(tmp_buffer, buffer, num_elements * sizeof(float));"),
    Edit("src/ndmath/linalg.c", "linalg.c:680 singular values back to the host",
         r"^(?P<old>[ \t]*cudaMemcpy\(singular_values, NDArray_FDATA\(svd\[1\]\), sizeof\(float\) \* NDArray_NUMELEMENTS\(svd\[1\]\), cudaMemcpyDeviceToHost\);)$",
         "np_memcpy_d2h(singular_values, NDArray_FDATA(svd[1]), sizeof(float) * NDArray_NUMELEMENTS(svd[1]));"),
    # ---- NDArray_Zeros: cudaMemset (initializers.c:439,443) ----
    Edit("src/initializers.c", "initializers.c:439 NDArray_Zeros (double)",
         r"^(?P<old>[ \t]*cudaMemset\(rtn->data, 0, rtn->descriptor->numElements \* sizeof\(double\)\);)$",
         "np_memset0(rtn->data, rtn->descriptor->numElements * sizeof(double));"),
    Edit("src/initializers.c", "initializers.c:443 NDArray_Zeros (float)",
         r"^(?P<old>[ \t]*cudaMemset\(rtn->data, 0, rtn->descriptor->numElements \* sizeof\(float\)\);)$",
         "np_memset0(rtn->data, rtn->descriptor->numElements * sizeof(float));"),
    # ---- device -> device copies: np_memcpy_d2d(dst, src, bytes) with size_t bytes (vmemcpyd2d of gpu_alloc.h:10 takes
    #      the reference's `unsigned int`: 4 GiB) ----
    Edit("src/initializers.c", "initializers.c:758 NDArray_Copy: D2D copy",
         r"^(?P<old>[ \t]*cudaMemcpy\(NDArray_FDATA\(rtn\), NDArray_FDATA\(a\), NDArray_NUMELEMENTS\(a\) \* sizeof\(float\), cudaMemcpyDeviceToDevice\);)$",
         "np_memcpy_d2d(NDArray_FDATA(rtn), NDArray_FDATA(a), NDArray_NUMELEMENTS(a) * sizeof(float));"),
    Edit("src/ndmath/linalg.c", "linalg.c:145-146 NDArray_SVD: D2D copy + sync",
         r"^(?P<old>[ \t]*cudaMemcpy\(output_data, NDArray_FDATA\(target_ptr\), sizeof\(float\) \* NDArray_NUMELEMENTS\(target\), cudaMemcpyDeviceToDevice\);\n"
         r"[ \t]*cudaDeviceSynchronize\(\);)$",
         "np_memcpy_d2d(output_data, NDArray_FDATA(target_ptr), sizeof(float) * NDArray_NUMELEMENTS(target));"),
    # ---- gpu_alloc.h:8,10-11: byte counts as size_t.  Every call site passes `numElements * sizeof(float)` — a size_t
    #      expression that the reference's `unsigned int size` truncates modulo 4 GiB (a 5 GiB array would get a 1 GiB buffer
    #      and its kernels would write past it); one MI355X holds 288 GB.  The glue defines the three with size_t when
    #      NP_GPU_ALLOC_WIDE is set (the m4 block below sets it), and with the reference's types otherwise ----
    Edit("src/gpu_alloc.h", "gpu_alloc.h:8 vmalloc: size_t bytes",
         r"^(?P<old>void vmalloc\(void \*\*target, unsigned int size\);)$",
         "#include <stddef.h>\nvoid vmalloc(void **target, size_t size);"),
    Edit("src/gpu_alloc.h", "gpu_alloc.h:10-11 vmemcpyd2d / vmemcpyh2d: size_t bytes",
         r"^(?P<old>void vmemcpyd2d\(char\* target, char\* dst, unsigned int size\);\n"
         r"void vmemcpyh2d\(char\* target, char\* dst, unsigned int size\);)$",
         "void vmemcpyd2d(char* target, char* dst, size_t size);\nvoid vmemcpyh2d(char* target, char* dst, size_t size);"),
    # ---- element counts: the reference hands NDArray_NUMELEMENTS (a long) to the `int` of cuda_fill_float / cuda_equal_float /
    #      cuda_min|max_float / cuda_prod|sum_float (cuda_math.h:56-66): an array of 2^31 elements or more arrives as a negative
    #      count.  The seven call sites of the hot path call the C ABI with their size_t count instead (same kernels, same values) ----
    Edit("src/initializers.c", "initializers.c:639 NDArray_Fill: size_t count",
         r"^(?P<old>[ \t]*cuda_fill_float\(NDArray_FDATA\(a\), fill_value, NDArray_NUMELEMENTS\(a\)\);)$",
         "if (np_fill(NDArray_FDATA(a), fill_value, (size_t) NDArray_NUMELEMENTS(a)) != NP_OK) {\n"
         "    zend_throw_error(NULL, \"%s\", np_last_error());\n"
         "}"),
    Edit("src/logic.c", "logic.c:683 compare_ndarrays: size_t count",
         r"^(?P<old>[ \t]*diff = cuda_equal_float\(NDArray_NUMELEMENTS\(a\), NDArray_FDATA\(a\), NDArray_FDATA\(b\), NDArray_NUMELEMENTS\(a\)\);)$",
         "int any_mismatch = 1;\n"
         "if (np_count_mismatch(NP_MISMATCH_EXACT, NDArray_FDATA(a), NDArray_FDATA(b), (size_t) NDArray_NUMELEMENTS(a), 0.0f, 0.0f,\n"
         "                      &any_mismatch) != NP_OK) {\n"
         "    zend_throw_error(NULL, \"%s\", np_last_error());\n"
         "}\n"
         "diff = any_mismatch ? 0 : 1;"),
    Edit("src/ndarray.c", "ndarray.c:759 NDArray_Min: size_t count",
         r"^(?P<old>[ \t]*return cuda_min_float\(array, NDArray_NUMELEMENTS\(target\)\);)$",
         "min = 0.0f;\n"
         "if (np_reduce_all(NP_MIN, array, (size_t) NDArray_NUMELEMENTS(target), &min) != NP_OK) {\n"
         "    zend_throw_error(NULL, \"%s\", np_last_error());\n"
         "}\n"
         "return min;"),
    Edit("src/ndarray.c", "ndarray.c:946 NDArray_Max: size_t count",
         r"^(?P<old>[ \t]*return cuda_max_float\(array, NDArray_NUMELEMENTS\(target\)\);)$",
         "max = 0.0f;\n"
         "if (np_reduce_all(NP_MAX, array, (size_t) NDArray_NUMELEMENTS(target), &max) != NP_OK) {\n"
         "    zend_throw_error(NULL, \"%s\", np_last_error());\n"
         "}\n"
         "return max;"),
    Edit("src/ndmath/arithmetics.c", "arithmetics.c:41 NDArray_Float_Prod: size_t count",
         r"^(?P<old>[ \t]*cuda_prod_float\(NDArray_NUMELEMENTS\(a\), NDArray_FDATA\(a\), &value, NDArray_NUMELEMENTS\(a\)\);)$",
         "if (np_reduce_all(NP_PROD, NDArray_FDATA(a), (size_t) NDArray_NUMELEMENTS(a), &value) != NP_OK) {\n"
         "    zend_throw_error(NULL, \"%s\", np_last_error());\n"
         "}"),
    Edit("src/ndmath/arithmetics.c", "arithmetics.c:63,86 NDArray_Sum_Float / NDArray_Mean_Float: size_t count",
         r"^(?P<old>[ \t]*cuda_sum_float\(NDArray_NUMELEMENTS\(a\), NDArray_FDATA\(a\), &value, NDArray_NUMELEMENTS\(a\)\);)$",
         "if (np_reduce_all(NP_SUM, NDArray_FDATA(a), (size_t) NDArray_NUMELEMENTS(a), &value) != NP_OK) {\n"
         "    zend_throw_error(NULL, \"%s\", np_last_error());\n"
         "}", expect=2),
    # ---- the per-op cudaDeviceSynchronize() after the result allocation: arithmetics.c:218,497,633,758,883 ----
    Edit("src/ndmath/arithmetics.c", "arithmetics.c:218,497,633,758,883: sync after vmalloc (add, subtract, divide, mod, pow)",
         _SYNC, "/* nothing: the back end's stream orders the allocation with the kernels; read-backs block */", expect=5),
    # ---- NDArray_FMatmul: cublasCreate / cublasSgemm / cublasDestroy per call (linalg.c:55-71) ----
    Edit("src/ndmath/linalg.c", "linalg.c:55-71 NDArray_FMatmul: cuBLAS -> np_sgemm",
         r"^(?P<old>[ \t]*cublasHandle_t handle;\n"
         r"[ \t]*cublasCreate\(&handle\);\n"
         r"(?:.*\n)*?"
         r"[ \t]*cublasSgemm\(handle, CUBLAS_OP_N, CUBLAS_OP_N, n, m, k, &alpha, NDArray_FDATA\(b\), n, NDArray_FDATA\(a\), k, &beta, deviceResult, n\);\n"
         r"[ \t]*vfree\(result->data\);\n"
         r"[ \t]*result->data = \(void\*\)deviceResult;\n"
         r"[ \t]*cublasDestroy\(handle\);)$",
         "/* row-major C[m x n] = A[m x k] . B[k x n], straight into the result NDArray_Zeros allocated: no handle,\n"
         " * no second buffer */\n"
         "if (np_sgemm((size_t) NDArray_SHAPE(a)[0], (size_t) NDArray_SHAPE(b)[1], (size_t) NDArray_SHAPE(a)[1],\n"
         "             NDArray_FDATA(a), NDArray_FDATA(b), NDArray_FDATA(result)) != NP_OK) {\n"
         "    zend_throw_error(NULL, \"%s\", np_last_error());\n"
         "}"),
    # ---- NDArray::setDevice (numpower.c:621-634) ----
    Edit("numpower.c", "numpower.c:623-633 NDArray::setDevice",
         r"^(?P<old>[ \t]*// Get the number of available CUDA devices\n"
         r"[ \t]*cudaError_t cudaError = cudaGetDeviceCount\(&numDevices\);\n"
         r"\n"
         r"[ \t]*if \(cudaError != cudaSuccess\) \{\n"
         r"[ \t]*zend_throw_error\(NULL, \"Error getting the number of CUDA devices\.\\n\"\);\n"
         r"[ \t]*return;\n"
         r"[ \t]*\}\n"
         r"[ \t]*if \(deviceId >= 0 && deviceId > \(numDevices - 1\)\) \{\n"
         r"[ \t]*zend_throw_error\(NULL, \"Device %d does not exist\.\\n\", \(int\)deviceId\);\n"
         r"[ \t]*return;\n"
         r"[ \t]*\}\n"
         r"[ \t]*cudaSetDevice\(deviceId\);)$",
         "if (np_device_count(&numDevices) != NP_OK) {\n"
         "    zend_throw_error(NULL, \"Error getting the number of devices.\\n\");\n"
         "    return;\n"
         "}\n"
         "if (deviceId >= 0 && deviceId > (numDevices - 1)) {\n"
         "    zend_throw_error(NULL, \"Device %d does not exist.\\n\", (int)deviceId);\n"
         "    return;\n"
         "}\n"
         "if (np_set_device((int) deviceId) != NP_OK) {\n"
         "    zend_throw_error(NULL, \"%s\", np_last_error());\n"
         "    return;\n"
         "}"),
    # ---- NDArray_DumpDevices (debug.c:220-254): the whole CUDA property dump ----
    Edit("src/debug.c", "debug.c:220-254 NDArray_DumpDevices",
         r"^(?P<old>[ \t]*int deviceCount;\n"
         r"[ \t]*cudaError_t err = cudaGetDeviceCount\(&deviceCount\);\n"
         r"(?:.*\n)*?"
         r"[ \t]*printf\(\"\\n==============================================================================\\n\"\);)\n"
         r"(?=#else\n[ \t]*php_printf\(\"\\nNo GPU devices available\. CUDA not enabled\.\\n\"\);)",
         "int deviceCount = 0;\n"
         "if (np_device_count(&deviceCount) != NP_OK) {\n"
         "    printf(\"Failed to retrieve device count: %s\\n\", np_last_error());\n"
         "    return;\n"
         "}\n"
         "printf(\"\\nNumber of HIP devices: %d (%s)\\n\", deviceCount, np_version());"),
    # ---- rsqrt passes cuda_float_arccos (numpower.c:1791); exp2 has no device branch (numpower.c:3153) ----
    Edit("numpower.c", "numpower.c:33: declare the two unary entry points cuda_math.h lacks",
         r"^(?P<old>#include \"src/ndmath/cuda/cuda_math\.h\")$",
         "#include \"src/ndmath/cuda/cuda_math.h\"\n"
         "void cuda_float_rsqrt(int nblocks, float *d_array);   /* src/hip/hip_math.c */\n"
         "void cuda_float_exp2(int nblocks, float *d_array);"),
    Edit("numpower.c", "numpower.c:1791 PHP_METHOD(rsqrt): the right device function",
         # the same statement is correct in PHP_METHOD(arccos): pinned by the CPU branch just above it
         r"^[ \t]*rtn = NDArray_Map\(nda, float_rsqrt\);\n[ \t]*\} else \{\n#ifdef HAVE_CUBLAS\n"
         r"(?P<old>[ \t]*rtn = NDArrayMathGPU_ElementWise\(nda, cuda_float_arccos\);)$",
         "rtn = NDArrayMathGPU_ElementWise(nda, cuda_float_rsqrt);"),
    Edit("numpower.c", "numpower.c:3153 PHP_METHOD(exp2): a device branch",
         r"^(?P<old>[ \t]*rtn = NDArray_Map\(nda, float_exp2\);)$",
         "if (NDArray_DEVICE(nda) == NDARRAY_DEVICE_CPU) {\n"
         "    rtn = NDArray_Map(nda, float_exp2);\n"
         "} else {\n"
         "    rtn = NDArrayMathGPU_ElementWise(nda, cuda_float_exp2);\n"
         "}", guard=True),
    # ---- section 2b: the fast path, reachable from PHP without replacing a single reference symbol ----
    Edit("src/logic.c", "logic.c:10-11: <hip_fast.h>",
         r"^(?P<old>#include \"ndmath/cuda/cuda_math\.h\"\n#include \"debug\.h\")$",
         "#include <hip_fast.h>", after=True),
    _fast_arith("Add", "NP_ADD", "161-166"),
    _fast_arith("Multiply", "NP_MULTIPLY", "294-300"),
    _fast_arith("Subtract", "NP_SUBTRACT", "440-446"),
    _fast_arith("Divide", "NP_DIVIDE", "567-574"),
    _fast_arith("Mod", "NP_MOD", "701-707"),
    _fast_arith("Pow", "NP_POW", "826-832"),
    _fast_compare("Greater", "NP_GREATER", "68-73"),
    _fast_compare("Less", "NP_LESS", "172-177"),
    _fast_compare("LessEqual", "NP_LESS_EQUAL", "272-277"),
    _fast_compare("GreaterEqual", "NP_GREATER_EQUAL", "378-383"),
    _fast_compare("Equal", "NP_EQUAL", "479-484"),
    _fast_compare("NotEqual", "NP_NOT_EQUAL", "580-585"),
    Edit("src/ndarray.c", "ndarray.c:570 reduce(): one np_reduce_axis launch for GPU arrays",
         r"^(?P<old>[ \t]*_reduce\(0, 0, axis, array, rtn, operation\);)$",
         "/* numpower_amd: sum / prod over an axis of a GPU array is ONE np_reduce_axis launch into the result allocated\n"
         " * above, instead of one operation() + allocation + copy per slice (_reduce, ndarray.c:394-429); CPU arrays, any\n"
         " * other operation and a negative axis (which reduce() lets through, ndarray.c:534) keep the reference's loop */\n"
         "if (rtn != NULL && NDArray_DEVICE(array) == NDARRAY_DEVICE_GPU && *axis >= 0 &&\n"
         "    (operation == NDArray_Add_Float || operation == NDArray_Multiply_Float)) {\n"
         "    /* section 2c: PHP_METHOD(sum / prod / mean) look their operand up without flushing it — a sum over the last axis (or the\n"
         "     * first of a 2-d array) of a pending value runs INSIDE its chain's kernel; anything else computes the value first */\n"
         "    int nph_done = operation == NDArray_Add_Float ? NPH_ChainReduceAxisInto(array, *axis, NP_SUM, rtn) : 0;\n"
         "    if (nph_done < 0 || (nph_done == 0 && (NPH_Flush(array) != 0 ||\n"
         "        NPH_ReduceAxisInto(array, *axis, operation == NDArray_Add_Float ? NP_SUM : NP_PROD,\n"
         "                           operation == NDArray_Multiply_Float ? NP_QUIRK_AVX_BODY : 0u, rtn) != 0))) {\n"
         "        NDArray_FREE(rtn);\n"
         "        rtn = NULL;\n"
         "    }\n"
         "} else if (NPH_Flush(array) != 0) {\n"
         "    NDArray_FREE(rtn);\n"
         "    rtn = NULL;\n"
         "} else {\n"
         "    _reduce(0, 0, axis, array, rtn, operation);\n"
         "}", guard=True),
    Edit("src/ndarray.c", "ndarray.c:509 single_reduce(): mean over an axis of a GPU array",
         r"^(?P<old>[ \t]*_single_reduce\(0, 0, axis, array, rtn, operation\);)$",
         "/* numpower_amd: PHP_METHOD(mean) sends GPU arrays with an axis here (numpower.c:2677), and the loop below writes NOTHING\n"
         " * for them (apply_single_reduce stores only when the target is on the CPU, ndarray.c:389).  For a GPU array this is what\n"
         " * the method's CPU branch computes — reduce(Add) / n, numpower.c:2662-2669 — as one np_reduce_axis launch; every other\n"
         " * operation (min / max / median / all) and every CPU array keeps the reference's loop */\n"
         "if (rtn != NULL && NDArray_DEVICE(array) == NDARRAY_DEVICE_GPU && *axis >= 0 && operation == NDArray_Mean_Float) {\n"
         "    int nph_done = NPH_ChainReduceAxisInto(array, *axis, NP_MEAN, rtn);   /* section 2c: a pending operand, inside its chain's kernel */\n"
         "    if (nph_done < 0 || (nph_done == 0 && (NPH_Flush(array) != 0 || NPH_ReduceAxisInto(array, *axis, NP_MEAN, 0u, rtn) != 0))) {\n"
         "        NDArray_FREE(rtn);\n"
         "        rtn = NULL;\n"
         "    }\n"
         "} else if (NPH_Flush(array) != 0) {\n"
         "    NDArray_FREE(rtn);\n"
         "    rtn = NULL;\n"
         "} else {\n"
         "    _single_reduce(0, 0, axis, array, rtn, operation);\n"
         "}", guard=True),
    # ---- section 2c: pending elementwise chains behind the NDArray handle (ext/hip_lazy.h) ----
    # the flush point: the ONE function through which a PHP handle becomes an NDArray*
    Edit("src/buffer.c", "buffer.c:5-6: <hip_lazy.h>",
         r"^(?P<old>#include \"string\.h\"\n#include \"ndarray\.h\")$",
         "#include <hip_lazy.h>", after=True),
    Edit("src/buffer.c", "buffer.c:80-81 buffer_get: the flush point of pending chains",
         r"^(?P<old>NDArray\* buffer_get\(int uuid\) \{\n[ \t]*assert\(MAIN_MEM_STACK\.buffer\[uuid\] != NULL\);)$",
         _LAZY_FLUSH_NOTE +
         "NPH_OnBufferGet(MAIN_MEM_STACK.buffer[uuid]);", after=True),
    Edit("src/ndarray.c", "ndarray.c:588-591 NDArray_FREE: a pending array that dies releases its chain",
         r"^(?P<old>NDArray_FREE\(NDArray \*array\) \{\n[ \t]*if \(array == NULL \|\| array->refcount == -1\) \{\n[ \t]*return;\n[ \t]*\})$",
         "/* numpower_amd 2c: the last reference to an array whose values were never asked for drops its chain (and the\n"
         " * references the chain holds on its inputs) — no kernel ever ran for it */\n"
         "NPH_OnFree(array);", after=True),
    Edit("numpower.c", "numpower.c:5250 PHP_RINIT: no pending chain, appender scope closed",
         r"^(?P<old>[ \t]*buffer_init\(2\);)$",
         "/* numpower_amd 2c: whatever the previous request of this process left behind (a bailout inside an appender's lookup) */\n"
         "NPH_RequestInit();", after=True),
    # the appenders look their operands up without flushing them
    Edit("numpower.c", "numpower.c:194-195 ndarray_do_operation_ex: operands looked up as an appender",
         r"^(?P<old>[ \t]*NDArray \*nda = ZVAL_TO_NDARRAY\(op1\);\n[ \t]*NDArray \*ndb = ZVAL_TO_NDARRAY\(op2\);)$",
         _LAZY_MARSHAL_NOTE +
         "NPH_LAZY_MARSHAL_BEGIN();\n"
         "NDArray *nda = ZVAL_TO_NDARRAY(op1);\n"
         "NDArray *ndb = ZVAL_TO_NDARRAY(op2);\n"
         "NPH_LAZY_MARSHAL_END();", guard=True),
    Edit("numpower.c", "numpower.c:3374-3540 PHP_METHOD(add ... pow): operands looked up as an appender",
         r"^(?P<old>[ \t]*NDArray \*nda = ZVAL_TO_NDARRAY\(a\);\n[ \t]*NDArray \*ndb = ZVAL_TO_NDARRAY\(b\);)\n"
         r"(?=(?:(?!PHP_METHOD)[^\n]*\n){1,16}?[ \t]*rtn = NDArray_(?:Add|Subtract|Multiply|Divide|Mod|Pow)_Float\(nda, ndb\);)",
         _LAZY_MARSHAL_NOTE +
         "NPH_LAZY_MARSHAL_BEGIN();\n"
         "NDArray *nda = ZVAL_TO_NDARRAY(a);\n"
         "NDArray *ndb = ZVAL_TO_NDARRAY(b);\n"
         "NPH_LAZY_MARSHAL_END();", expect=6, guard=True),
    Edit("numpower.c", "numpower.c:1608-3357 the unary PHP_METHODs: operand looked up as an appender",
         r"^(?P<old>[ \t]*NDArray \*nda = ZVAL_TO_NDARRAY\(array\);)\n"
         r"(?=(?:(?!PHP_METHOD)[^\n]*\n){1,14}?[ \t]*rtn = NDArrayMathGPU_ElementWise(?:1F|2F)?\(nda, )",
         _LAZY_MARSHAL_NOTE +
         "NPH_LAZY_MARSHAL_BEGIN();\n"
         "NDArray *nda = ZVAL_TO_NDARRAY(array);\n"
         "NPH_LAZY_MARSHAL_END();", expect=N_UNARY_APPENDERS, guard=True),
    # ... and append instead of launching
] + [
    Edit("numpower.c", "numpower.c:200-218,3384-3550 `rtn = NDArray_%s_Float(nda, ndb)`: append to the pending chain" % _name,
         r"^(?P<old>[ \t]*rtn = NDArray_%s_Float\(nda, ndb\);)$" % _name,
         "rtn = NPH_LazyBinary(%s, NDArray_%s_Float, nda, ndb);" % (_op, _name), expect=2, guard=True)
    for _name, _op in (("Add", "NP_ADD"), ("Subtract", "NP_SUBTRACT"), ("Multiply", "NP_MULTIPLY"), ("Divide", "NP_DIVIDE"),
                       ("Pow", "NP_POW"), ("Mod", "NP_MOD"))
] + [
    Edit("numpower.c", "numpower.c:1651-3348 `rtn = NDArrayMathGPU_ElementWise{,1F,2F}(nda, cuda_float_*)`: append to the pending chain",
         # (not the CUDA side of a --keep-cuda pair: PHP_METHOD(rsqrt)'s `#else` keeps the reference's statement)
         r"(?<!#else\n)^(?P<old>[ \t]*rtn = NDArrayMathGPU_ElementWise(?P<sfx>1F|2F)?\(nda, (?P<rest>cuda_float_\w+[^;\n]*)\);)$",
         r"rtn = NPH_LazyElementWise\g<sfx>(nda, \g<rest>);", expect=N_UNARY_APPENDERS, template=True),
    # ... and the full reductions reduce a pending operand inside its chain's kernel
    Edit("numpower.c", "numpower.c:4630-4738 PHP_METHOD(sum, min, max, prod): operand looked up as a chain-aware consumer",
         r"^(?P<old>[ \t]*NDArray \*nda = ZVAL_TO_NDARRAY\(a\);)\n"
         r"(?=(?:(?!PHP_METHOD)[^\n]*\n){1,16}?[^\n]*= NDArray_(?:Sum_Float|Float_Prod|Min|Max)\(nda\);)",
         _LAZY_REDUCE_NOTE +
         "NPH_LAZY_MARSHAL_BEGIN();\n"
         "NDArray *nda = ZVAL_TO_NDARRAY(a);\n"
         "NPH_LAZY_MARSHAL_END();", expect=4, guard=True),
    Edit("numpower.c", "numpower.c:2653 PHP_METHOD(mean): operand looked up as a chain-aware consumer",
         r"^(?P<old>[ \t]*NDArray \*nda = ZVAL_TO_NDARRAY\(array\);)\n"
         r"(?=(?:(?!PHP_METHOD)[^\n]*\n){1,8}?[ \t]*RETURN_DOUBLE\(\(NDArray_Sum_Float\(nda\) / NDArray_NUMELEMENTS\(nda\)\)\);)",
         _LAZY_REDUCE_NOTE +
         "NPH_LAZY_MARSHAL_BEGIN();\n"
         "NDArray *nda = ZVAL_TO_NDARRAY(array);\n"
         "NPH_LAZY_MARSHAL_END();", guard=True),
    Edit("numpower.c", "numpower.c:2660,2675 PHP_METHOD(mean): `NDArray_Sum_Float(nda) / n` inside the chain's kernel",
         r"^(?P<old>[ \t]*RETURN_DOUBLE\(\(NDArray_Sum_Float\(nda\) / NDArray_NUMELEMENTS\(nda\)\)\);)$",
         "RETURN_DOUBLE((NPH_ReduceAll(NP_SUM, NDArray_Sum_Float, nda) / NDArray_NUMELEMENTS(nda)));", expect=2, guard=True),
] + [
    Edit("numpower.c", "numpower.c:%s `%s = %s(nda)`: reduce inside the chain's kernel" % (_line, _lhs, _fn),
         r"^(?P<old>[ \t]*%s = %s\(nda\);)$" % (re.escape(_lhs), _fn),
         "%s = NPH_ReduceAll(%s, %s, nda);" % (_lhs, _op, _fn), guard=True)
    for _line, _lhs, _fn, _op in (("4638", "double value", "NDArray_Sum_Float", "NP_SUM"), ("4673", "value", "NDArray_Min", "NP_MIN"),
                                  ("4712", "value", "NDArray_Max", "NP_MAX"), ("4744", "value", "NDArray_Float_Prod", "NP_PROD"))
] + [
    # ---- config.m4: the option, and the source list ----
    Edit("config.m4", "config.m4:7-8: --with-hip next to --with-cuda",
         r"^(?P<old>PHP_ARG_WITH\(cuda, for CUDA support,\n\[  --with-cuda           Include CUDA support\], \[no\], \[no\]\))$",
         "PHP_ARG_WITH(cuda, for CUDA support,\n[  --with-cuda           Include CUDA support], [no], [no])\n\n@HIP_M4_BLOCK@",
         wrap=False),
    Edit("config.m4", "config.m4:128: src/gpu_alloc.c -> the glue when --with-hip",
         r"^(?P<old>      src/gpu_alloc\.c \\)$",
         "      $NP_GPU_ALLOC_SOURCES \\", wrap=False),
]

# What each compiled snippet needs from the reference function around it (names and types as in the reference).
CONTEXTS = {
    "ndarray.c:1055-1060 NDArray_ToGPU: H2D copy": "float *tmp_gpu = 0; NDArray *target = 0;",
    "ndarray.c:1090 NDArray_ToCPU: D2H copy": "NDArray *rtn = 0, *target = 0;",
    "ndarray.c:1021 NDArray_ToIntVector: one float back": "double *tmp_val = 0; NDArray *nda = 0; int i = 0;",
    "debug.c:201 print_matrix_float: D2H copy": "float *tmp_buffer = 0, *buffer = 0; int num_elements = 0;",
    "linalg.c:680 singular values back to the host": "float *singular_values = 0; NDArray *svd[3] = {0, 0, 0};",
    "initializers.c:439 NDArray_Zeros (double)": "NDArray *rtn = 0;",
    "initializers.c:443 NDArray_Zeros (float)": "NDArray *rtn = 0;",
    "initializers.c:758 NDArray_Copy: D2D copy": "NDArray *rtn = 0, *a = 0;",
    "linalg.c:145-146 NDArray_SVD: D2D copy + sync": "float *output_data = 0; NDArray *target_ptr = 0, *target = 0;",
    "linalg.c:55-71 NDArray_FMatmul: cuBLAS -> np_sgemm": "NDArray *a = 0, *b = 0, *result = 0;",
    "initializers.c:639 NDArray_Fill: size_t count": "NDArray *a = 0; float fill_value = 0;",
    "logic.c:683 compare_ndarrays: size_t count": "NDArray *a = 0, *b = 0; int diff = 1;",
    "ndarray.c:759 NDArray_Min: size_t count": "NDArray *target = 0; float *array = 0; float min;",
    "ndarray.c:946 NDArray_Max: size_t count": "NDArray *target = 0; float *array = 0; float max;",
    "arithmetics.c:41 NDArray_Float_Prod: size_t count": "NDArray *a = 0; float value = 1;",
    "arithmetics.c:63,86 NDArray_Sum_Float / NDArray_Mean_Float: size_t count": "NDArray *a = 0; float value = 0;",
    "numpower.c:623-633 NDArray::setDevice": "int numDevices = 0; long deviceId = 0;",
    "debug.c:220-254 NDArray_DumpDevices": " ",
    "numpower.c:1791 PHP_METHOD(rsqrt): the right device function": "NDArray *rtn = 0, *nda = 0;",
    "numpower.c:3153 PHP_METHOD(exp2): a device branch": "NDArray *rtn = 0, *nda = 0;",
    "ndarray.c:570 reduce(): one np_reduce_axis launch for GPU arrays":
        "NDArray *array = 0, *rtn = 0; int *axis = 0; NDArray *(*operation)(NDArray *, NDArray *) = 0;",
    "ndarray.c:509 single_reduce(): mean over an axis of a GPU array":
        "NDArray *array = 0, *rtn = 0; int *axis = 0; float (*operation)(NDArray *) = 0;",
}
CONTEXTS.update({
    "buffer.c:80-81 buffer_get: the flush point of pending chains": "struct { NDArray **buffer; } MAIN_MEM_STACK = {0}; int uuid = 0;",
    "ndarray.c:588-591 NDArray_FREE: a pending array that dies releases its chain": "NDArray *array = 0;",
    "numpower.c:194-195 ndarray_do_operation_ex: operands looked up as an appender": "zval *op1 = 0, *op2 = 0;",
    "numpower.c:5250 PHP_RINIT: no pending chain, appender scope closed": " ",
    "numpower.c:3374-3540 PHP_METHOD(add ... pow): operands looked up as an appender": "zval *a = 0, *b = 0;",
    "numpower.c:1608-3357 the unary PHP_METHODs: operand looked up as an appender": "zval *array = 0;",
    "numpower.c:1651-3348 `rtn = NDArrayMathGPU_ElementWise{,1F,2F}(nda, cuda_float_*)`: append to the pending chain":
        "NDArray *rtn = 0, *nda = 0;",
})
CONTEXTS.update({
    "numpower.c:4630-4738 PHP_METHOD(sum, min, max, prod): operand looked up as a chain-aware consumer": "zval *a = 0;",
    "numpower.c:2653 PHP_METHOD(mean): operand looked up as a chain-aware consumer": "zval *array = 0;",
})
for _e in EDITS:
    if "reduce inside the chain's kernel" in _e.what:
        CONTEXTS[_e.what] = "NDArray *nda = 0;" + ("" if _e.what.startswith("numpower.c:4638") else " double value = 0;")
for _e in EDITS:
    if "`rtn = NDArray_" in _e.what:
        CONTEXTS[_e.what] = "NDArray *rtn = 0, *nda = 0, *ndb = 0;"
for _e in EDITS:
    if _e.what.endswith("GPU early-out"):
        CONTEXTS[_e.what] = "NDArray *nda = 0, *ndb = 0;" if _e.file == "src/logic.c" else "NDArray *a = 0, *b = 0;"

HIP_M4_BLOCK = '''dnl ---- MI355X (gfx950) through numpower_amd: added by numpower_amd/tools/apply_with_hip.py ----
dnl No device compiler step: the kernels live in a prebuilt libnp_hip.so, everything compiled here is plain C
dnl (src/hip/*.c), so the stock phpize / libtool flow builds it (no Makefile.frag, no nvcc).
PHP_ARG_WITH([hip],
  [for MI355X (HIP, gfx950) support through numpower_amd],
  [AS_HELP_STRING([--with-hip=DIR],
    [Run NDArray GPU paths on AMD MI355X; DIR = directory holding libnp_hip.so])],
  [no], [no])

NP_GPU_ALLOC_SOURCES="src/gpu_alloc.c"
if test "$PHP_HIP" != "no"; then
  if test "$PHP_CUDA" != "no"; then
    AC_MSG_ERROR([--with-hip and --with-cuda are mutually exclusive])
  fi
  if test "$PHP_HIP" = "yes"; then
    AC_MSG_ERROR([--with-hip needs the directory of libnp_hip.so: --with-hip=/path/to/numpower_amd/lib])
  fi
  if test ! -f "$PHP_HIP/libnp_hip.so"; then
    AC_MSG_ERROR([$PHP_HIP/libnp_hip.so not found: run `python -m numpower_amd.build` first])
  fi
  PHP_ADD_INCLUDE([$abs_srcdir/src/hip])
  PHP_ADD_LIBRARY_WITH_PATH([np_hip], [$PHP_HIP], [NDARRAY_SHARED_LIBADD])
  PHP_CHECK_LIBRARY([np_hip], [np_sgemm],
    [AC_MSG_RESULT([numpower_amd device back end detected])],
    [AC_MSG_ERROR([libnp_hip.so does not export np_sgemm])],
    [-L$PHP_HIP])
  AC_DEFINE([HAVE_CUBLAS], [1], [a device back end is present (the C files gate every GPU branch on this name)])
  AC_DEFINE([HAVE_NP_HIP], [1], [the device back end is numpower_amd / MI355X])
  CFLAGS+=" -DNUMPOWER_NDARRAY_HEADER='\\"src/initializers.h\\"' -DNP_GPU_ALLOC_WIDE=1 "
  NP_GPU_ALLOC_SOURCES="src/hip/gpu_alloc_hip.c src/hip/hip_math.c src/hip/hip_math_drivers.c src/hip/hip_fast.c src/hip/hip_lazy.c src/hip/zend_hooks.c"
fi'''

GLUE_FILES = ["ext/gpu_alloc_hip.c", "ext/hip_math.c", "ext/hip_math.h", "ext/hip_math_drivers.c", "ext/hip_fast.c",
              "ext/hip_fast.h", "ext/hip_lazy.c", "ext/hip_lazy.h", "ext/zend_hooks.c", "ext/np_ext_hooks.h", "include/np_hip.h"]
# not compiled in a --with-hip build: replaced wholesale by the glue
REPLACED_BY_GLUE = ("src/gpu_alloc.c", "src/ndmath/cuda/")


for _e in EDITS:
    _e.context = CONTEXTS.get(_e.what, "")
assert all(k in {e.what for e in EDITS} for k in CONTEXTS), "CONTEXTS names an edit that does not exist"


def snippet_check_source() -> str:
    """One C translation unit holding the HAVE_NP_HIP side of every statement-level edit, each in a function of its own
    with the locals it uses declared as in the reference.  `gcc -fsyntax-only -Wall -Werror` on it (tests/
    test_apply_with_hip_cpu.py) proves that the new statements are well-typed against the C ABI and the glue headers —
    a wrong argument order, a missing status check's type, a misspelt entry point fail THERE, not in a maintainer's
    PHP build.  The only foreign declarations are the three reference symbols the new text itself calls."""
    out = ["/* generated by tools/apply_with_hip.py: snippet_check_source() */",
           "#include <stdio.h>", "#include <stddef.h>",
           '#include "np_hip.h"', '#include "numpower_host.h"', '#include "hip_math.h"', '#include "hip_fast.h"', '#include "hip_lazy.h"',
           "void zend_throw_error(void *exception_ce, const char *format, ...);   /* Zend/zend_exceptions.h */",
           "typedef struct _zval_struct zval;                                        /* Zend/zend_types.h */",
           "NDArray *ZVAL_TO_NDARRAY(zval *obj);                                     /* numpower.c:89 */",
           "void _reduce(int current_axis, int rtn_init, int *axis, NDArray *target, NDArray *rtn,",
           "             NDArray *(*operation)(NDArray *, NDArray *));                 /* src/ndarray.c:394 */",
           "void _single_reduce(int current_axis, int rtn_init, int *axis, NDArray *target, NDArray *rtn,",
           "                    float (*operation)(NDArray *));                        /* src/ndarray.c:431 */",
           "NDArray *NDArray_Map(NDArray *array, float (*op)(float));                /* src/ndarray.h */",
           "float float_exp2(float val);                                             /* src/ndmath/double_math.h */", ""]
    for k, e in enumerate(EDITS):
        if not e.context:
            continue
        ret = ("void *" if ("return NULL;" in e.new or "return NPH_" in e.new) else
               "float " if ("return min;" in e.new or "return max;" in e.new) else "void ")   # NDArray_Min / NDArray_Max return a float
        out.append("/* %s */" % e.what)
        out.append("%ssnippet_%d(void) {" % (ret, k))
        out.append("    " + e.context)
        new = e.new
        if e.template:   # one instance of a templated edit: the exp method's call
            new = new.replace("\\g<sfx>", "").replace("\\g<rest>", "cuda_float_exp")
        out.append(_indent(new, "    "))
        if ret == "void *":
            out.append("    return (void *) 0;")
        out.append("}")
        out.append("")
    return "\n".join(out)


FAST_BINARY = [(e.what.split(" ")[1].rstrip(":"), e) for e in EDITS if e.what.endswith("GPU early-out")]   # ("NDArray_Add_Float", edit)


def fast_path_program_source() -> str:
    """A C99 PROGRAM around the text section 2b inserts: for each early-out a function with the reference function's
    signature whose body is the inserted text VERBATIM (as EDITS holds it) followed by a stand-in for "the reference's code
    below the insertion" that only counts how often it is reached; for reduce() the reference's result allocation, then the
    edited statement with `_reduce` as such a counter.  Built against include/numpower_host.h + ext/hip_fast.h and linked
    with libnumpower_host.so (which carries ext/hip_fast.c), -Wall -Wextra -Werror.

        fast_path_bodies cpu          every patched function called with CPU operands: each call must FALL THROUGH to the
                                      stand-in (BASELINE config 1 still reaches arithmetics.c) and no device is touched —
                                      runs without a GPU (tests/test_apply_with_hip_cpu.py)
        fast_path_bodies gpu <file>   the same calls with ->gpu() operands: none may fall through; results are written to
                                      <file> (method_bodies.c's record format) and checked against the oracle by
                                      tests/test_gpu_fast_path.py
    """
    reduce_edit = next(e for e in EDITS if e.what.startswith("ndarray.c:570 reduce()"))
    single_edit = next(e for e in EDITS if e.what.startswith("ndarray.c:509 single_reduce()"))
    o = ["/* generated by tools/apply_with_hip.py: fast_path_program_source() — do not edit */",
         "#define _POSIX_C_SOURCE 200809L",
         "#include <stdint.h>", "#include <stdio.h>", "#include <stdlib.h>", "#include <string.h>", "",
         "#define HAVE_NP_HIP 1", '#include "numpower_host.h"', '#include "hip_fast.h"', '#include "hip_lazy.h"', "",
         "static int g_fell_through;   /* calls that reached the reference's own code (the stand-ins below) */",
         "static NDArray *reference_body(void) { g_fell_through++; return NULL; }",
         "static void _reduce(int current_axis, int rtn_init, int *axis, NDArray *target, NDArray *rtn,",
         "                    NDArray *(*operation)(NDArray *, NDArray *)) {",
         "    (void) current_axis; (void) rtn_init; (void) axis; (void) target; (void) rtn; (void) operation;",
         "    g_fell_through++;", "}",
         "static void _single_reduce(int current_axis, int rtn_init, int *axis, NDArray *target, NDArray *rtn, float (*operation)(NDArray *)) {",
         "    (void) current_axis; (void) rtn_init; (void) axis; (void) target; (void) rtn; (void) operation;",
         "    g_fell_through++;", "}", ""]
    for name, e in FAST_BINARY:
        args = "NDArray *nda, NDArray *ndb" if e.file == "src/logic.c" else "NDArray *a, NDArray *b"
        o += ["static NDArray *patched_%s(%s) {" % (name, args), "#ifdef HAVE_NP_HIP", _indent(e.new, "    "), "#endif",
              "    return reference_body();", "}", ""]
    o += ["static NDArray *patched_reduce(NDArray *array, int *axis, NDArray *(*operation)(NDArray *, NDArray *)) {",
          "    /* reduce()'s own shape arithmetic and result allocation (ndarray.c:541-569), then the edited statement */",
          "    int out_shape[8], j = 0;",
          "    for (int i = 0; i < NDArray_NDIM(array); i++) if (i != *axis) out_shape[j++] = NDArray_SHAPE(array)[i];",
          '    NDArray *rtn = NDArray_Zeros(out_shape, j, "float32", NDArray_DEVICE(array));',
          "#ifdef HAVE_NP_HIP", _indent(reduce_edit.new, "    "), "#endif", "    return rtn;", "}", "",
          "static NDArray *patched_single_reduce(NDArray *array, int *axis, float (*operation)(NDArray *)) {",
          "    /* single_reduce()'s own result allocation (ndarray.c:476-508), then the edited statement */",
          "    int out_shape[8], j = 0;",
          "    for (int i = 0; i < NDArray_NDIM(array); i++) if (i != *axis) out_shape[j++] = NDArray_SHAPE(array)[i];",
          '    NDArray *rtn = NDArray_Zeros(out_shape, j, "float32", NDArray_DEVICE(array));',
          "#ifdef HAVE_NP_HIP", _indent(single_edit.new, "    "), "#endif", "    return rtn;", "}", ""]
    table = ",\n".join('    {"%s", patched_%s}' % (name[len("NDArray_"):], name) for name, _ in FAST_BINARY)
    o.append(_FAST_PROGRAM_MAIN.replace("@TABLE@", table))
    return "\n".join(o)


_FAST_PROGRAM_MAIN = r'''typedef NDArray *(*Binary)(NDArray *, NDArray *);
static const struct { const char *name; Binary fn; } kPatched[] = {
@TABLE@
};
enum { kCount = (int) (sizeof kPatched / sizeof kPatched[0]) };

static FILE *g_out;
static int g_failed;

/* method_bodies.c's record: 32-byte name, int32 ndim, int32 dims[4], the floats */
static void dump(const char *op, const char *form, NDArray *a) {
    char label[32];
    int32_t head[5] = {0, 1, 1, 1, 1};
    if (a == NULL) {
        fprintf(stderr, "fast_path_bodies: %s.%s returned NULL: %s\n", op, form, numpower_host_last_error());
        g_failed = 1;
        return;
    }
    memset(label, 0, sizeof label);
    snprintf(label, sizeof label, "%s.%s", op, form);
    head[0] = NDArray_NDIM(a);
    for (int i = 0; i < NDArray_NDIM(a) && i < 4; i++) head[1 + i] = NDArray_SHAPE(a)[i];
    NDArray *host = NDArray_ToCPU(a);
    if (host == NULL) {
        fprintf(stderr, "fast_path_bodies: cpu() of %s failed: %s\n", label, numpower_host_last_error());
        g_failed = 1;
        return;
    }
    fwrite(label, 1, sizeof label, g_out);
    fwrite(head, sizeof(int32_t), 5, g_out);
    fwrite(NDArray_FDATA(host), sizeof(float), (size_t) NDArray_NUMELEMENTS(host), g_out);
    NDArray_FREE(host);
}

/* x[i] = lo + (hi - lo) * frac(i * 0.6180339887 + seed * 0.37): what tests/test_gpu_method_bodies.py::c_input rebuilds */
static NDArray *input(const int *shape, int ndim, int seed, float lo, float hi) {
    long n = 1;
    for (int i = 0; i < ndim; i++) n *= shape[i];
    float *host = (float *) malloc(sizeof(float) * (size_t) n);
    for (long i = 0; i < n; i++) {
        double t = (double) i * 0.6180339887 + (double) seed * 0.37;
        t -= (double) (long) t;
        host[i] = (float) ((double) lo + ((double) hi - (double) lo) * t);
    }
    NDArray *cpu = NDArray_FromHostBuffer(host, shape, ndim);
    free(host);
    return cpu;
}

static NDArray *placed(NDArray *host, int on_gpu) {   /* $a->gpu() when asked; the caller keeps `host` */
    if (!on_gpu) return host;
    NDArray *dev = NDArray_ToGPU(host);
    if (dev == NULL) {
        fprintf(stderr, "fast_path_bodies: gpu() failed: %s\n", numpower_host_last_error());
        exit(1);
    }
    return dev;
}

int main(int argc, char **argv) {
    const int gpu = argc >= 2 && strcmp(argv[1], "gpu") == 0;
    if (argc < 2 || (!gpu && strcmp(argv[1], "cpu") != 0) || (gpu && argc != 3)) {
        fprintf(stderr, "usage: %s cpu | gpu <output file>\n", argv[0]);
        return 2;
    }
    if (gpu && (g_out = fopen(argv[2], "wb")) == NULL) {
        perror(argv[2]);
        return 2;
    }
    const int rows = 257, cols = 255;                        /* AVX2 body + ragged tail */
    const int s2[2] = {rows, cols}, s1[1] = {cols}, scol[2] = {rows, 1}, s3[3] = {6, 37, 20};
    NDArray *hx = input(s2, 2, 101, -50, 50), *hy = input(s2, 2, 102, -50, 50), *hrow = input(s1, 1, 103, 0.5f, 4);
    NDArray *hcol = input(scol, 2, 105, 0.5f, 4), *hp = input(s2, 2, 104, 0.25f, 4), *h3 = input(s3, 3, 106, -2, 2);
    NDArray *two = NDArray_CreateFromDoubleScalar(2.5);      /* a PHP float operand: 0-d, on the host (numpower.c:193-229) */
    NDArray *x = placed(hx, gpu), *y = placed(hy, gpu), *row = placed(hrow, gpu), *col = placed(hcol, gpu);
    NDArray *p = placed(hp, gpu), *a3 = placed(h3, gpu), *two_dev = placed(two, gpu);
    int calls = 0;
    for (int k = 0; k < kCount; k++) {
        const int is_pow = strcmp(kPatched[k].name, "Pow_Float") == 0;
        NDArray *lhs = is_pow ? p : x;                       /* positive bases for pow */
        const struct { const char *form; NDArray *a, *b; } forms[] = {
            {"full", lhs, y}, {"row", lhs, row}, {"col", lhs, col}, {"scalar", lhs, two}, {"rscalar", two, lhs},
            {"devscalar", lhs, two_dev}, {"rrow", row, lhs},
        };
        for (size_t f = 0; f < sizeof forms / sizeof forms[0]; f++) {
            if (is_pow && (strcmp(forms[f].form, "full") == 0 || strcmp(forms[f].form, "rscalar") == 0)) continue;
            NDArray *r = kPatched[k].fn(forms[f].a, forms[f].b);
            calls++;
            if (gpu) {
                dump(kPatched[k].name, forms[f].form, r);
                if (r) NDArray_FREE(r);
            } else if (r != NULL) {
                fprintf(stderr, "fast_path_bodies: %s.%s computed something for CPU operands\n", kPatched[k].name, forms[f].form);
                g_failed = 1;
            }
        }
    }
    for (int axis_i = 0; axis_i < 3; axis_i++) {             /* reduce(nda, &axis_i, NDArray_Add_Float | Multiply) */
        static const struct { const char *name; Binary op; } kReduce[] = {{"sum", NDArray_Add_Float}, {"prod", NDArray_Multiply_Float}};
        for (int k = 0; k < 2; k++) {
            char form[16];
            snprintf(form, sizeof form, "axis%d", axis_i);
            NDArray *r = patched_reduce(a3, &axis_i, kReduce[k].op);
            calls++;
            if (gpu) dump(kReduce[k].name, form, r);
            if (r) NDArray_FREE(r);
        }
    }
    /* any other operation keeps the reference's loop, GPU array or not */
    { int axis_i = 0; NDArray *r = patched_reduce(a3, &axis_i, NDArray_Subtract_Float); if (r) NDArray_FREE(r); }
    for (int axis_i = 0; axis_i < 3; axis_i++) {             /* single_reduce(nda, &i_axis, NDArray_Mean_Float): PHP_METHOD(mean) on a GPU array */
        char form[16];
        snprintf(form, sizeof form, "axis%d", axis_i);
        NDArray *r = patched_single_reduce(a3, &axis_i, NDArray_Mean_Float);
        calls++;
        if (gpu) dump("mean", form, r);
        if (r) NDArray_FREE(r);
    }
    { int axis_i = 1; NDArray *r = patched_single_reduce(a3, &axis_i, NDArray_Min); calls++; if (r) NDArray_FREE(r); }   /* min keeps the loop */
    const int expect_fell = gpu ? 2 : calls + 1;
    printf("fast_path_bodies %s: %d calls, %d reached the reference's own code (expected %d)\n", argv[1], calls + 1,
           g_fell_through, expect_fell);
    if (g_fell_through != expect_fell) g_failed = 1;
    if (gpu) {
        NDArray_FREE(x); NDArray_FREE(y); NDArray_FREE(row); NDArray_FREE(col); NDArray_FREE(p); NDArray_FREE(a3); NDArray_FREE(two_dev);
        fclose(g_out);
        if (NDArray_LiveDeviceAllocations() != 0) {
            fprintf(stderr, "fast_path_bodies: %ld device allocations leaked\n", NDArray_LiveDeviceAllocations());
            g_failed = 1;
        }
    }
    NDArray_FREE(hx); NDArray_FREE(hy); NDArray_FREE(hrow); NDArray_FREE(hcol); NDArray_FREE(hp); NDArray_FREE(h3); NDArray_FREE(two);
    return g_failed;
}
'''


# ---- section 2c as a program ------------------------------------------------------------------------------------------
def _edit(prefix: str) -> Edit:
    return next(e for e in EDITS if e.what.startswith(prefix))


LAZY_UNARY = [("exp", "", "cuda_float_exp"), ("log", "", "cuda_float_log"), ("sqrt", "", "cuda_float_sqrt"),
              ("sin", "", "cuda_float_sin"), ("negate", "", "cuda_float_negate"),
              ("round", "1F", "cuda_float_round, (float)precision"), ("clip", "2F", "cuda_float_clip, (float)min, (float)max")]
LAZY_BINARY = [("Add", "ZEND_ADD"), ("Subtract", "ZEND_SUB"), ("Multiply", "ZEND_MUL"), ("Divide", "ZEND_DIV"), ("Pow", "ZEND_POW"),
               ("Mod", "ZEND_MOD")]


def lazy_program_source() -> str:
    """A C99 PROGRAM around the text section 2c inserts (numpower_amd/lib/lazy_bodies): a stand-in for the Zend side — a zval
    that is a number or an object handle, the reference's object table (MAIN_MEM_STACK: add_to_buffer / buffer_get /
    buffer_ndarray_free), ZVAL_TO_NDARRAY, CHECK_INPUT_AND_FREE, RETURN_NDARRAY restated after numpower.c:89-150 and
    src/buffer.c:61-120 — and, inside it, functions with the shape of ndarray_do_operation_ex, PHP_METHOD(add ...) and the unary
    PHP_METHODs whose EDITED statements are the tool's text VERBATIM (as EDITS holds it; the unary call with its function
    name filled in).  Expressions are then evaluated the way PHP does: every operator result is a fresh object, temporaries are
    destroyed as soon as the next operator has consumed them.  Linked with libnumpower_host.so (which carries ext/hip_lazy.c,
    and whose NDArray_FREE holds the NPH_OnFree hook), -Wall -Wextra -Werror.

        lazy_bodies cpu           every appender with CPU operands: each call must reach the stand-in for the reference's own
                                  code, nothing becomes pending, no device is touched (tests/test_apply_with_hip_cpu.py)
        lazy_bodies gpu <file>    the same expressions with ->gpu() operands, chains on and off: launches counted through
                                  np_debug_launch_count, values written to <file> (method_bodies.c's record format; checked
                                  against the oracle by tests/test_gpu_lazy_bodies.py) and compared bit for bit, chain
                                  against eager, by the program itself
    """
    flush, marshal_op = _edit("buffer.c:80-81"), _edit("numpower.c:194-195")
    marshal_bin, marshal_un = _edit("numpower.c:3374-3540"), _edit("numpower.c:1608-3357")
    call_un = _edit("numpower.c:1651-3348")
    o = ["/* generated by tools/apply_with_hip.py: lazy_program_source() — do not edit */",
         "#define _POSIX_C_SOURCE 200809L",
         "#include <assert.h>", "#include <stdint.h>", "#include <stdio.h>", "#include <stdlib.h>", "#include <string.h>", "",
         "#define HAVE_NP_HIP 1", "#define HAVE_CUBLAS 1",
         '#include "numpower_host.h"', '#include "hip_fast.h"', '#include "hip_lazy.h"', '#include "hip_math.h"', '#include "np_hip_debug.h"', "",
         _LAZY_PROGRAM_ZEND,
         "NDArray* buffer_get(int uuid) {", "    assert(MAIN_MEM_STACK.buffer[uuid] != NULL);",
         "#ifdef HAVE_NP_HIP", _indent(flush.new, "    "), "#endif", "    return MAIN_MEM_STACK.buffer[uuid];", "}", "",
         _LAZY_PROGRAM_MARSHAL, ""]
    # the L2 functions as section 2b leaves them: GPU operands -> NPH_Binary_Float, CPU operands -> "the reference's body"
    for name, _ in LAZY_BINARY:
        e = next(x for n, x in FAST_BINARY if n == "NDArray_%s_Float" % name)
        o += ["static NDArray *patched_NDArray_%s_Float(NDArray *a, NDArray *b) {" % name, "#ifdef HAVE_NP_HIP", _indent(e.new, "    "),
              "#endif", "    g_reference_bodies++;   /* arithmetics.c: the scalar expand, the broadcast, the AVX2 loop */", "    return NULL;", "}",
              "#define NDArray_%s_Float patched_NDArray_%s_Float" % (name, name), ""]
    # ndarray_do_operation_ex (numpower.c:193-229)
    o += ["static int patched_do_operation_ex(int opcode, zval *result, zval *op1, zval *op2) {",
          "#ifdef HAVE_NP_HIP", _indent(marshal_op.new, "    "), "#endif",
          "    if (nda == NULL || ndb == NULL) {", "        return FAILURE;", "    }", "    NDArray *rtn = NULL;", "    switch (opcode) {"]
    for name, zend in LAZY_BINARY:
        e = _edit("numpower.c:200-218,3384-3550 `rtn = NDArray_%s_Float" % name)
        o += ["    case %s:" % zend, "#ifdef HAVE_NP_HIP", _indent(e.new, "        "), "#endif", "        break;"]
    o += ["    default:", "        return FAILURE;", "    }", "    CHECK_INPUT_AND_FREE(op1, nda);", "    CHECK_INPUT_AND_FREE(op2, ndb);",
          "    RETURN_NDARRAY(rtn, result);", "    return rtn != NULL ? SUCCESS : FAILURE;", "}", ""]
    # PHP_METHOD(NDArray, add) ... (numpower.c:3364-3553): the static form, one function per operator
    for name, _ in LAZY_BINARY:
        e = _edit("numpower.c:200-218,3384-3550 `rtn = NDArray_%s_Float" % name)
        o += ["static void patched_method_%s(zval *a, zval *b, zval *return_value) {" % name.lower(), "    NDArray *rtn = NULL;",
              "#ifdef HAVE_NP_HIP", _indent(marshal_bin.new, "    "), "#endif",
              "    if (nda == NULL) {", "        return;", "    }", "    if (ndb == NULL) {", "        CHECK_INPUT_AND_FREE(a, nda);", "        return;", "    }",
              "#ifdef HAVE_NP_HIP", _indent(e.new, "    "), "#endif",
              "    CHECK_INPUT_AND_FREE(a, nda);", "    CHECK_INPUT_AND_FREE(b, ndb);", "    RETURN_NDARRAY(rtn, return_value);", "}", ""]
    # the unary PHP_METHODs (numpower.c:1608-3357)
    rx = re.compile(call_un.anchor, re.M)
    for name, sfx, rest in LAZY_UNARY:
        ref_line = "        rtn = NDArrayMathGPU_ElementWise%s(nda, %s);" % (sfx, rest)
        m = rx.search(ref_line)
        assert m, ref_line
        params = {"": "", "1F": ", long precision", "2F": ", double min, double max"}[sfx]
        o += ["static void patched_method_%s(zval *array%s, zval *return_value) {" % (name, params), "    NDArray *rtn = NULL;",
              "#ifdef HAVE_NP_HIP", _indent(marshal_un.new, "    "), "#endif",
              "    if (nda == NULL) {", "        return;", "    }",
              "    if (NDArray_DEVICE(nda) == NDARRAY_DEVICE_CPU) {",
              "        g_reference_bodies++;   /* rtn = NDArray_Map(nda, float_%s); */" % name, "    } else {", "#ifdef HAVE_CUBLAS",
              _indent(m.expand(call_un.new), "        "), "#endif", "    }",
              "    RETURN_NDARRAY(rtn, return_value);", "}", ""]
    # the full reductions (numpower.c:4620-4751, 2642-2688): the no-axis branch of each method
    marshal_red, marshal_mean = _edit("numpower.c:4630-4738"), _edit("numpower.c:2653")
    mean_call = _edit("numpower.c:2660,2675")
    for fn in ("NDArray_Sum_Float", "NDArray_Float_Prod", "NDArray_Min", "NDArray_Max"):
        o += ["static float patched_%s(NDArray *a) {" % fn,
              "    if (NDArray_DEVICE(a) == NDARRAY_DEVICE_GPU) return %s(a);   /* the device branch of the reference function (section 2a) */" % fn,
              "    g_reference_bodies++;                                          /* its CPU loop */", "    return 0.0f;", "}",
              "#define %s patched_%s" % (fn, fn), ""]
    for name, prefix in (("sum", "numpower.c:4638"), ("min", "numpower.c:4673"), ("max", "numpower.c:4712"), ("prod", "numpower.c:4744")):
        e = _edit(prefix)
        o += ["static double patched_method_%s(zval *a) {" % name] + (["    double value;"] if name != "sum" else []) + [
              "#ifdef HAVE_NP_HIP", _indent(marshal_red.new, "    "), "#endif",
              "    if (nda == NULL) {", "        return -1.0;", "    }",
              "#ifdef HAVE_NP_HIP", _indent(e.new, "    "), "#endif",
              "    CHECK_INPUT_AND_FREE(a, nda);", "    return value;", "}", ""]
    # reduce() (src/ndarray.c:523-578) around its edited statement, and PHP_METHOD(sum)'s axis branch in front of it
    reduce_edit = _edit("ndarray.c:570 reduce()")
    o += ["static void _reduce(int current_axis, int rtn_init, int *axis, NDArray *target, NDArray *rtn, NDArray *(*operation)(NDArray *, NDArray *)) {",
          "    (void) current_axis; (void) rtn_init; (void) axis; (void) target; (void) rtn; (void) operation;",
          "    g_reference_bodies++;   /* the reference's slice-by-slice loop */", "}",
          "static NDArray *patched_reduce(NDArray *array, int *axis, NDArray *(*operation)(NDArray *, NDArray *)) {",
          "    int out_shape[8], j = 0;",
          "    for (int i = 0; i < NDArray_NDIM(array); i++) if (i != *axis) out_shape[j++] = NDArray_SHAPE(array)[i];",
          '    NDArray *rtn = NDArray_Zeros(out_shape, j, "float32", NDArray_DEVICE(array));',
          "#ifdef HAVE_NP_HIP", _indent(reduce_edit.new, "    "), "#endif", "    return rtn;", "}",
          "static zval patched_method_sum_axis(zval *a, int axis_i) {   /* rtn = reduce(nda, &axis_i, NDArray_Add_Float); numpower.c:4636 */",
          "    zval return_value = {IS_UNDEF, 0.0, 0};",
          "#ifdef HAVE_NP_HIP", _indent(marshal_red.new, "    "), "#endif",
          "    if (nda == NULL) {", "        return return_value;", "    }",
          "    NDArray *rtn = patched_reduce(nda, &axis_i, NDArray_Add_Float);",
          "    CHECK_INPUT_AND_FREE(a, nda);", "    RETURN_NDARRAY(rtn, &return_value);", "    return return_value;", "}", ""]
    o += ["#define RETURN_DOUBLE(d) return (d)",
          "static double patched_method_mean(zval *array) {",
          "#ifdef HAVE_NP_HIP", _indent(marshal_mean.new, "    "), "#endif",
          "    if (nda == NULL) {", "        return -1.0;", "    }",
          "#ifdef HAVE_NP_HIP", _indent(mean_call.new, "    "), "#endif", "}", ""]
    rinit = _edit("numpower.c:5250 PHP_RINIT")
    o.append(_LAZY_PROGRAM_MAIN.replace("    @RINIT@", "    nph_marshal_lazy = 7;   /* as a request that bailed out inside an appender's lookup leaves it */\n" +
                                        _indent(rinit.new, "    ")))
    return "\n".join(o)


_LAZY_PROGRAM_ZEND = r"""/* ---- the Zend side, restated: a zval is a PHP float or an NDArray object (its handle = the `id` property, numpower.c:84-87) ---- */
enum { IS_UNDEF = 0, IS_DOUBLE = 5, IS_OBJECT = 8 };
enum { SUCCESS = 0, FAILURE = -1 };
enum { ZEND_ADD = 1, ZEND_SUB = 2, ZEND_MUL = 3, ZEND_DIV = 4, ZEND_MOD = 5, ZEND_POW = 12 };
typedef struct zval { int type; double dval; int handle; } zval;
#define Z_TYPE_P(z) ((z)->type)

static int g_reference_bodies;   /* calls that reached "the reference's own code" (CPU operands) */

/* src/buffer.h:9-16, src/buffer.c:91-120 */
struct MemoryStack { NDArray **buffer; int bufferSize; int numElements; int lastFreed; };
static struct MemoryStack MAIN_MEM_STACK = {NULL, 0, 0, -1};

static void add_to_buffer(NDArray *ndarray) {
    if (MAIN_MEM_STACK.lastFreed > -1) {
        ndarray->uuid = MAIN_MEM_STACK.lastFreed;
        MAIN_MEM_STACK.buffer[MAIN_MEM_STACK.lastFreed] = ndarray;
        MAIN_MEM_STACK.lastFreed = -1;
        return;
    }
    if (MAIN_MEM_STACK.numElements >= MAIN_MEM_STACK.bufferSize) {
        const int size = MAIN_MEM_STACK.bufferSize == 0 ? 8 : MAIN_MEM_STACK.bufferSize * 2;
        MAIN_MEM_STACK.buffer = (NDArray **) realloc(MAIN_MEM_STACK.buffer, (size_t) size * sizeof(NDArray *));
        MAIN_MEM_STACK.bufferSize = size;
    }
    ndarray->uuid = MAIN_MEM_STACK.numElements;
    MAIN_MEM_STACK.buffer[MAIN_MEM_STACK.numElements++] = ndarray;
}

/* ndarray_destructor -> buffer_ndarray_free (numpower.c:292-298, src/buffer.c:61-75): the table is read WITHOUT buffer_get */
static void buffer_ndarray_free(int uuid) {
    if (MAIN_MEM_STACK.lastFreed == -1) MAIN_MEM_STACK.lastFreed = uuid;
    if (MAIN_MEM_STACK.buffer[uuid] != NULL) {
        NDArray_FREE(MAIN_MEM_STACK.buffer[uuid]);
        MAIN_MEM_STACK.buffer[uuid] = NULL;
    }
}
"""

_LAZY_PROGRAM_MARSHAL = r"""/* numpower.c:89-150 */
static NDArray *ZVAL_TO_NDARRAY(zval *obj) {
    if (Z_TYPE_P(obj) == IS_DOUBLE) return NDArray_CreateFromDoubleScalar(obj->dval);
    if (Z_TYPE_P(obj) == IS_OBJECT) return buffer_get(obj->handle);
    return NULL;
}
static void CHECK_INPUT_AND_FREE(zval *a, NDArray *nda) {
    if (nda == NULL || a == NULL) return;
    if (Z_TYPE_P(a) == IS_DOUBLE) NDArray_FREE(nda);
}
static void RETURN_NDARRAY(NDArray *array, zval *return_value) {
    return_value->type = IS_UNDEF;
    if (array == NULL) return;                       /* RETURN_THROWS() */
    add_to_buffer(array);
    return_value->type = IS_OBJECT;
    return_value->handle = array->uuid;
}
/* PHP lets go of a value (a temporary after the operator that consumed it, unset($x), a variable overwritten) */
static void zval_dtor(zval *z) {
    if (Z_TYPE_P(z) == IS_OBJECT) buffer_ndarray_free(z->handle);
    z->type = IS_UNDEF;
}
static NDArray *buffer_peek(zval *z) { return Z_TYPE_P(z) == IS_OBJECT ? MAIN_MEM_STACK.buffer[z->handle] : NULL; }   /* tests only: no flush */
static zval number(double v) { zval z = {IS_DOUBLE, v, 0}; return z; }
static zval object_of(NDArray *a) { zval z = {IS_UNDEF, 0.0, 0}; RETURN_NDARRAY(a, &z); return z; }"""

_LAZY_PROGRAM_MAIN = r"""/* ---- consumers: methods that are NOT appenders marshal through ZVAL_TO_NDARRAY -> buffer_get, unedited ---- */
static NDArray *method_cpu(zval *obj) { return NDArray_ToCPU(ZVAL_TO_NDARRAY(obj)); }               /* numpower.c:541-559 */
static float method_sum(zval *obj) { return NDArray_Sum_Float(ZVAL_TO_NDARRAY(obj)); }               /* numpower.c:4620-4640 */
static void method_fill(zval *obj, float v) { (void) NDArray_Fill(ZVAL_TO_NDARRAY(obj), v); }        /* numpower.c:4788-4800: writes in place */
static zval method_slice0(zval *obj, int i) { return object_of(NDArray_LeadingSlice(ZVAL_TO_NDARRAY(obj), i)); }   /* $a[i]: a view */
static zval method_equal(zval *a, zval *b) {                                                         /* numpower.c:1042-1068: a comparison is no appender */
    NDArray *nda = ZVAL_TO_NDARRAY(a), *ndb = ZVAL_TO_NDARRAY(b);
    return object_of(NDArray_Equal(nda, ndb));
}

static FILE *g_out;
static int g_failed;
#define CHECK(cond, ...) do { if (!(cond)) { fprintf(stderr, "lazy_bodies: " __VA_ARGS__); fprintf(stderr, "\n"); g_failed = 1; } } while (0)

static unsigned long long launches(void) {
    unsigned long long n = 0;
    np_debug_launch_count(&n);
    return n;
}

/* method_bodies.c's record: 32-byte name, int32 ndim, int32 dims[4], the floats; returns the host copy (caller frees) */
static NDArray *dump(const char *label_text, zval *value) {
    char label[32];
    int32_t head[5] = {0, 1, 1, 1, 1};
    if (Z_TYPE_P(value) != IS_OBJECT) {
        fprintf(stderr, "lazy_bodies: %s has no value: %s\n", label_text, numpower_host_last_error());
        g_failed = 1;
        return NULL;
    }
    NDArray *host = method_cpu(value);
    if (host == NULL) {
        fprintf(stderr, "lazy_bodies: cpu() of %s failed: %s\n", label_text, numpower_host_last_error());
        g_failed = 1;
        return NULL;
    }
    memset(label, 0, sizeof label);
    snprintf(label, sizeof label, "%s", label_text);
    head[0] = NDArray_NDIM(host);
    for (int i = 0; i < NDArray_NDIM(host) && i < 4; i++) head[1 + i] = NDArray_SHAPE(host)[i];
    fwrite(label, 1, sizeof label, g_out);
    fwrite(head, sizeof(int32_t), 5, g_out);
    fwrite(NDArray_FDATA(host), sizeof(float), (size_t) NDArray_NUMELEMENTS(host), g_out);
    return host;
}

/* x[i] = lo + (hi - lo) * frac(i * 0.6180339887 + seed * 0.37): what tests/test_gpu_method_bodies.py::c_input rebuilds */
static NDArray *input(const int *shape, int ndim, int seed, float lo, float hi) {
    long n = 1;
    for (int i = 0; i < ndim; i++) n *= shape[i];
    float *host = (float *) malloc(sizeof(float) * (size_t) n);
    for (long i = 0; i < n; i++) {
        double t = (double) i * 0.6180339887 + (double) seed * 0.37;
        t -= (double) (long) t;
        host[i] = (float) ((double) lo + ((double) hi - (double) lo) * t);
    }
    NDArray *cpu = NDArray_FromHostBuffer(host, shape, ndim);
    free(host);
    return cpu;
}
static zval placed(const int *shape, int ndim, int seed, float lo, float hi, int on_gpu) {   /* nd::array(...)[->gpu()] */
    NDArray *host = input(shape, ndim, seed, lo, hi);
    if (!on_gpu) return object_of(host);
    NDArray *dev = NDArray_ToGPU(host);
    NDArray_FREE(host);
    if (dev == NULL) {
        fprintf(stderr, "lazy_bodies: gpu() failed: %s\n", numpower_host_last_error());
        exit(1);
    }
    return object_of(dev);
}

/* `$r = <expression>` evaluated as PHP evaluates it: one object per operator, temporaries dropped once consumed */
typedef struct Env { zval x, y, p, row, col, wide, w; } Env;
typedef zval (*Expr)(Env *);

static zval op2(int opcode, zval a, zval b, int drop_a, int drop_b) {   /* $a (op) $b through the do_operation handler */
    zval r = {IS_UNDEF, 0.0, 0};
    (void) patched_do_operation_ex(opcode, &r, &a, &b);
    if (drop_a) zval_dtor(&a);
    if (drop_b) zval_dtor(&b);
    return r;
}
#define UN(name, v, drop) un_##name(v, drop)
#define DEFINE_UN(name)                                                       \
    static zval un_##name(zval a, int drop) {                                 \
        zval r = {IS_UNDEF, 0.0, 0};                                          \
        patched_method_##name(&a, &r);                                        \
        if (drop) zval_dtor(&a);                                              \
        return r;                                                             \
    }
DEFINE_UN(exp) DEFINE_UN(log) DEFINE_UN(sqrt) DEFINE_UN(sin) DEFINE_UN(negate)

/* nd::exp($x) * $y + 2 */
static zval e_exp_mul_add(Env *e) { return op2(ZEND_ADD, op2(ZEND_MUL, UN(exp, e->x, 0), e->y, 1, 0), number(2.0), 1, 0); }
/* 2.5 - nd::sqrt($p): the number comes first (swap) */
static zval e_rscalar(Env *e) { return op2(ZEND_SUB, number(2.5), UN(sqrt, e->p, 0), 0, 1); }
/* $y / nd::exp($x): the array comes first, the chain is the second operand */
static zval e_rdiv(Env *e) { return op2(ZEND_DIV, e->y, UN(exp, e->x, 0), 0, 1); }
/* (nd::sin($x) * $row) / $col: a row and a column operand of the same 2-D view */
static zval e_bcast(Env *e) { return op2(ZEND_DIV, op2(ZEND_MUL, UN(sin, e->x, 0), e->row, 1, 0), e->col, 1, 0); }
/* ($x % $y) * $y - $x: the AVX-body quirks of mod and multiply travel with the chain */
static zval e_quirks(Env *e) { return op2(ZEND_SUB, op2(ZEND_MUL, op2(ZEND_MOD, e->x, e->y, 0, 0), e->y, 1, 0), e->x, 1, 0); }
/* $row * -$x: the smaller operand first */
static zval e_rrow(Env *e) { return op2(ZEND_MUL, e->row, UN(negate, e->x, 0), 0, 1); }
/* nd::exp($x) * nd::log($p): two pending operands — the second one is computed and joins as an array */
static zval e_two_pending(Env *e) { return op2(ZEND_MUL, UN(exp, e->x, 0), UN(log, e->p, 0), 1, 1); }
/* $p ** 2 + nd::round($x, 1) through the STATIC methods (PHP_METHOD(pow), PHP_METHOD(add)) and a 1F driver: the square and the
 * add are one chain that starts from $p (`** 2` with a PHP number is x * x stand-alone and inside a chain alike); the pending
 * round is computed when the add consumes it */
static zval e_static(Env *e) {
    zval two = number(2.0), t1 = {IS_UNDEF, 0.0, 0}, t2 = {IS_UNDEF, 0.0, 0}, r = {IS_UNDEF, 0.0, 0};
    patched_method_pow(&e->p, &two, &t1);
    patched_method_round(&e->x, 1, &t2);
    patched_method_add(&t1, &t2, &r);
    zval_dtor(&t1);
    zval_dtor(&t2);
    return r;
}
/* nd::mod(nd::subtract(nd::multiply($x, $y), $row), nd::divide($y, 2.0)): the other four static methods */
static zval e_static2(Env *e) {
    zval half = number(2.0), t = {IS_UNDEF, 0.0, 0}, u = {IS_UNDEF, 0.0, 0}, v = {IS_UNDEF, 0.0, 0}, r = {IS_UNDEF, 0.0, 0};
    patched_method_multiply(&e->x, &e->y, &t);
    patched_method_subtract(&t, &e->row, &u);
    zval_dtor(&t);
    patched_method_divide(&e->y, &half, &v);
    patched_method_mod(&u, &v, &r);
    zval_dtor(&u);
    zval_dtor(&v);
    return r;
}
/* nd::clip(nd::exp($x), 0.5, 3.0) - a 2F driver on a pending operand */
static zval e_clip(Env *e) {
    zval t = UN(exp, e->x, 0), r = {IS_UNDEF, 0.0, 0};
    patched_method_clip(&t, 0.5, 3.0, &r);
    zval_dtor(&t);
    return r;
}
/* fifteen steps: the chain is full after twelve, its value starts a second chain */
static zval e_long(Env *e) {
    zval v = UN(negate, e->x, 0);
    for (int i = 0; i < 7; i++) {
        v = op2(ZEND_ADD, v, number(0.25 * (i + 1)), 1, 0);
        v = UN(negate, v, 1);
    }
    return v;
}
/* nd::exp($row) * $wide: the second operand is LARGER than the pending one — the chain cannot grow; exp is computed and the
 * multiply is the eager launch (NPH_Binary_Float), the result takes $wide's shape */
static zval e_grow(Env *e) { return op2(ZEND_MUL, UN(exp, e->row, 0), e->wide, 1, 0); }
/* $row + $wide: the smaller operand first is still ONE step of a chain that starts from $wide */
static zval e_small_first(Env *e) { return op2(ZEND_ADD, e->row, e->wide, 0, 0); }
/* $x + $y: one step — the flush IS the stand-alone launch */
static zval e_single(Env *e) { return op2(ZEND_ADD, e->x, e->y, 0, 0); }
/* ($x - $y) ** 2: the squared difference of a loss, one launch */
static zval e_sq_diff(Env *e) { return op2(ZEND_POW, op2(ZEND_SUB, e->x, e->y, 0, 0), number(2.0), 1, 0); }
/* ($p ** $y) * 0.5 and nd::sqrt($p) ** 1.5: pow with an array and with a number other than 2 are steps like any other */
static zval e_pow_array(Env *e) { return op2(ZEND_MUL, op2(ZEND_POW, e->p, e->y, 0, 0), number(0.5), 1, 0); }
static zval e_pow_number(Env *e) { return op2(ZEND_POW, UN(sqrt, e->p, 0), number(1.5), 1, 0); }
/* 2 ** $x: the number is the BASE (swap) - the general pow, not a square */
static zval e_pow_base2(Env *e) { return op2(ZEND_ADD, op2(ZEND_POW, number(2.0), e->x, 0, 0), number(1.0), 1, 0); }

/* ---- stress: a seeded random PROGRAM over a pool of 20 (or 96) PHP variables — operators (arrays, numbers, pending values, views), unary
 * methods, assignments over live variables, unset(), in-place writes (directly and through a view), reductions, reads — evaluated
 * through the inserted text exactly as the expressions above.  Run twice, chains on and off: every value read on the way and every
 * variable alive at the end must agree bit for bit (NaNs as NaNs), nothing may stay pending, no device allocation may leak. ---- */
enum { kPoolMax = 96 };      /* more variables than the table of pending chains has rows (NPH_MAX_PENDING = 64) */
typedef struct Rng { unsigned long long s; } Rng;
static unsigned rnd(Rng *r, unsigned n) {
    r->s = r->s * 6364136223846793005ULL + 1442695040888963407ULL;
    return (unsigned) ((r->s >> 33) % n);
}
static unsigned long long fnv(unsigned long long h, const void *p, size_t n) {
    const unsigned char *q = (const unsigned char *) p;
    for (size_t i = 0; i < n; i++) h = (h ^ q[i]) * 1099511628211ULL;
    return h;
}
static unsigned long long digest_value(unsigned long long h, zval *z) {
    if (Z_TYPE_P(z) != IS_OBJECT) return fnv(h, "none", 4);
    NDArray *host = method_cpu(z);
    if (host == NULL) {
        numpower_host_clear_error();
        return fnv(h, "fail", 4);
    }
    const int ndim = NDArray_NDIM(host);
    h = fnv(h, &ndim, sizeof ndim);
    h = fnv(h, NDArray_SHAPE(host), sizeof(int) * (size_t) ndim);
    const float *v = NDArray_FDATA(host);
    for (long i = 0; i < (long) NDArray_NUMELEMENTS(host); i++) {
        uint32_t bits;
        memcpy(&bits, v + i, 4);
        if (v[i] != v[i]) bits = 0x7fc00000u;              /* a NaN is a NaN */
        h = fnv(h, &bits, 4);
    }
    NDArray_FREE(host);
    return h;
}
static zval fresh_leaf(Rng *r, int rows, int cols, int serial) {
    const int s2[2] = {rows, cols}, s1[1] = {cols}, sc[2] = {rows, 1};
    switch (rnd(r, 5)) {
        case 0: return placed(s1, 1, 400 + serial, 0.5f, 2.0f, 1);
        case 1: return placed(sc, 2, 400 + serial, 0.5f, 2.0f, 1);
        default: return placed(s2, 2, 400 + serial, -2.0f, 2.0f, 1);
    }
}
static int g_stress_most_pending;
static unsigned long long stress(int lazy, unsigned seed, int n_steps, int rows, int cols, int kPool, unsigned long *evaluated) {
    static const int kOps[6] = {ZEND_ADD, ZEND_SUB, ZEND_MUL, ZEND_DIV, ZEND_MOD, ZEND_POW};
    Rng r = {seed * 2654435761ULL + 12345ULL};
    zval pool[kPoolMax];
    unsigned long long h = 1469598103934665603ULL;
    int serial = 0;
    NPH_SetLazy(lazy);
    for (int i = 0; i < kPool; i++) {
        pool[i].type = IS_UNDEF;
        if (i < 8 || i < kPool / 3) pool[i] = fresh_leaf(&r, rows, cols, serial++);
    }
    for (int step = 0; step < n_steps; step++) {
        const unsigned what = rnd(&r, 100);
        const int a = (int) rnd(&r, (unsigned) kPool), b = (int) rnd(&r, (unsigned) kPool), dst = (int) rnd(&r, (unsigned) kPool);
        zval result = {IS_UNDEF, 0.0, 0};
        if (Z_TYPE_P(&pool[a]) != IS_OBJECT) {             /* an unset variable: give it a new array */
            pool[a] = fresh_leaf(&r, rows, cols, serial++);
            continue;
        }
        if (what < 45) {                                    /* $dst = $a (op) $b | number */
            const int op = kOps[rnd(&r, 6)];
            zval rhs = (Z_TYPE_P(&pool[b]) != IS_OBJECT || rnd(&r, 100) < 30) ? number(op == ZEND_POW ? 2.0 : 0.5 + 0.25 * (double) rnd(&r, 8)) : pool[b];
            if (op == ZEND_POW && Z_TYPE_P(&rhs) == IS_OBJECT) rhs = number(2.0);
            if (rnd(&r, 100) < 25 && op != ZEND_POW) (void) patched_do_operation_ex(op, &result, &rhs, &pool[a]);
            else (void) patched_do_operation_ex(op, &result, &pool[a], &rhs);
        } else if (what < 70) {                             /* $dst = nd::f($a) */
            switch (rnd(&r, 6)) {
                case 0: patched_method_sin(&pool[a], &result); break;
                case 1: patched_method_negate(&pool[a], &result); break;
                case 2: patched_method_sqrt(&pool[a], &result); break;
                case 3: patched_method_exp(&pool[a], &result); break;
                case 4: patched_method_clip(&pool[a], -4.0, 4.0, &result); break;
                default: patched_method_round(&pool[a], 1, &result); break;
            }
        } else if (what < 78) {                             /* unset($a) */
            zval_dtor(&pool[a]);
            continue;
        } else if (what < 85) {                             /* $a->fill(c): whoever reads $a's buffer is computed first */
            method_fill(&pool[a], 0.25f * (float) rnd(&r, 9));
            continue;
        } else if (what < 90) {                             /* $dst = $a[0] (a view), now and then written through */
            NDArray *peek = buffer_peek(&pool[a]);
            if (peek == NULL || NDArray_NDIM(peek) != 2) continue;
            result = method_slice0(&pool[a], 0);
            if (Z_TYPE_P(&result) == IS_OBJECT && rnd(&r, 100) < 50) method_fill(&result, 1.5f);
        } else if (what < 95) {                             /* nd::max($a) / nd::min($a): bit-identical inside a chain and over stored values */
            const double v = rnd(&r, 2) ? patched_method_max(&pool[a]) : patched_method_min(&pool[a]);
            float f = (float) v;
            if (f != f) f = 0.0f;
            h = fnv(h, &f, sizeof f);
            (*evaluated)++;
            continue;
        } else {                                            /* print_r($a) */
            h = digest_value(h, &pool[a]);
            (*evaluated)++;
            continue;
        }
        if (Z_TYPE_P(&result) != IS_OBJECT) {               /* the operator threw (shapes that do not broadcast ...): $dst keeps its value */
            numpower_host_clear_error();
            h = fnv(h, "threw", 5);
            continue;
        }
        zval_dtor(&pool[dst]);                              /* assignment: the old value of $dst goes once the new one exists */
        pool[dst] = result;
        if (NPH_PendingCount() > g_stress_most_pending) g_stress_most_pending = NPH_PendingCount();
    }
    for (int i = 0; i < kPool; i++) {
        h = digest_value(h, &pool[i]);
        zval_dtor(&pool[i]);
    }
    NPH_SetLazy(1);
    return h;
}

static const struct { const char *name; Expr fn; int steps; int lazy_launches; } kExpr[] = {
    {"exp_mul_add", e_exp_mul_add, 3, 1}, {"rscalar", e_rscalar, 2, 1}, {"rdiv", e_rdiv, 2, 1}, {"bcast", e_bcast, 3, 1},
    {"quirks", e_quirks, 3, 1}, {"rrow", e_rrow, 2, 1}, {"two_pending", e_two_pending, 3, 2}, {"static", e_static, 3, 2}, {"static2", e_static2, 4, 2},
    {"clip", e_clip, 2, 1}, {"long", e_long, 15, 2}, {"grow", e_grow, 2, 2}, {"small_first", e_small_first, 1, 1}, {"single", e_single, 1, 1},
    {"sq_diff", e_sq_diff, 2, 1}, {"pow_array", e_pow_array, 2, 1}, {"pow_number", e_pow_number, 2, 1}, {"pow_base2", e_pow_base2, 2, 1},
};
enum { kExprCount = (int) (sizeof kExpr / sizeof kExpr[0]) };

int main(int argc, char **argv) {
    const int gpu = argc >= 2 && strcmp(argv[1], "gpu") == 0;
    if (argc < 2 || (!gpu && strcmp(argv[1], "cpu") != 0) || (gpu && argc != 3)) {
        fprintf(stderr, "usage: %s cpu | gpu <output file>\n", argv[0]);
        return 2;
    }
    if (gpu && (g_out = fopen(argv[2], "wb")) == NULL) {
        perror(argv[2]);
        return 2;
    }
    @RINIT@
    const int rows = 257, cols = 255;                        /* AVX2 body + ragged tail */
    const int s2[2] = {rows, cols}, s1[1] = {cols}, scol[2] = {rows, 1};
    Env e;
    e.x = placed(s2, 2, 201, -3, 3, gpu);
    e.y = placed(s2, 2, 202, 0.5f, 4, gpu);
    e.p = placed(s2, 2, 203, 0.25f, 4, gpu);
    e.row = placed(s1, 1, 204, -2, 2, gpu);
    e.col = placed(scol, 2, 205, 0.5f, 2, gpu);
    e.wide = placed(s2, 2, 206, -1, 1, gpu);
    e.w = placed(s2, 2, 207, 0.9997f, 1.0003f, gpu);       /* factors of a product that stays near 1 */
    NPH_LazyStats st0, st1;

    if (!gpu) {
        /* BASELINE config 1: CPU operands.  Every appender must hand the call to the reference's own code; nothing pends. */
        int calls = 0;
        for (int k = 0; k < kExprCount; k++) {
            zval r = kExpr[k].fn(&e);
            calls += kExpr[k].steps;
            CHECK(Z_TYPE_P(&r) == IS_UNDEF, "%s computed something for CPU operands", kExpr[k].name);
            CHECK(NPH_PendingCount() == 0, "%s left a pending chain for CPU operands", kExpr[k].name);
        }
        const int before = g_reference_bodies;
        (void) patched_method_sum(&e.x); (void) patched_method_min(&e.x); (void) patched_method_max(&e.x); (void) patched_method_prod(&e.x);
        (void) patched_method_mean(&e.x);
        CHECK(g_reference_bodies - before == 5 && NPH_PendingCount() == 0, "the reductions of a CPU array: %d reached the reference's code", g_reference_bodies - before);
        /* (an expression whose first step yields nothing stops there, as PHP would on the exception: count what ran) */
        printf("lazy_bodies cpu: %d expressions, %d reached the reference's own code, 0 pending\n", kExprCount, g_reference_bodies);
        CHECK(g_reference_bodies >= kExprCount, "only %d calls reached the reference's code", g_reference_bodies);
        (void) calls;
        return g_failed;
    }

    /* ---- 1. every expression: chains on (launches counted) against chains off, bit for bit ---- */
    for (int k = 0; k < kExprCount; k++) {
        char label[32];
        NPH_SetLazy(1);
        NPH_GetLazyStats(&st0);
        const unsigned long long l0 = launches();
        zval r = kExpr[k].fn(&e);
        const unsigned long long l_build = launches() - l0;
        snprintf(label, sizeof label, "lazy.%s", kExpr[k].name);
        NDArray *lazy = dump(label, &r);
        const unsigned long long l_lazy = launches() - l0;
        NPH_GetLazyStats(&st1);
        zval_dtor(&r);
        CHECK(NPH_PendingCount() == 0, "%s: %d chains still pending after its value was read", kExpr[k].name, NPH_PendingCount());

        NPH_SetLazy(0);
        const unsigned long long l1 = launches();
        zval q = kExpr[k].fn(&e);
        snprintf(label, sizeof label, "eager.%s", kExpr[k].name);
        NDArray *eager = dump(label, &q);
        const unsigned long long l_eager = launches() - l1;
        zval_dtor(&q);
        NPH_SetLazy(1);

        printf("%-12s %2d steps: %llu launch(es) with chains (%llu before the value was asked for), %llu without; chains flushed %lu, steps in them %lu, "
               "discarded %lu, eager steps %lu\n", kExpr[k].name, kExpr[k].steps, l_lazy, l_build, l_eager,
               st1.flushed_chains - st0.flushed_chains, st1.flushed_steps - st0.flushed_steps,
               st1.discarded_chains - st0.discarded_chains, st1.eager_steps - st0.eager_steps);
        CHECK(l_lazy == (unsigned long long) kExpr[k].lazy_launches, "%s took %llu launches with chains, expected %d", kExpr[k].name, l_lazy,
              kExpr[k].lazy_launches);
        CHECK(l_eager == (unsigned long long) kExpr[k].steps, "%s took %llu launches without chains, expected %d", kExpr[k].name, l_eager,
              kExpr[k].steps);
        if (lazy != NULL && eager != NULL) {
            CHECK(NDArray_NUMELEMENTS(lazy) == NDArray_NUMELEMENTS(eager) && NDArray_NDIM(lazy) == NDArray_NDIM(eager) &&
                  memcmp(NDArray_FDATA(lazy), NDArray_FDATA(eager), sizeof(float) * (size_t) NDArray_NUMELEMENTS(lazy)) == 0,
                  "%s: the chain's values differ from the op-by-op values", kExpr[k].name);
        }
        if (lazy) NDArray_FREE(lazy);
        if (eager) NDArray_FREE(eager);
    }

    /* ---- 2. a value nobody asks for costs nothing ---- */
    {
        NPH_GetLazyStats(&st0);
        const unsigned long long l0 = launches();
        zval t = op2(ZEND_MUL, UN(exp, e.x, 0), e.y, 1, 0);
        CHECK(NPH_PendingCount() == 1, "expected one pending chain, have %d", NPH_PendingCount());
        zval_dtor(&t);
        NPH_GetLazyStats(&st1);
        CHECK(launches() == l0 && NPH_PendingCount() == 0 && st1.discarded_chains - st0.discarded_chains == 2,
              "an unused expression launched %llu kernel(s), %d chains pending", launches() - l0, NPH_PendingCount());
        printf("unused value: 0 launches, 2 chains discarded\n");
    }
    /* ---- 3. writes to an input: the chains that read it are computed first (fill, and fill through a view) ---- */
    {
        const int s[2] = {64, 100};
        zval a = placed(s, 2, 301, -1, 1, 1), b = placed(s, 2, 301, -1, 1, 1);
        zval c = op2(ZEND_ADD, a, number(1.0), 0, 0);         /* $c = $a + 1;      pending */
        method_fill(&a, 0.0f);                                /* $a->fill(0);      $c must have been computed from the OLD $a */
        zval view = method_slice0(&b, 3);                     /* $v = $b[3];       a view of $b */
        zval d = op2(ZEND_MUL, b, number(2.0), 0, 0);         /* $d = $b * 2;      pending, reads $b */
        method_fill(&view, 9.0f);                             /* $v->fill(9);      writes $b's buffer through the view */
        NDArray *hc = dump("write.c", &c), *hd = dump("write.d", &d), *ha = dump("write.a", &a), *hb = dump("write.b", &b);
        if (hc) NDArray_FREE(hc);
        if (hd) NDArray_FREE(hd);
        if (ha) NDArray_FREE(ha);
        if (hb) NDArray_FREE(hb);
        zval_dtor(&view); zval_dtor(&c); zval_dtor(&d); zval_dtor(&a); zval_dtor(&b);
    }
    /* ---- 4. an input PHP lets go of before the value is asked for stays alive for the chain ---- */
    {
        const int s[1] = {1000};
        zval t = placed(s, 1, 302, 1, 2, 1);
        zval c = op2(ZEND_DIV, number(1.0), t, 0, 0);         /* $c = 1 / $t; unset($t); */
        zval_dtor(&t);
        NDArray *hc = dump("unset.c", &c);
        if (hc) NDArray_FREE(hc);
        zval_dtor(&c);
    }
    /* ---- 5. consumers see finished values: a reduction and a comparison of pending operands ---- */
    {
        zval c = op2(ZEND_MUL, UN(exp, e.x, 0), e.y, 1, 0);
        const unsigned long long l0 = launches();
        const float total = method_sum(&c);
        CHECK(launches() - l0 >= 2, "sum of a pending value: %llu launch(es)", launches() - l0);   /* the chain, then the reduction */
        zval one = placed(s1, 1, 303, 7, 8, 1);
        float sum_host[1] = {total};
        const int s0[1] = {1};
        zval total_obj = object_of(NDArray_FromHostBuffer(sum_host, s0, 1));
        NDArray *ht = dump("consumer.sum", &total_obj);
        if (ht) NDArray_FREE(ht);
        zval c2 = op2(ZEND_MUL, UN(exp, e.x, 0), e.y, 1, 0);
        zval eq = method_equal(&c, &c2);                      /* nd::equal($c, $c2): $c2 pending, $c computed above */
        NDArray *he = dump("consumer.equal", &eq);
        if (he) NDArray_FREE(he);
        zval_dtor(&eq); zval_dtor(&c2); zval_dtor(&c); zval_dtor(&one); zval_dtor(&total_obj);
    }
    /* ---- 5b. the full reductions reduce a pending operand INSIDE its chain's kernel: one launch, the values never stored ---- */
    {
        static const struct { const char *name; double (*fn)(zval *); int exact; } kRed[] = {
            {"sum", patched_method_sum, 0}, {"mean", patched_method_mean, 0}, {"max", patched_method_max, 1}, {"min", patched_method_min, 1},
            {"prod", patched_method_prod, 0}, {"mse", patched_method_mean, 0}};
        for (int k = 0; k < 6; k++) {
            double got[2];
            unsigned long long cost[2];
            for (int lazy = 1; lazy >= 0; lazy--) {
                NPH_SetLazy(lazy);
                NPH_GetLazyStats(&st0);
                const unsigned long long l0 = launches();
                /* nd::sum(nd::exp($x) * $y) ...; a product that stays near 1: nd::prod(nd::clip($w, 0.9998, 1.0002)); a mean squared
                 * error: nd::mean(($x - $y) ** 2) */
                zval c = {IS_UNDEF, 0.0, 0};
                if (k == 4) patched_method_clip(&e.w, 0.9998, 1.0002, &c);
                else if (k == 5) c = op2(ZEND_POW, op2(ZEND_SUB, e.x, e.y, 0, 0), number(2.0), 1, 0);
                else c = op2(ZEND_MUL, UN(exp, e.x, 0), e.y, 1, 0);
                got[lazy] = kRed[k].fn(&c);
                cost[lazy] = launches() - l0;
                NPH_GetLazyStats(&st1);
                if (lazy) CHECK(st1.fused_reductions - st0.fused_reductions == 1 && NPH_IsPending(buffer_peek(&c)),
                                "%s of a pending value: %lu fused reductions, pending afterwards %d", kRed[k].name,
                                st1.fused_reductions - st0.fused_reductions, NPH_IsPending(buffer_peek(&c)));
                zval_dtor(&c);
            }
            NPH_SetLazy(1);
            const int steps = k == 4 ? 1 : 2;
            printf("%-4s of a pending value: %llu launch(es), %llu without chains; %.9g / %.9g\n", kRed[k].name, cost[1], cost[0], got[1], got[0]);
            CHECK(cost[1] == 1 && cost[0] == (unsigned long long) steps + 1, "%s: %llu launches with chains, %llu without", kRed[k].name, cost[1], cost[0]);
            const double tol = kRed[k].exact ? 0.0 : 2e-6 * (got[0] < 0 ? -got[0] : got[0]);
            CHECK((got[1] > got[0] ? got[1] - got[0] : got[0] - got[1]) <= tol, "%s: %.9g inside the chain, %.9g of the stored values", kRed[k].name, got[1], got[0]);
            float both[2] = {(float) got[1], (float) got[0]};
            const int s2v[1] = {2};
            char label[32];
            snprintf(label, sizeof label, "reduce.%s", kRed[k].name);
            zval obj = object_of(NDArray_FromHostBuffer(both, s2v, 1));
            NDArray *h = dump(label, &obj);
            if (h) NDArray_FREE(h);
            zval_dtor(&obj);
        }
        /* nd::sum(nd::exp($x) * $y, $axis): the last axis, and the first of a 2-d array, run inside the chain's kernel (one launch
         * or two: a column sum folds its chunks in a second one); the middle axis of a 3-d array computes the value first */
        for (int axis = 1; axis >= 0; axis--) {
            char label[32];
            const unsigned long long l0 = launches();
            zval c = op2(ZEND_MUL, UN(exp, e.x, 0), e.y, 1, 0);
            NPH_GetLazyStats(&st0);
            zval red = patched_method_sum_axis(&c, axis);
            NPH_GetLazyStats(&st1);
            const unsigned long long cost = launches() - l0;
            CHECK(st1.fused_reductions - st0.fused_reductions == 1 && NPH_IsPending(buffer_peek(&c)) && cost <= 2,
                  "sum(axis %d) of a pending value: %llu launches, %lu fused", axis, cost, st1.fused_reductions - st0.fused_reductions);
            printf("sum(axis %d) of a pending value: %llu launch(es), nothing stored\n", axis, cost);
            snprintf(label, sizeof label, "reduce.axis%d", axis);
            NDArray *hr = dump(label, &red);
            if (hr) NDArray_FREE(hr);
            zval_dtor(&red); zval_dtor(&c);
        }
        {
            const int s3[3] = {6, 37, 20};
            zval a3 = placed(s3, 3, 208, -2, 2, 1);
            zval c = UN(exp, a3, 0);
            NPH_GetLazyStats(&st0);
            zval red = patched_method_sum_axis(&c, 1);
            NPH_GetLazyStats(&st1);
            CHECK(st1.fused_reductions == st0.fused_reductions && st1.flushed_chains - st0.flushed_chains == 1 && !NPH_IsPending(buffer_peek(&c)),
                  "sum over the middle axis of a pending 3-d value must compute the value first");
            NDArray *hr = dump("reduce.mid3", &red);
            if (hr) NDArray_FREE(hr);
            zval_dtor(&red); zval_dtor(&c); zval_dtor(&a3);
        }
    }
    /* ---- 6. errors are the eager path's: a CPU array next to a GPU array, shapes that do not broadcast ---- */
    {
        const int s[2] = {4, 5}, t[1] = {7};
        zval g = placed(s, 2, 304, 0, 1, 1), h = placed(s, 2, 305, 0, 1, 0), odd = placed(t, 1, 306, 0, 1, 1);
        numpower_host_clear_error();
        zval r1 = op2(ZEND_ADD, UN(exp, g, 0), h, 1, 0);
        CHECK(Z_TYPE_P(&r1) == IS_UNDEF && strstr(numpower_host_last_error(), "Device mismatch") != NULL, "GPU + CPU array: %s", numpower_host_last_error());
        numpower_host_clear_error();
        zval r2 = op2(ZEND_ADD, UN(exp, g, 0), odd, 1, 0);
        CHECK(Z_TYPE_P(&r2) == IS_UNDEF && strstr(numpower_host_last_error(), "broadcast") != NULL, "4x5 + 7: %s", numpower_host_last_error());
        numpower_host_clear_error();
        zval_dtor(&g); zval_dtor(&h); zval_dtor(&odd);
    }
    /* ---- 7. random programs, chains on against chains off ---- */
    {
        static const int kShapes[4][2] = {{37, 53}, {64, 64}, {8, 1001}, {129, 255}};
        unsigned long evaluated = 0;
        NPH_GetLazyStats(&st0);
        const unsigned long long l0 = launches();
        unsigned long long l_lazy = 0, l_eager = 0;
        for (unsigned seed = 0; seed < 12; seed++) {
            const int rows = kShapes[seed % 4][0], cols = kShapes[seed % 4][1];
            const int vars = seed < 8 ? 20 : kPoolMax;      /* the last four: more live variables than the table of pending chains holds */
            const unsigned long long before = launches();
            const int statements = seed < 8 ? 400 : 1200;
            const unsigned long long lazy_digest = stress(1, seed, statements, rows, cols, vars, &evaluated);
            CHECK(NPH_PendingCount() == 0, "stress %u: %d chains pending after every variable was released", seed, NPH_PendingCount());
            const unsigned long long mid = launches();
            const unsigned long long eager_digest = stress(0, seed, statements, rows, cols, vars, &evaluated);
            l_lazy += mid - before;
            l_eager += launches() - mid;
            CHECK(lazy_digest == eager_digest, "stress %u (%d x %d): values differ between chains on (%016llx) and off (%016llx)", seed, rows, cols,
                  lazy_digest, eager_digest);
        }
        NPH_GetLazyStats(&st1);
        printf("stress: 12 random programs of 400 / 1200 statements, %lu values read on the way: identical with chains on and off; %llu launches with chains, "
               "%llu without (chains flushed %lu holding %lu steps, discarded %lu, eager steps %lu; at most %d chains pending at once)\n", evaluated / 2,
               l_lazy, l_eager, st1.flushed_chains - st0.flushed_chains, st1.flushed_steps - st0.flushed_steps,
               st1.discarded_chains - st0.discarded_chains, st1.eager_steps - st0.eager_steps, g_stress_most_pending);
        CHECK(g_stress_most_pending == NPH_MAX_PENDING, "the stress never filled the table of pending chains (%d of %d)", g_stress_most_pending, NPH_MAX_PENDING);
        (void) l0;
    }
    CHECK(NPH_PendingCount() == 0, "%d chains pending at the end", NPH_PendingCount());
    zval_dtor(&e.x); zval_dtor(&e.y); zval_dtor(&e.p); zval_dtor(&e.row); zval_dtor(&e.col); zval_dtor(&e.wide); zval_dtor(&e.w);
    fclose(g_out);
    if (NDArray_LiveDeviceAllocations() != 0) {
        fprintf(stderr, "lazy_bodies: %ld device allocations leaked\n", NDArray_LiveDeviceAllocations());
        g_failed = 1;
    }
    printf("lazy_bodies gpu: %s\n", g_failed ? "FAILED" : "ok");
    return g_failed;
}
"""


def _indent(text: str, pad: str) -> str:
    return "\n".join((pad + line) if line else line for line in text.split("\n"))


def apply_edit(text: str, e: Edit, keep_cuda: bool = False):
    """-> (new text, number of matches); raises PatchError if the anchor count is not e.expect."""
    rx = re.compile(e.anchor, re.M)
    matches = list(rx.finditer(text))
    if len(matches) != e.expect:
        raise PatchError("%s: anchor for [%s] matched %d time(s), expected %d" % (e.file, e.what, len(matches), e.expect))
    out, pos = [], 0
    for m in matches:
        old = m.group("old")
        pad = re.match(r"[ \t]*", old).group(0)
        new = e.new.replace("@HIP_M4_BLOCK@", HIP_M4_BLOCK)
        if e.template:
            new = m.expand(new)
        if e.after:
            last_pad = re.match(r"[ \t]*", old.split("\n")[-1]).group(0)
            if old.lstrip().startswith("#"):
                last_pad = ""          # behind a preprocessor line: at the margin
            rep = "%s\n#ifdef HAVE_NP_HIP\n%s\n#endif" % (old, _indent(new, last_pad))
        elif e.wrap and (keep_cuda or e.guard):
            rep = "#ifdef HAVE_NP_HIP\n%s\n#else\n%s\n#endif" % (_indent(new, pad), old)
        elif e.wrap:
            rep = _indent(new, pad)
        else:
            rep = new
        out.append(text[pos:m.start("old")])
        out.append(rep)
        pos = m.end("old")
    out.append(text[pos:])
    return "".join(out), len(matches)


_COND = re.compile(r"^\s*#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)$")
KNOWN = {"HAVE_NP_HIP": True, "HAVE_CUBLAS": True, "HAVE_CUDNN": False}
_CUDA_NAME = re.compile(r"\b(cuda[A-Z]\w*|cublas[A-Z]\w*|CUBLAS_\w+)\b|<cuda_runtime\.h>|<cublas_v2\.h>")


def _eval(kind: str, expr: str):
    """True / False when the condition is decided by KNOWN, None when it is not (both branches stay visible)."""
    expr = re.sub(r"/\*.*?\*/|//.*$", "", expr).strip()
    if kind in ("ifdef", "ifndef"):
        if expr in KNOWN:
            return KNOWN[expr] if kind == "ifdef" else not KNOWN[expr]
        return None
    m = re.fullmatch(r"(!?)\s*(?:defined\s*\(?\s*)?(\w+)\s*\)?", expr)
    if m and m.group(2) in KNOWN:
        v = KNOWN[m.group(2)]
        return (not v) if m.group(1) else v
    return None


def _strip_comments(text: str) -> str:
    """Comments blanked out, line numbers kept (string literals in this code base hold no comment openers)."""
    text = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def hip_visible_cuda_names(text: str):
    """[(line number, line)] of lines that a --with-hip build compiles and that still name CUDA / cuBLAS."""
    bad, stack = [], []   # stack of [state of the current branch, state of the condition as first written]
    for no, line in enumerate(_strip_comments(text).split("\n"), 1):
        m = _COND.match(line)
        if m:
            kind, expr = m.group(1), m.group(2)
            if kind in ("if", "ifdef", "ifndef"):
                v = _eval(kind, expr)
                stack.append([v, v])
            elif kind == "elif":
                first = stack[-1][1]
                stack[-1][0] = False if first is True else None
            elif kind == "else":
                first = stack[-1][1]
                stack[-1][0] = None if first is None else (not first)
            elif kind == "endif":
                stack.pop()
            continue
        if any(s[0] is False for s in stack):
            continue
        if _CUDA_NAME.search(line):
            bad.append((no, line.strip()))
    if stack:
        raise PatchError("unbalanced conditionals")
    return bad


def raw_cuda_names(text: str):
    """[(line number, line)] of EVERY line that names CUDA / cuBLAS — comments and inactive branches included."""
    return [(no, line.strip()) for no, line in enumerate(text.split("\n"), 1) if _CUDA_NAME.search(line)]


def check_tree(out: Path, raw: bool = True):
    """Every C source a --with-hip build compiles.  raw (the HIP-only default): no CUDA / cuBLAS name anywhere in the text;
    not raw (--keep-cuda): none on a line a --with-hip build's preprocessor lets through."""
    problems = []
    for path in sorted(list(out.glob("*.c")) + list(out.glob("*.h")) + list(out.glob("src/**/*.c")) + list(out.glob("src/**/*.h"))):
        rel = path.relative_to(out).as_posix()
        if rel.startswith(REPLACED_BY_GLUE):
            continue
        text = path.read_text(errors="replace")
        if raw and rel.startswith("src/hip/"):
            text = _strip_comments(text)   # the glue's comments CITE the reference statement each entry point replaces
        for no, line in (raw_cuda_names(text) if raw else hip_visible_cuda_names(text)):
            if rel.startswith("src/hip/") and re.search(r"\bcuda_\w+", line) and not _CUDA_NAME.search(line):
                continue
            problems.append("%s:%d: %s" % (rel, no, line))
    return problems


def apply(checkout: Path, out: Path, keep_cuda: bool = False):
    """-> {edit description: times applied}.  Raises PatchError on any anchor problem or leftover CUDA name."""
    if out.exists():
        raise PatchError("%s exists; give a fresh output directory" % out)
    if not (checkout / "numpower.c").exists() or not (checkout / "config.m4").exists():
        raise PatchError("%s does not look like a NumPower checkout (numpower.c / config.m4 missing)" % checkout)
    shutil.copytree(checkout, out, ignore=shutil.ignore_patterns(".git"))
    applied = {}
    by_file = {}
    for e in EDITS:
        by_file.setdefault(e.file, []).append(e)
    for file, edits in by_file.items():
        path = out / file
        if not path.exists():
            raise PatchError("%s: file missing from the checkout" % file)
        text = path.read_text()
        for e in edits:
            text, n = apply_edit(text, e, keep_cuda)
            applied[e.what] = n
        path.write_text(text)
    (out / "src" / "hip").mkdir(parents=True, exist_ok=True)
    for g in GLUE_FILES:
        shutil.copy2(ROOT / g, out / "src" / "hip" / Path(g).name)
    problems = check_tree(out, raw=not keep_cuda)
    if problems:
        raise PatchError("CUDA / cuBLAS names still %s:\n  " % ("visible to a --with-hip build" if keep_cuda else "in the tree's text") +
                         "\n  ".join(problems))
    return applied


def main(argv):
    keep_cuda = "--keep-cuda" in argv
    argv = [a for a in argv if a != "--keep-cuda"]
    if len(argv) != 3:
        print(__doc__, file=sys.stderr)
        return 2
    try:
        applied = apply(Path(argv[1]).resolve(), Path(argv[2]).resolve(), keep_cuda)
    except PatchError as e:
        print("apply_with_hip: %s" % e, file=sys.stderr)
        return 2
    for what, n in applied.items():
        print("applied x%d  %s" % (n, what))
    print("%d edits in %d files; glue in src/hip/.  Next: phpize && ./configure --with-hip=<dir of libnp_hip.so> && make"
          % (len(applied), len({e.file for e in EDITS})))
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
