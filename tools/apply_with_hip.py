#!/usr/bin/env python3
"""apply_with_hip.py — turn a NumPower checkout into a `--with-hip` tree (as code).

    python tools/apply_with_hip.py <NumPower checkout> <output directory>

Copies the checkout to <output directory>, applies every edit of INTEGRATION.md section 2a by ANCHORED regex and
copies the glue (ext/*.c, ext/*.h, include/np_hip.h) to <output>/src/hip/.  Every edit names its file, the exact
text it expects to find and how many times; a missing or ambiguous anchor is a hard error (exit status 2) — the
table cannot silently rot when the reference moves.  Each replaced statement is kept:

    #ifdef HAVE_NP_HIP
        <the np_* / v* call>
    #else
        <the reference's CUDA-runtime statement, untouched>
    #endif

so the output still builds `--with-cuda`; `--with-hip` (config.m4, added by the last edits) defines HAVE_CUBLAS —
the name the reference's C files gate every NDARRAY_DEVICE_GPU branch on — and HAVE_NP_HIP.

After the edits the tool CHECKS the tree (check_tree): with HAVE_NP_HIP and HAVE_CUBLAS defined and HAVE_CUDNN
undefined, no preprocessor-visible line of the extension's C sources may name the CUDA runtime or cuBLAS
(cuda[A-Z]*, cublas[A-Z]*, CUBLAS_*, <cuda_runtime.h>, <cublas_v2.h>).  src/gpu_alloc.c and src/ndmath/cuda/ are
replaced wholesale by the glue and are not part of a --with-hip build.

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
    wrap: bool = True  # True: #ifdef HAVE_NP_HIP new #else old #endif;  False: plain substitution by `new`
    context: str = ""  # C declarations of the locals of the surrounding reference function that `new` uses (CONTEXTS
                       # below): with them the new text is compiled on its own — snippet_check_source() — against
                       # include/np_hip.h, include/numpower_host.h and ext/hip_math.h; "" = nothing to compile


def _cuda_includes(file: str, line: str) -> Edit:
    return Edit(file, "%s: CUDA headers -> <np_hip.h>" % line,
                r"^(?P<old>#include <cuda_runtime\.h>\n#include <cublas_v2\.h>)$",
                "#include <np_hip.h>")


_SYNC = (r"^(?P<old>[ \t]*cudaDeviceSynchronize\(\);)$")

EDITS = [
    # ---- headers ----
    _cuda_includes("numpower.c", "numpower.c:31-32"),
    _cuda_includes("src/initializers.c", "initializers.c:15-16"),
    _cuda_includes("src/ndarray.c", "ndarray.c:18-19"),
    _cuda_includes("src/ndmath/arithmetics.c", "arithmetics.c:14-15"),
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
         "}"),
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
    "numpower.c:623-633 NDArray::setDevice": "int numDevices = 0; long deviceId = 0;",
    "debug.c:220-254 NDArray_DumpDevices": " ",
    "numpower.c:1791 PHP_METHOD(rsqrt): the right device function": "NDArray *rtn = 0, *nda = 0;",
    "numpower.c:3153 PHP_METHOD(exp2): a device branch": "NDArray *rtn = 0, *nda = 0;",
}

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
  CFLAGS+=" -DNUMPOWER_NDARRAY_HEADER='\\"src/initializers.h\\"' "
  NP_GPU_ALLOC_SOURCES="src/hip/gpu_alloc_hip.c src/hip/hip_math.c src/hip/hip_math_drivers.c src/hip/zend_hooks.c"
fi'''

GLUE_FILES = ["ext/gpu_alloc_hip.c", "ext/hip_math.c", "ext/hip_math.h", "ext/hip_math_drivers.c", "ext/zend_hooks.c",
              "ext/np_ext_hooks.h", "include/np_hip.h"]
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
           '#include "np_hip.h"', '#include "numpower_host.h"', '#include "hip_math.h"',
           "void zend_throw_error(void *exception_ce, const char *format, ...);   /* Zend/zend_exceptions.h */",
           "NDArray *NDArray_Map(NDArray *array, float (*op)(float));                /* src/ndarray.h */",
           "float float_exp2(float val);                                             /* src/ndmath/double_math.h */", ""]
    for k, e in enumerate(EDITS):
        if not e.context:
            continue
        ret = "void *" if "return NULL;" in e.new else "void "
        out.append("/* %s */" % e.what)
        out.append("%ssnippet_%d(void) {" % (ret, k))
        out.append("    " + e.context)
        out.append(_indent(e.new, "    "))
        if ret == "void *":
            out.append("    return (void *) 0;")
        out.append("}")
        out.append("")
    return "\n".join(out)


def _indent(text: str, pad: str) -> str:
    return "\n".join((pad + line) if line else line for line in text.split("\n"))


def apply_edit(text: str, e: Edit):
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
        if e.wrap:
            rep = "#ifdef HAVE_NP_HIP\n%s\n#else\n%s\n#endif" % (_indent(new, pad), old)
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


def check_tree(out: Path):
    """Every C source a --with-hip build compiles: no CUDA / cuBLAS name on a visible line."""
    problems = []
    for path in sorted(list(out.glob("*.c")) + list(out.glob("src/**/*.c")) + list(out.glob("src/**/*.h"))):
        rel = path.relative_to(out).as_posix()
        if rel.startswith(REPLACED_BY_GLUE):
            continue
        for no, line in hip_visible_cuda_names(path.read_text(errors="replace")):
            if rel.startswith("src/hip/") and re.search(r"\bcuda_\w+", line) and not _CUDA_NAME.search(line):
                continue
            problems.append("%s:%d: %s" % (rel, no, line))
    return problems


def apply(checkout: Path, out: Path):
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
            text, n = apply_edit(text, e)
            applied[e.what] = n
        path.write_text(text)
    (out / "src" / "hip").mkdir(parents=True, exist_ok=True)
    for g in GLUE_FILES:
        shutil.copy2(ROOT / g, out / "src" / "hip" / Path(g).name)
    problems = check_tree(out)
    if problems:
        raise PatchError("CUDA / cuBLAS names still visible to a --with-hip build:\n  " + "\n  ".join(problems))
    return applied


def main(argv):
    if len(argv) != 3:
        print(__doc__, file=sys.stderr)
        return 2
    try:
        applied = apply(Path(argv[1]).resolve(), Path(argv[2]).resolve())
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
