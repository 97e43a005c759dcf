dnl config.m4 fragment: `--with-hip` for the NumPower extension (MI355X / gfx950 through numpower_amd).
dnl
dnl Goes next to the reference's `--with-cuda` block (config.m4:7-16 of NumPower/numpower @ 2024_08_07)
dnl and is mutually exclusive with it.  Unlike the CUDA build there is NO device compiler step: the
dnl kernels live in a prebuilt libnp_hip.so (python -m numpower_amd.build, hipcc --offload-arch=gfx950)
dnl and everything added to the extension is plain C, so the stock phpize / libtool flow builds it —
dnl no Makefile.frag rule that re-compiles every source with nvcc (Makefile.frag:65-90 of the reference).
dnl
dnl   phpize && ./configure --with-hip=/opt/numpower_amd && make && make install
dnl
dnl What the option does:
dnl   1. adds  <prefix>/include (np_hip.h) and <prefix>/ext (hip_math.h, np_ext_hooks.h) to the includes
dnl   2. links <prefix>/numpower_amd/lib/libnp_hip.so (rpath'd)
dnl   3. defines HAVE_CUBLAS  — the reference's C files gate every NDARRAY_DEVICE_GPU branch on this
dnl      name (arithmetics.c:241, logic.c:119, linalg.c:52, ndarray.c:757 ...); it means "a device back
dnl      end is present", not "cuBLAS" — and HAVE_NP_HIP, which the few call sites that name the CUDA
dnl      runtime directly switch on (INTEGRATION.md section 2a lists them with file:line)
dnl   4. compiles the glue: ext/gpu_alloc_hip.c instead of src/gpu_alloc.c, ext/hip_math.c +
dnl      ext/hip_math_drivers.c instead of src/ndmath/cuda/cuda_math.cu, ext/hip_fast.c (what the GPU early-outs
dnl      tools/apply_with_hip.py inserts into arithmetics.c / logic.c / ndarray.c call), ext/hip_lazy.c (the pending
dnl      elementwise chains behind the NDArray handle: numpower.c's operators / unary methods append, buffer_get flushes),
dnl      ext/zend_hooks.c

PHP_ARG_WITH([hip],
  [for MI355X (HIP, gfx950) support through numpower_amd],
  [AS_HELP_STRING([--with-hip=DIR],
    [Run NDArray GPU paths on AMD MI355X; DIR = numpower_amd checkout or install prefix])],
  [no], [no])

if test "$PHP_HIP" != "no"; then
  if test "$PHP_CUDA" != "no" && test -n "$PHP_CUDA"; then
    AC_MSG_ERROR([--with-hip and --with-cuda are mutually exclusive])
  fi
  if test "$PHP_HIP" = "yes"; then
    AC_MSG_ERROR([--with-hip needs the numpower_amd prefix: --with-hip=/path/to/numpower_amd])
  fi
  AC_MSG_CHECKING([for np_hip.h under $PHP_HIP])
  if test ! -f "$PHP_HIP/include/np_hip.h"; then
    AC_MSG_ERROR([$PHP_HIP/include/np_hip.h not found])
  fi
  AC_MSG_RESULT([found])
  NP_HIP_LIBDIR="$PHP_HIP/numpower_amd/lib"
  if test ! -f "$NP_HIP_LIBDIR/libnp_hip.so"; then
    AC_MSG_ERROR([$NP_HIP_LIBDIR/libnp_hip.so not found: run `python -m numpower_amd.build` in $PHP_HIP first])
  fi

  PHP_ADD_INCLUDE([$PHP_HIP/include])
  PHP_ADD_INCLUDE([$PHP_HIP/ext])
  PHP_ADD_LIBRARY_WITH_PATH([np_hip], [$NP_HIP_LIBDIR], [NDARRAY_SHARED_LIBADD])
  PHP_CHECK_LIBRARY([np_hip], [np_sgemm],
    [AC_MSG_RESULT([numpower_amd device back end detected])],
    [AC_MSG_ERROR([libnp_hip.so does not export np_sgemm])],
    [-L$NP_HIP_LIBDIR])

  AC_DEFINE([HAVE_CUBLAS], [1], [a device back end is present (name kept from the CUDA build: the C files test it)])
  AC_DEFINE([HAVE_NP_HIP], [1], [the device back end is numpower_amd / MI355X])

  dnl glue sources, compiled by the ordinary C compiler together with the extension's own files
  NP_HIP_GLUE="$PHP_HIP/ext/gpu_alloc_hip.c $PHP_HIP/ext/hip_math.c $PHP_HIP/ext/hip_math_drivers.c $PHP_HIP/ext/hip_fast.c $PHP_HIP/ext/hip_lazy.c $PHP_HIP/ext/zend_hooks.c"
  NP_HIP_GLUE_CFLAGS="-DNUMPOWER_NDARRAY_HEADER='\"src/initializers.h\"'"
  PHP_SUBST([NP_HIP_GLUE])
  dnl In PHP_NEW_EXTENSION(ndarray, ...) of the reference's config.m4: drop src/gpu_alloc.c from the source
  dnl list and append $NP_HIP_GLUE; append $NP_HIP_GLUE_CFLAGS to the extension's extra cflags.
fi
