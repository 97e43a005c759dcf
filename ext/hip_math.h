/*
 * hip_math.h — the symbols of the reference's GPU math header (src/ndmath/cuda/cuda_math.h:10-79)
 * as implemented by ext/hip_math.c over libnp_hip.so.
 *
 * In a PHP build the reference's own cuda_math.h stays the header every C file includes — the
 * names and signatures below are identical to it on purpose (that is what makes the back end a
 * drop-in for `NDArrayMathGPU_ElementWise(nda, cuda_float_sin)` at numpower.c:1651 and the other
 * call sites listed per function in hip_math.c).  This header exists so that the glue can be
 * compiled, linked and tested where PHP is absent; it declares only raw-pointer functions (no
 * NDArray type), and additionally the four lookup functions the out-of-place drivers use.
 *
 * `nblocks` is what the reference passes as a launch-size hint (usually the element count); the
 * HIP back end sizes its own grids and ignores it, except where the reference uses it AS the
 * element count (the in-place unary family: cuda_math.cu:1160-1166).
 */
#ifndef NUMPOWER_AMD_EXT_HIP_MATH_H
#define NUMPOWER_AMD_EXT_HIP_MATH_H

#ifdef __cplusplus
extern "C" {
#endif

/* function-pointer shapes of cuda_math.h:10-13 */
typedef void (*ElementWiseFloatGPUOperation)(int, float *);
typedef void (*ElementWiseFloatGPUOperation2F)(int, float *, float, float);
typedef void (*ElementWiseFloatGPUOperation1F)(int, float *, float);
typedef void (*ElementWiseFloatGPUOperation1N)(int, float *, float *);

/* ---- in-place unary family: d_array[i] = f(d_array[i]) for i < nblocks -------------------- */
#define NP_HIP_MATH_UNARY_LIST(X)                                                                   \
    X(abs, NP_ABS) X(sqrt, NP_SQRT) X(exp, NP_EXP) X(expm1, NP_EXPM1) X(log, NP_LOG)              \
    X(logb, NP_LOGB) X(log2, NP_LOG2) X(log1p, NP_LOG1P) X(log10, NP_LOG10) X(sin, NP_SIN)        \
    X(cos, NP_COS) X(tan, NP_TAN) X(arcsin, NP_ARCSIN) X(arccos, NP_ARCCOS) X(arctan, NP_ARCTAN)  \
    X(degrees, NP_DEGREES) X(radians, NP_RADIANS) X(sinh, NP_SINH) X(cosh, NP_COSH)               \
    X(tanh, NP_TANH) X(arcsinh, NP_ARCSINH) X(arccosh, NP_ARCCOSH) X(arctanh, NP_ARCTANH)         \
    X(rint, NP_RINT) X(fix, NP_FIX) X(ceil, NP_CEIL) X(floor, NP_FLOOR) X(sinc, NP_SINC)          \
    X(trunc, NP_TRUNC) X(negate, NP_NEGATE) X(sign, NP_SIGN) X(positive, NP_POSITIVE)             \
    X(reciprocal, NP_RECIPROCAL)
/* Two unary methods have no usable device function in the reference: PHP_METHOD(rsqrt) hands
 * cuda_float_arccos to the driver (numpower.c:1791, a slip) and PHP_METHOD(exp2) has no device branch at
 * all (numpower.c:3153 maps the CPU kernel over whatever pointer the array holds).  These two names are
 * NOT in cuda_math.h; tools/apply_with_hip.py declares them in numpower.c and points the two methods at them. */
#define NP_HIP_MATH_EXTRA_UNARY_LIST(X) X(rsqrt, NP_RSQRT) X(exp2, NP_EXP2)
#define NP_HIP_MATH_DECLARE_UNARY(name, code) void cuda_float_##name(int nblocks, float *d_array);
NP_HIP_MATH_UNARY_LIST(NP_HIP_MATH_DECLARE_UNARY)
NP_HIP_MATH_EXTRA_UNARY_LIST(NP_HIP_MATH_DECLARE_UNARY)
#undef NP_HIP_MATH_DECLARE_UNARY
void cuda_float_clip(int nblocks, float *d_array, float minVal, float maxVal);
void cuda_float_round(int nblocks, float *d_array, float decimals);
void cuda_float_arctan2(int nblocks, float *d_array, float *y_array);

/* ---- binary elementwise: rtn[i] = a[i] (op) b[i] for i < nelements ------------------------- */
void cuda_add_float(int nblocks, float *a, float *b, float *rtn, int nelements);
void cuda_subtract_float(int nblocks, float *a, float *b, float *rtn, int nelements);
void cuda_multiply_float(int nblocks, float *a, float *b, float *rtn, int nelements);
void cuda_divide_float(int nblocks, float *a, float *b, float *rtn, int nelements);
void cuda_mod_float(int nblocks, float *a, float *b, float *rtn, int nelements);
void cuda_pow_float(int nblocks, float *a, float *b, float *rtn, int nelements);

/* ---- comparisons: result[i] = 1.0f / 0.0f ------------------------------------------------------ */
void cuda_float_compare_equal(int nblocks, float *a_array, float *b_array, float *result, int n);
void cuda_float_compare_not_equal(int nblocks, float *a_array, float *b_array, float *result, int n);
void cuda_float_compare_greater(int nblocks, float *a_array, float *b_array, float *result, int n);
void cuda_float_compare_greater_equal(int nblocks, float *a_array, float *b_array, float *result, int n);
void cuda_float_compare_less(int nblocks, float *a_array, float *b_array, float *result, int n);
void cuda_float_compare_less_equal(int nblocks, float *a_array, float *b_array, float *result, int n);
int cuda_equal_float(int nblocks, float *a, float *b, int nelements);   /* 1 = identical */

/* ---- reductions to one host float --------------------------------------------------------------- */
void cuda_sum_float(int nblocks, float *a, float *rtn, int nelements);    /* *rtn += sum  (host ptr) */
void cuda_prod_float(int nblocks, float *a, float *rtn, int nelements);   /* *rtn *= prod (host ptr) */
float cuda_max_float(float *a, int nelements);
float cuda_min_float(float *a, int nelements);

/* ---- fill, matrix * vector, outer product, transpose ------------------------------------------- */
void cuda_fill_float(float *a, float value, int n);
void cuda_float_multiply_matrix_vector(int nblocks, float *a_array, float *b_array, float *result, int rows, int cols);
void cuda_calculate_outer_product(int m, int n, float *a_array, float *b_array, float *r_array);
void cuda_float_transpose(int tiledim, int blockrows, const float *d_in, float *d_out, int width, int height);

/* ---- dense factorizations: OUT OF SCOPE (linalg.c beyond matmul/dot/outer).  Present so
 * that an extension built --with-hip links and loads; each raises "... is not available on the HIP
 * back end" through np_ext_throw and returns failure / leaves its outputs untouched. ---- */
int cuda_svd_float(float *d_A, float *d_U, float *d_V, float *d_S, int m, int n);
int cuda_det_float(float *a, float *result, int n);
void cuda_matrix_float_inverse(float *matrix, int n);
void cuda_float_lu(float *matrix, float *L, float *U, float *P, int size);
void cuda_lstsq_float(float *A, int m, int n, float *B, int k, float *X);

/* ---- what the out-of-place drivers (hip_math_drivers.c) ask: which np_hip.h op is this pointer?
 * np_unary_op / np_binary_op code, or -1 if `op` is not one of the functions above (a caller's own
 * kernel wrapper): the driver then falls back to the reference's copy + in-place call. ---- */
int np_hip_math_unary_code(ElementWiseFloatGPUOperation op);
int np_hip_math_unary1f_code(ElementWiseFloatGPUOperation1F op);
int np_hip_math_unary2f_code(ElementWiseFloatGPUOperation2F op);
int np_hip_math_binary1n_code(ElementWiseFloatGPUOperation1N op);

#ifdef __cplusplus
}
#endif
#endif
