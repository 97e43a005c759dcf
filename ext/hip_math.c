/*
 * hip_math.c — `hip_math`: the reference's GPU math entry points (src/ndmath/cuda/cuda_math.h:14-79,
 * implemented for CUDA in src/ndmath/cuda/cuda_math.cu:1064-1558) over the C ABI of libnp_hip.so.
 *
 * Pure C, raw device pointers in and out, no NDArray, no Zend: with this file (and gpu_alloc_hip.c)
 * in place of cuda_math.cu / gpu_alloc.c, the reference's own C files — arithmetics.c, logic.c,
 * linalg.c, ndarray.c, manipulation.c, initializers.c, numpower.c — compile and link unchanged and
 * their NDARRAY_DEVICE_GPU branches run on an MI355X.  Call sites are cited per function.
 *
 * Result policy (BASELINE north_star: "results match the reference's own CPU path"): where the
 * reference's CUDA kernels and its AVX2 CPU loops disagree, these entry points follow the CPU:
 *   multiply  zero products are -0.0f in the 8-wide body and +0.0f in the tail   (arithmetics.c:397-412)
 *   mod       a - floor(a/b)*b in the body, fmodf in the tail                    (arithmetics.c:788-800)
 *   equal / not_equal   exact compare in the body, |a-b| <= 1e-7 in the tail     (logic.c:541-552,642-655)
 * (np_hip.h: NP_QUIRK_AVX_BODY, body = the first np_avx_body_end(n) elements.)  The CUDA kernels use
 * IEEE multiply, fmodf and the tolerance form everywhere (cuda_math.cu:243,617-631).
 *
 * Errors: the reference ignores every CUDA error here.  A failing np_* call is raised through
 * np_ext_throw (np_ext_hooks.h) with the back end's message; void functions then just return.
 * No call blocks except the ones that hand a value back to the host.
 */
#include <stddef.h>

#include "hip_math.h"
#include "np_ext_hooks.h"
#include "np_hip.h"

static void raise_if(int rc) {
    if (rc != NP_OK) np_ext_throw_last();
}

/* The reference's interface counts elements in an `int` (cuda_math.h:14-79), and its callers pass NDArray_NUMELEMENTS — a long:
 * an array of 2^31 elements or more arrives here as a NEGATIVE count (or a small wrong one).  Negative is refused out loud instead
 * of being treated as empty: `nd::sum()` of such an array must not come back as 0.  (The paths a `--with-hip` tree sends GPU
 * arrays through by default — hip_fast.c, hip_math_drivers.c, hip_lazy.c — count in size_t and do not come here.) */
static size_t count(int n) {
    if (n < 0) {
        np_ext_throw("element count does not fit the int of the reference's cuda_* interface (2^31 - 1 elements)");
        return 0;
    }
    return (size_t)n;
}

/* ---- unary family --------------------------------------------------------------------------------
 * numpower.c:1651-3348 pass these as `op` to NDArrayMathGPU_ElementWise (table: numpower.c:5136-5174);
 * called directly they work in place, as the CUDA versions do (cuda_math.cu:1160-1166 and siblings). */
#define NP_HIP_MATH_DEFINE_UNARY(name, code)                                                        \
    void cuda_float_##name(int nblocks, float *d_array) {                                          \
        raise_if(np_unary(code, d_array, d_array, count(nblocks), 0.0f, 0.0f));                    \
    }
NP_HIP_MATH_UNARY_LIST(NP_HIP_MATH_DEFINE_UNARY)
NP_HIP_MATH_EXTRA_UNARY_LIST(NP_HIP_MATH_DEFINE_UNARY)   /* rsqrt, exp2: see hip_math.h */
#undef NP_HIP_MATH_DEFINE_UNARY

/* numpower.c:2487 (clip), :2959 (round) */
void cuda_float_clip(int nblocks, float *d_array, float minVal, float maxVal) {
    raise_if(np_unary(NP_CLIP, d_array, d_array, count(nblocks), minVal, maxVal));
}
void cuda_float_round(int nblocks, float *d_array, float decimals) {
    raise_if(np_unary(NP_ROUND, d_array, d_array, count(nblocks), decimals, 0.0f));
}
/* numpower.c:1899: d_array[i] = atan2f(d_array[i], y_array[i]) (cuda_math.cu:489-494) */
void cuda_float_arctan2(int nblocks, float *d_array, float *y_array) {
    raise_if(np_binary(NP_ARCTAN2, d_array, NP_FULL, y_array, NP_FULL, d_array, 1, count(nblocks), 0, 0));
}

/* which np_unary_op is this function?  (drivers: out-of-place fast path) */
int np_hip_math_unary_code(ElementWiseFloatGPUOperation op) {
#define NP_HIP_MATH_MATCH_UNARY(name, code) if (op == cuda_float_##name) return code;
    NP_HIP_MATH_UNARY_LIST(NP_HIP_MATH_MATCH_UNARY)
    NP_HIP_MATH_EXTRA_UNARY_LIST(NP_HIP_MATH_MATCH_UNARY)
#undef NP_HIP_MATH_MATCH_UNARY
    return -1;
}
int np_hip_math_unary1f_code(ElementWiseFloatGPUOperation1F op) { return op == cuda_float_round ? NP_ROUND : -1; }
int np_hip_math_unary2f_code(ElementWiseFloatGPUOperation2F op) { return op == cuda_float_clip ? NP_CLIP : -1; }
int np_hip_math_binary1n_code(ElementWiseFloatGPUOperation1N op) { return op == cuda_float_arctan2 ? NP_ARCTAN2 : -1; }

/* ---- binary elementwise ---------------------------------------------------------------------------
 * arithmetics.c:243 (add), :522 (subtract), :307,390 (multiply), :658 (divide), :783 (mod), :908 (pow):
 * operands already broadcast to `nelements` by the caller. */
static void binary_full(int op, float *a, float *b, float *rtn, int nelements, int cpu_quirk) {
    const size_t n = count(nelements);
    raise_if(np_binary(op, a, NP_FULL, b, NP_FULL, rtn, 1, n, cpu_quirk ? NP_QUIRK_AVX_BODY : 0u,
                       cpu_quirk ? np_avx_body_end(n) : 0));
}
void cuda_add_float(int nblocks, float *a, float *b, float *rtn, int nelements) {
    (void)nblocks;
    binary_full(NP_ADD, a, b, rtn, nelements, 0);
}
void cuda_subtract_float(int nblocks, float *a, float *b, float *rtn, int nelements) {
    (void)nblocks;
    binary_full(NP_SUBTRACT, a, b, rtn, nelements, 0);
}
void cuda_multiply_float(int nblocks, float *a, float *b, float *rtn, int nelements) {
    (void)nblocks;
    /* arithmetics.c:302-316: the 0-d x 0-d short cut (nelements == 1) is a plain product on the CPU too */
    binary_full(NP_MULTIPLY, a, b, rtn, nelements, nelements > 1);
}
void cuda_divide_float(int nblocks, float *a, float *b, float *rtn, int nelements) {
    (void)nblocks;
    binary_full(NP_DIVIDE, a, b, rtn, nelements, 0);
}
void cuda_mod_float(int nblocks, float *a, float *b, float *rtn, int nelements) {
    (void)nblocks;
    binary_full(NP_MOD, a, b, rtn, nelements, 1);
}
void cuda_pow_float(int nblocks, float *a, float *b, float *rtn, int nelements) {
    (void)nblocks;
    binary_full(NP_POW, a, b, rtn, nelements, 0);
}

/* ---- comparisons: logic.c:121 (greater), :221 (less), :326 (less_equal), :427 (greater_equal),
 * :528 (equal), :629 (not_equal) ---- */
void cuda_float_compare_equal(int nblocks, float *a_array, float *b_array, float *result, int n) {
    (void)nblocks;
    binary_full(NP_EQUAL, a_array, b_array, result, n, 1);
}
void cuda_float_compare_not_equal(int nblocks, float *a_array, float *b_array, float *result, int n) {
    (void)nblocks;
    binary_full(NP_NOT_EQUAL, a_array, b_array, result, n, 1);
}
void cuda_float_compare_greater(int nblocks, float *a_array, float *b_array, float *result, int n) {
    (void)nblocks;
    binary_full(NP_GREATER, a_array, b_array, result, n, 0);
}
void cuda_float_compare_greater_equal(int nblocks, float *a_array, float *b_array, float *result, int n) {
    (void)nblocks;
    binary_full(NP_GREATER_EQUAL, a_array, b_array, result, n, 0);
}
void cuda_float_compare_less(int nblocks, float *a_array, float *b_array, float *result, int n) {
    (void)nblocks;
    binary_full(NP_LESS, a_array, b_array, result, n, 0);
}
void cuda_float_compare_less_equal(int nblocks, float *a_array, float *b_array, float *result, int n) {
    (void)nblocks;
    binary_full(NP_LESS_EQUAL, a_array, b_array, result, n, 0);
}
/* logic.c:683 (compare_ndarrays behind `==` and NDArray_ArrayEqual): 1 when no a[i] != b[i]
 * (cuda_math.cu:767-774: exact compare, NaN counts as a difference). */
int cuda_equal_float(int nblocks, float *a, float *b, int nelements) {
    int any = 1;
    (void)nblocks;
    raise_if(np_count_mismatch(NP_MISMATCH_EXACT, a, b, count(nelements), 0.0f, 0.0f, &any));
    return any ? 0 : 1;
}

/* ---- reductions -------------------------------------------------------------------------------------
 * arithmetics.c:41 (prod), :63,86 (sum, mean): `rtn` is a HOST float the caller initialised (0 for a
 * sum, 1 for a product) and the CUDA kernel accumulates onto (cuda_math.cu:921-945).  The CUDA product
 * kernel atomicAdd()s its per-block products (cuda_math.cu:776-800), which is not a product; this one
 * is (same value as the CPU loop up to summation order, tolerance stated in DESIGN.md §5). */
void cuda_sum_float(int nblocks, float *a, float *rtn, int nelements) {
    float v = 0.0f;
    (void)nblocks;
    const int rc = np_reduce_all(NP_SUM, a, count(nelements), &v);
    raise_if(rc);
    if (rc == NP_OK) *rtn += v;
}
void cuda_prod_float(int nblocks, float *a, float *rtn, int nelements) {
    float v = 1.0f;
    (void)nblocks;
    const int rc = np_reduce_all(NP_PROD, a, count(nelements), &v);
    raise_if(rc);
    if (rc == NP_OK) *rtn *= v;
}
/* ndarray.c:946 (max), :759 (min) */
float cuda_max_float(float *a, int nelements) {
    float v = 0.0f;
    raise_if(np_reduce_all(NP_MAX, a, count(nelements), &v));
    return v;
}
float cuda_min_float(float *a, int nelements) {
    float v = 0.0f;
    raise_if(np_reduce_all(NP_MIN, a, count(nelements), &v));
    return v;
}

/* ---- fill / matrix-vector / outer / transpose ----------------------------------------------------- */
/* initializers.c:639 */
void cuda_fill_float(float *a, float value, int n) { raise_if(np_fill(a, value, count(n))); }

/* linalg.c:373: result[rows] = a[rows x cols] . b[cols] */
void cuda_float_multiply_matrix_vector(int nblocks, float *a_array, float *b_array, float *result, int rows, int cols) {
    (void)nblocks;
    raise_if(np_sgemv(count(rows), count(cols), a_array, b_array, result));
}

/* linalg.c:746: r[m x n] = a[m] outer b[n] */
void cuda_calculate_outer_product(int m, int n, float *a_array, float *b_array, float *r_array) {
    raise_if(np_outer(a_array, count(m), b_array, count(n), r_array));
}

/* manipulation.c:124: the input is `height` rows of `width` floats, the output `width` rows of `height`
 * (cuda_math.cu:136-148).  The reference calls it IN PLACE on a copy of the array (d_in == d_out) with a
 * fixed 16 x 16 grid of 16 x 16 threads — racy, and only the first 256 x 256 elements are touched
 * (cuda_math.cu:1288-1294); here any size, and in-place calls go through a pooled temporary. */
void cuda_float_transpose(int tiledim, int blockrows, const float *d_in, float *d_out, int width, int height) {
    const size_t rows = count(height), cols = count(width);
    (void)tiledim;
    (void)blockrows;
    if (rows == 0 || cols == 0) return;
    if (d_in != d_out) {
        raise_if(np_transpose2d(d_in, d_out, 1, rows, cols));
        return;
    }
    void *tmp = NULL;
    const size_t bytes = rows * cols * sizeof(float);
    if (np_malloc(&tmp, bytes) != NP_OK) {
        np_ext_throw("device memory allocation failed");
        return;
    }
    int rc = np_memcpy_d2d(tmp, d_in, bytes);
    if (rc == NP_OK) rc = np_transpose2d((const float *)tmp, d_out, 1, rows, cols);
    raise_if(rc);
    np_free(tmp);   /* stream-ordered pool: safe right behind the launch */
}

/* ---- out of scope: link-completeness only ----------------------------------------------------------- */
static void not_available(const char *what) {
    char msg[160];
    size_t i = 0;
    const char *tail = " is not available on the HIP back end (dense factorizations are outside the hot path)";
    while (*what && i < 60) msg[i++] = *what++;
    while (*tail && i + 1 < sizeof(msg)) msg[i++] = *tail++;
    msg[i] = 0;
    np_ext_throw(msg);
}
int cuda_svd_float(float *d_A, float *d_U, float *d_V, float *d_S, int m, int n) {
    (void)d_A; (void)d_U; (void)d_V; (void)d_S; (void)m; (void)n;
    not_available("svd");
    return 0;
}
int cuda_det_float(float *a, float *result, int n) {
    (void)a; (void)result; (void)n;
    not_available("det");
    return 0;
}
void cuda_matrix_float_inverse(float *matrix, int n) {
    (void)matrix; (void)n;
    not_available("inv");
}
void cuda_float_lu(float *matrix, float *L, float *U, float *P, int size) {
    (void)matrix; (void)L; (void)U; (void)P; (void)size;
    not_available("lu");
}
void cuda_lstsq_float(float *A, int m, int n, float *B, int k, float *X) {
    (void)A; (void)m; (void)n; (void)B; (void)k; (void)X;
    not_available("lstsq");
}
