/*
 * np_oracle.c — CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / the timed CPU baseline — never as part of the product path.
 *
 * What it restates (file:line relative to the reference tree, NumPower/numpower @ 2024_08_07):
 *   binary ops incl. scalar expand + broadcast + AVX2 body/scalar tail
 *                      src/ndmath/arithmetics.c:160-926, src/ndarray.c:1124-1294
 *   unary float_* ops + NDArray_Map drivers
 *                      src/ndmath/double_math.c:9-265, src/ndarray.c:682-744
 *   full reductions    src/ndmath/arithmetics.c:36-102, src/ndarray.c:752-772,939-959
 *   axis reductions    src/ndarray.c:358-368,394-429,523-578 (+ mean: numpower.c:2660-2670)
 *   matmul / dot       src/ndmath/linalg.c:44-82,216-245,354-393 (cblas_sgemm / cblas_sgemv)
 *   comparisons, all   src/logic.c:25-670
 *
 * Parity pinning.  The reference itself cannot be compiled in this image: every file on the path
 * includes <php.h> / <Zend/zend_types.h> (e.g. arithmetics.c:1-3, ndarray.h:9) and PHP's headers
 * are absent; writing stand-ins for them is not allowed, so there is no oracle/_ref.  The oracle
 * is pinned against the reference's own known-answer tests instead: tests/golden/phpt_vectors.json
 * holds the inputs, calls and --EXPECT-- text of tests/math/002..044-*.phpt and
 * tests/linalg/001-ndarray-matmul.phpt, and tests/test_oracle_phpt.py replays them through this
 * file and compares the print_r text byte for byte.  Those KATs have <= 4 elements, so they pin
 * the scalar tails only; the AVX2 loop bodies (arithmetics.c:247-261 etc.) are restated from the
 * source and are NOT pinned by any reference-side vector ("parity unpinned" for: multiply's
 * fix_negative_zero body, mod's floor-based body, and every path with >= 8 elements).
 *
 * Third-party arithmetic: matmul numerics live in CBLAS (OpenBLAS, version unpinned by the
 * reference: config.m4:67-87).  oracle_set_blas() is pointed at an OpenBLAS found at run time
 * (scipy's bundled libscipy_openblas, 0.3.28 here); without one a plain blocked sgemm is used
 * and flagged through oracle_blas_kind().
 *
 * Build: gcc -O2 -mavx2 -mfma (numpower_amd/build.py).  -mfma matters: the reference is built
 * with `-mavx2 -march=native` (config.m4:36,50), and on an FMA-capable host gcc's default
 * -ffp-contract=fast fuses `a - floor(a/b)*b` and the rsqrt Newton step into fnmadd; the same
 * expressions are written here with the same shapes so that gcc makes the same choice.
 */
#include <dlfcn.h>
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

enum { O_ADD = 0, O_SUBTRACT, O_MULTIPLY, O_DIVIDE, O_MOD, O_POW, O_ARCTAN2,
       /* comparisons, src/logic.c:67-670 */
       O_EQUAL, O_NOT_EQUAL, O_GREATER, O_GREATER_EQUAL, O_LESS, O_LESS_EQUAL, O_MAXIMUM, O_MINIMUM };

enum {
    U_ABS = 0, U_SQRT, U_EXP, U_EXP2, U_EXPM1, U_LOG, U_LOG2, U_LOG10, U_LOG1P, U_LOGB,
    U_SIN, U_COS, U_TAN, U_ARCSIN, U_ARCCOS, U_ARCTAN, U_DEGREES, U_RADIANS,
    U_SINH, U_COSH, U_TANH, U_ARCSINH, U_ARCCOSH, U_ARCTANH,
    U_RINT, U_FIX, U_FLOOR, U_CEIL, U_TRUNC, U_SINC, U_NEGATE, U_SIGN,
    U_CLIP, U_ROUND, U_RSQRT, U_POSITIVE, U_RECIPROCAL, U_COUNT
};

enum { R_SUM = 0, R_PROD, R_MIN, R_MAX, R_MEAN };

static char g_error[256];
const char *oracle_last_error(void) { return g_error; }
static int fail(const char *msg) {
    snprintf(g_error, sizeof(g_error), "%s", msg);
    return -1;
}

/* ------------------------------------------------------------------------------------------
 * unary kernels: double_math.c:9-265
 * ------------------------------------------------------------------------------------------ */

static float u_rsqrt(float val) {   /* double_math.c:111-126 */
    const float threehalfs = 1.5F;
    float x2 = val * 0.5F;
    float y = val;
    uint32_t i;
    memcpy(&i, &y, 4);
    i = 0x5f3759df - (i >> 1);
    memcpy(&y, &i, 4);
    y = y * (threehalfs - (x2 * y * y));
    return y;
}

static float u_rint(float val) {    /* double_math.c:200-210 */
    float rounded = rintf(val);
    int floorInt = (int) floorf(val);
    if (rounded - (float) floorInt == 0.5f && ((int) rounded % 2 != 0)) rounded -= 1.0f;
    return rounded;
}

static float u_sinc(float val) {    /* double_math.c:228-235 */
    float pi = 3.1415927f;
    if (val == 0.0) val = 1.0e-20f;
    val = pi * val;
    return sinf(val) / val;
}

/* Domain errors: the reference prints and calls exit(1) for arccos/arccosh/arctanh outside
 * their domain (double_math.c:144-148,180-198).  The oracle reports them through *domain_error
 * and returns libm's value (NaN) instead of killing the test process. */
static float unary_apply(int op, float x, float p0, float p1, int *domain_error) {
    switch (op) {
        case U_ABS: return fabsf(x);
        case U_SQRT: return sqrtf(x);
        case U_EXP: return expf(x);
        case U_EXP2: return exp2f(x);
        case U_EXPM1: return expm1f(x);
        case U_LOG: return logf(x);
        case U_LOG2: return log2f(x);
        case U_LOG10: return log10f(x);
        case U_LOG1P: return log1pf(x);
        case U_LOGB: return logbf(x);
        case U_SIN: return sinf(x);
        case U_COS: return cosf(x);
        case U_TAN: return tanf(x);
        case U_ARCSIN: return asinf(x);
        case U_ARCCOS:
            if (x < -1.0 || x > 1.0) *domain_error = 1;
            return acosf(x);
        case U_ARCTAN: return atanf(x);
        case U_DEGREES: return (float) (x * (180.0 / 3.1415926535));   /* :156-158 */
        case U_RADIANS: return (float) (x * (3.1415926535 / 180.0));   /* :160-162 */
        case U_SINH: return sinhf(x);
        case U_COSH: return coshf(x);
        case U_TANH: return tanhf(x);
        case U_ARCSINH: return asinhf(x);
        case U_ARCCOSH:
            if (x < 1.0) *domain_error = 1;
            return acoshf(x);
        case U_ARCTANH:
            if (fabsf(x) == 1.0f || x < -1.0f || x > 1.0f) *domain_error = 1;
            return atanhf(x);
        case U_RINT: return u_rint(x);
        case U_FIX: return truncf(x);
        case U_FLOOR: return floorf(x);
        case U_CEIL: return ceilf(x);
        case U_TRUNC: return truncf(x);
        case U_SINC: return u_sinc(x);
        case U_NEGATE: return -x;
        case U_SIGN: return (float) ((x > 0.0f) - (x < 0.0f));                 /* :246-248 */
        case U_CLIP: return fminf(p1, fmaxf(x, p0));                           /* :250-252 */
        case U_ROUND: {                                                         /* :254-257 */
            float factor = powf(10, p0);
            return roundf(x * factor) / factor;
        }
        case U_RSQRT: return u_rsqrt(x);
        case U_POSITIVE: if (x < 0) return -x; return x;                       /* :241-244 */
        case U_RECIPROCAL: return 1 / x;
        default: return 0.0f;
    }
}

/* NDArray_Map / Map1F / Map2F (ndarray.c:682-744): out[i] = op(in[i]) in index order.
 * Returns the number of elements that hit a domain error (reference: exit(1) on the first). */
long oracle_map(int op, const float *in, float *out, long n, float p0, float p1) {
    long bad = 0;
    for (long i = 0; i < n; i++) {
        int d = 0;
        out[i] = unary_apply(op, in[i], p0, p1, &d);
        bad += d;
    }
    return bad;
}

/* ------------------------------------------------------------------------------------------
 * broadcast: ndarray.c:1124-1294
 * ------------------------------------------------------------------------------------------ */

static long numel(const int *shape, int ndim) {
    long n = 1;
    for (int i = 0; i < ndim; i++) n *= shape[i];
    return n;
}

/* NDArray_IsBroadcastable (ndarray.c:1124-1162) */
int oracle_is_broadcastable(const int *s1, int n1, const int *s2, int n2) {
    if (n1 == 1 && n2 > 1) return s1[0] == s2[n2 - 1];
    if (n1 > 1 && n2 == 1) return s2[0] == s1[n1 - 1];
    int maxd = n1 > n2 ? n1 : n2;
    for (int i = 0; i < maxd; i++) {
        int a = i < n1 ? s1[i] : 1, b = i < n2 ? s2[i] : 1;
        if (a != b && a != 1 && b != 1) return 0;
    }
    return 1;
}

/* NDArray_Broadcast(src -> dst shape) (ndarray.c:1172-1294), CPU branches.
 * Returns 0 and fills out (numel(dst) floats), 1 if src already has dst's shape (out untouched,
 * caller uses src), -1 on "Broadcast shape mismatch.", -2 if the reference would return a buffer
 * it never wrote (NDArray_EmptyLike left uninitialised: patterns outside the four it handles). */
int oracle_broadcast(const float *src, const int *ss, int sn, const int *ds, int dn, float *out) {
    if (sn == dn) {
        int all_equal = 1;
        for (int i = 0; i < sn; i++)
            if (ss[i] != ds[i]) all_equal = 0;
        if (all_equal) return 1;
    }
    if (!oracle_is_broadcastable(ss, sn, ds, dn)) {
        fail("Broadcast shape mismatch.");
        return -1;
    }
    long dnum = numel(ds, dn);
    int written = 0;
    if (sn == 0 && dn > 0) {   /* :1196-1200 */
        for (long i = 0; i < dnum; i++) out[i] = src[0];
        written = 1;
    }
    if (sn == 1 && dn > 1) {   /* :1202-1223 — copies dst.shape[-2] rows only */
        if (ss[0] == ds[dn - 2] || ss[0] == ds[dn - 1]) {
            float *p = out;
            for (int i = 0; i < ds[dn - 2]; i++) {
                memcpy(p, src, sizeof(float) * ds[dn - 1]);
                p += ss[0];
            }
            /* the reference fills exactly one (rows x cols) matrix; anything beyond it is
             * uninitialised memory */
            written = (dn == 2) ? 1 : -1;
        }
    }
    if (sn == 2 && dn == 2) {
        if (ss[0] == ds[0]) {   /* :1227-1247 column (R x 1) or single element */
            long snum = numel(ss, sn);
            for (int i = 0; i < ds[0]; i++)
                for (int j = 0; j < ds[1]; j++) out[(long) i * ds[1] + j] = (snum != 1) ? src[i] : src[0];
            written = 1;
            if (snum == 1) return 0;
        }
        if (ss[1] == ds[0]) {   /* :1273-1291 row memcpy; note the test is against dst.shape[0] */
            float *p = out;
            for (int i = 0; i < ds[0]; i++) {
                memcpy(p, src, sizeof(float) * ds[1]);
                p += ss[1];
            }
            written = 1;
        }
    }
    if (written == 1) return 0;
    fail("reference NDArray_Broadcast leaves the result uninitialised for this shape pair");
    return -2;
}

/* ------------------------------------------------------------------------------------------
 * binary ops: arithmetics.c:160-926
 * ------------------------------------------------------------------------------------------ */

static __m256 fix_negative_zero(__m256 vec) {   /* arithmetics.c:280-284 */
    __m256 zero = _mm256_set1_ps(-0.0f);
    __m256 mask = _mm256_cmp_ps(vec, zero, _CMP_EQ_OQ);
    return _mm256_blendv_ps(vec, zero, mask);
}

/* The hot loop shared by the six ops once both operands have numElements elements: 8-wide AVX2
 * body while i < loop_numel_a - 7, scalar tail for the rest (arithmetics.c:247-261, 397-414,
 * 527-541, 664-678, 788-802; pow has no AVX body :912-914). */
static void binary_loop(int op, const float *a, const float *b, float *r, long n, long loop_numel_a) {
    long i = 0;
    if (op != O_POW && op != O_ARCTAN2 && op != O_MAXIMUM && op != O_MINIMUM) {
        for (i = 0; i < loop_numel_a - 7; i += 8) {
            __m256 v1 = _mm256_loadu_ps(&a[i]);
            __m256 v2 = _mm256_loadu_ps(&b[i]);
            __m256 o;
            const __m256 ones = _mm256_set1_ps(1.0f);
            switch (op) {
                case O_ADD: o = _mm256_add_ps(v1, v2); break;
                case O_SUBTRACT: o = _mm256_sub_ps(v1, v2); break;
                case O_MULTIPLY: o = fix_negative_zero(_mm256_mul_ps(v1, v2)); break;
                case O_DIVIDE: o = _mm256_div_ps(v1, v2); break;
                /* comparisons: mask AND 1.0f / blend(0, 1, mask) (logic.c:134-138, 234-237, 339, 440, 541, 642) */
                case O_EQUAL: o = _mm256_and_ps(_mm256_cmp_ps(v1, v2, _CMP_EQ_OQ), ones); break;
                case O_NOT_EQUAL: o = _mm256_and_ps(_mm256_cmp_ps(v1, v2, _CMP_NEQ_OQ), ones); break;
                case O_GREATER: o = _mm256_and_ps(_mm256_cmp_ps(v1, v2, _CMP_GT_OQ), ones); break;
                case O_GREATER_EQUAL: o = _mm256_and_ps(_mm256_cmp_ps(v1, v2, _CMP_GE_OS), ones); break;
                case O_LESS: o = _mm256_and_ps(_mm256_cmp_ps(v1, v2, _CMP_LT_OQ), ones); break;
                case O_LESS_EQUAL: o = _mm256_and_ps(_mm256_cmp_ps(v1, v2, _CMP_LE_OQ), ones); break;
                default:   /* O_MOD, arithmetics.c:794 */
                    o = _mm256_sub_ps(v1, _mm256_mul_ps(_mm256_floor_ps(_mm256_div_ps(v1, v2)), v2));
                    break;
            }
            _mm256_storeu_ps(&r[i], o);
        }
    }
    for (; i < n; i++) {
        switch (op) {
            case O_ADD: r[i] = a[i] + b[i]; break;
            case O_SUBTRACT: r[i] = a[i] - b[i]; break;
            case O_MULTIPLY:
                r[i] = a[i] * b[i];
                if (r[i] == 0.0f && signbit(r[i])) r[i] = 0.0f;   /* :410-412 */
                break;
            case O_DIVIDE: r[i] = a[i] / b[i]; break;
            case O_MOD: r[i] = fmodf(a[i], b[i]); break;
            case O_POW: r[i] = powf(a[i], b[i]); break;
            /* scalar tails of the comparisons: logic.c:146, 245, 350, 451, 552, 655 */
            case O_EQUAL: r[i] = (fabsf(a[i] - b[i]) <= 0.0000001f) ? 1.0f : 0.0f; break;
            case O_NOT_EQUAL: r[i] = (fabsf(a[i] - b[i]) <= 0.0000001f) ? 0.0f : 1.0f; break;
            case O_GREATER: r[i] = a[i] > b[i] ? 1.0f : 0.0f; break;
            case O_GREATER_EQUAL: r[i] = a[i] >= b[i] ? 1.0f : 0.0f; break;
            case O_LESS: r[i] = a[i] < b[i] ? 1.0f : 0.0f; break;
            case O_LESS_EQUAL: r[i] = a[i] <= b[i] ? 1.0f : 0.0f; break;
            /* NDArray_Maximum / NDArray_Minimum: plain scalar loops over fmaxf / fminf (ndarray.c:880-882,
             * 923-925; the loop bound there is numel(a) BEFORE the broadcast — restated over all elements) */
            case O_MAXIMUM: r[i] = fmaxf(a[i], b[i]); break;
            case O_MINIMUM: r[i] = fminf(a[i], b[i]); break;
            default: r[i] = atan2f(a[i], b[i]); break;   /* float_arctan2 via Map1ND, ndarray.c:716 */
        }
    }
}

/* NDArray_{Add,Subtract,Multiply,Divide,Mod,Pow}_Float.  Inputs are (data, shape, ndim); the
 * result is malloc'd into *out with its shape in out_shape/out_ndim (caller frees *out).
 * Returns 0, or -1 with oracle_last_error() = the message the reference throws. */
int oracle_binary(int op, const float *a, const int *as, int an, const float *b, const int *bs, int bn,
                  float **out, int *out_shape, int *out_ndim) {
    *out = NULL;
    /* 0-d x 0-d short cuts: multiply (arithmetics.c:302-316) and divide (:575-580) */
    if (an == 0 && bn == 0 && (op == O_MULTIPLY || op == O_DIVIDE)) {
        float *r = (float *) malloc(sizeof(float));
        r[0] = (op == O_MULTIPLY) ? a[0] * b[0] : a[0] / b[0];
        *out = r;
        *out_ndim = 0;
        return 0;
    }
    float *a_exp = NULL, *b_exp = NULL;
    const float *ad = a, *bd = b;
    const int *ash = as, *bsh = bs;
    int and_ = an, bnd = bn;
    /* scalar expand = Zeros + Fill to the other operand's shape (:169-181) */
    if (an == 0 && bn > 0) {
        long n = numel(bs, bn);
        a_exp = (float *) malloc(sizeof(float) * (n > 0 ? n : 1));
        for (long i = 0; i < n; i++) a_exp[i] = a[0];
        ad = a_exp; ash = bs; and_ = bn;
    } else if (bn == 0 && an > 0) {
        long n = numel(as, an);
        b_exp = (float *) malloc(sizeof(float) * (n > 0 ? n : 1));
        for (long i = 0; i < n; i++) b_exp[i] = b[0];
        bd = b_exp; bsh = as; bnd = an;
    }
    long na = numel(ash, and_), nb = numel(bsh, bnd);
    float *bro = NULL;
    const float *a_broad = ad, *b_broad = bd;
    const int *rs = ash;
    int rn = and_;
    int rc = 1;
    if (na < nb) {          /* :186-189 */
        bro = (float *) malloc(sizeof(float) * (nb > 0 ? nb : 1));
        rc = oracle_broadcast(ad, ash, and_, bsh, bnd, bro);
        a_broad = (rc == 1) ? ad : bro;
        rs = bsh; rn = bnd;
    } else if (nb < na) {   /* :190-193 */
        bro = (float *) malloc(sizeof(float) * (na > 0 ? na : 1));
        rc = oracle_broadcast(bd, bsh, bnd, ash, and_, bro);
        b_broad = (rc == 1) ? bd : bro;
    }
    if (rc < 0) {
        if (rc == -1) fail("Can't broadcast arrays.");   /* :199-202 after "Broadcast shape mismatch." */
        free(a_exp); free(b_exp); free(bro);
        return -1;
    }
    long n = numel(rs, rn);
    float *r = (float *) malloc(sizeof(float) * (n > 0 ? n : 1));
    /* loop bound of the AVX body is NDArray_NUMELEMENTS(a) with `a` = the (possibly scalar-
     * expanded, NOT broadcast) first operand (:251; same in Less/Equal, logic.c:228,535);
     * Greater/LessEqual/GreaterEqual/NotEqual loop over the broadcast operand (logic.c:128,333,434,636).
     * NOTE Less/Equal also INDEX the un-broadcast operands (logic.c:230-231,537-538), which reads
     * past the smaller buffer when one operand was broadcast: undefined in the reference; the
     * restatement uses the broadcast data there. */
    long bound = na;
    if (op == O_GREATER || op == O_LESS_EQUAL || op == O_GREATER_EQUAL || op == O_NOT_EQUAL) bound = n;
    binary_loop(op, a_broad, b_broad, r, n, bound);
    for (int i = 0; i < rn; i++) out_shape[i] = rs[i];
    *out_ndim = rn;
    *out = r;
    free(a_exp); free(b_exp); free(bro);
    return 0;
}

void oracle_free(void *p) { free(p); }

/* ------------------------------------------------------------------------------------------
 * full reductions
 * ------------------------------------------------------------------------------------------ */

float oracle_sum(const float *a, long n) {      /* NDArray_Sum_Float, arithmetics.c:58-71 */
    float value = 0;
    for (long i = 0; i < n; i++) value += a[i];
    return value;
}
float oracle_prod(const float *a, long n) {     /* NDArray_Float_Prod, arithmetics.c:36-49 */
    float value = 1;
    for (long i = 0; i < n; i++) value *= a[i];
    return value;
}
float oracle_min(const float *a, long n) {      /* NDArray_Min, ndarray.c:752-772 */
    float m = a[0];
    for (long i = 1; i < n; i++)
        if (a[i] < m) m = a[i];
    return m;
}
float oracle_max(const float *a, long n) {      /* NDArray_Max, ndarray.c:939-959 */
    float m = a[0];
    for (long i = 1; i < n; i++)
        if (a[i] > m) m = a[i];
    return m;
}
/* float_argmax / float_argmin (calculation.c:9-72) applied along the middle axis of an
 * outer x len x inner view (NDArray_ArgMinMaxCommon moves the axis last with a transposed copy and
 * walks contiguous rows, calculation.c:97-181 — same element order per row). */
void oracle_argreduce(int is_max, const float *in, long outer, long len, long inner, float *out) {
    for (long o = 0; o < outer; o++)
        for (long j = 0; j < inner; j++) {
            const float *ip = in + o * len * inner + j;
            float mp = *ip;
            float ind = 0;
            if (!isnan(mp)) {
                for (long i = 1; i < len; i++) {
                    const float v = ip[i * inner];
                    if (is_max ? (v > mp) : !(mp <= v)) {
                        mp = v;
                        ind = (float) i;
                        if (isnan(mp)) break;
                    }
                }
            }
            out[o * inner + j] = ind;
        }
}

/* NDArray_Transpose (manipulation.c:68-130): permute shape and byte strides of a copy, then
 * NDArray_ToContiguous walks the permuted view element by element (manipulation.c:381-421).
 * perm == NULL reverses the axes.  Returns 0, or -1 with the reference's message. */
int oracle_transpose(const float *in, const int *shape, int ndim, const int *perm, float *out, int *out_shape) {
    int p[32], seen[32];
    long in_stride[32];
    if (ndim > 32) return fail("axes don't match array");
    for (int i = 0; i < ndim; i++) seen[i] = 0;
    for (int i = 0; i < ndim; i++) {
        int axis = perm ? perm[i] : ndim - 1 - i;
        if (axis < 0) axis += ndim;
        if (axis < 0 || axis >= ndim) return fail("axes don't match array");
        if (seen[axis]) return fail("repeated axis in transpose");
        seen[axis] = 1;
        p[i] = axis;
    }
    long s = 1;
    for (int i = ndim - 1; i >= 0; i--) {
        in_stride[i] = s;
        s *= shape[i];
    }
    long n = s;
    for (int i = 0; i < ndim; i++) out_shape[i] = shape[p[i]];
    int idx[32];
    for (int i = 0; i < ndim; i++) idx[i] = 0;
    for (long o = 0; o < n; o++) {
        long off = 0;
        for (int i = 0; i < ndim; i++) off += idx[i] * in_stride[p[i]];
        out[o] = in[off];
        for (int i = ndim - 1; i >= 0; i--) {   /* odometer over the output index */
            if (++idx[i] < out_shape[i]) break;
            idx[i] = 0;
        }
    }
    return 0;
}

/* NDArray_Variance (statistics.c:117-130): mean, Subtract, Abs, Pow(…, 2), Sum / n — every step
 * a full pass with fp32 rounding, restated in the same order. */
float oracle_variance(const float *a, long n) {
    float mean = oracle_sum(a, n) / n;
    float value = 0;
    for (long i = 0; i < n; i++) value += powf(fabsf(a[i] - mean), 2.0f);
    return value / n;
}
/* NDArray_Std (statistics.c:88-108) */
float oracle_std(const float *a, long n) {
    float mean = oracle_sum(a, n) / n;
    float sum = 0.0f;
    for (long i = 0; i < n; i++) sum += powf(a[i] - mean, 2);
    return sqrtf(sum / (float) n);
}
/* NDArray_Average with weights (statistics.c:145-151): Multiply_Float then two Sum_Float */
float oracle_average_weighted(const float *a, const float *w, long n) {
    float s_w = oracle_sum(w, n);
    float s_aw = 0;
    for (long i = 0; i < n; i++) {
        float p = a[i] * w[i];   /* the -0.0 fix of Multiply_Float does not change a sum */
        s_aw += p;
    }
    return s_aw / s_w;
}

/* NDArray_All (logic.c:25-58), including the body's `mask != 0x0F` test on an 8-lane mask */
float oracle_all(const float *array, long n) {
    long i;
    __m256 zero = _mm256_set1_ps(0.0f);
    for (i = 0; i < n - 7; i += 8) {
        __m256 elements = _mm256_loadu_ps(&array[i]);
        __m256 comparison = _mm256_cmp_ps(elements, zero, _CMP_NEQ_OQ);
        int mask = _mm256_movemask_ps(comparison);
        if (mask != 0x0F) return 0;
    }
    for (; i < n; i++)
        if (array[i] == 0.0) return 0;
    return 1;
}

/* NDArray_Arange (initializers.c:836-839): x[0] = (float)start; x[i] = x[i-1] + step with the sum
 * formed in double (float + double) and stored as float */
void oracle_arange(float *out, double start, double step, long n) {
    if (n <= 0) return;
    out[0] = (float)start;
    for (long i = 1; i < n; i++) out[i] = out[i - 1] + step;
}

/* compare_ndarrays (logic.c:678-693), CPU branch: 1 unless some a[i] != b[i] (NaN != NaN) */
int oracle_array_equal(const float *a, const float *b, long n) {
    int same = 1;
    for (long i = 0; i < n; i++)
        if (a[i] != b[i]) same = 0;
    return same;
}

/* float_allclose (logic.c:719-738) with the element index the loop means: the reference computes
 * index = i*sizeof(float) + i*strides[0]/sizeof(float) (= 5i for a contiguous vector), i.e. it
 * reads past the buffer for every i > 0 — its own KAT (tests/logic/002) only survives because the
 * first element already decides / the PHP method short-cuts identical handles.  Restated per
 * element i; the tolerance is one fused multiply-add, as gcc -march=native contracts
 * `atol + rtol * fabsf(b)`. */
int oracle_allclose(const float *a, const float *b, long n, float rtol, float atol) {
    for (long i = 0; i < n; i++) {
        float diff = fabsf(a[i] - b[i]);
        float tolerance = fmaf(rtol, fabsf(b[i]), atol);
        if (diff > tolerance) return 0;
    }
    return 1;
}

/* Comparator of both qsort() calls (arithmetics.c:105-109, statistics.c:8-12). */
static int cmp_float(const void *a, const void *b) {
    float fa = *(const float *)a, fb = *(const float *)b;
    return (fa > fb) - (fa < fb);
}

/* calculate_median (arithmetics.c:111-138): sort a copy; even count -> (t[n/2-1] + t[n/2]) / 2.0f,
 * odd -> t[n/2].  out2 (optional) receives the two order statistics the value was formed from. */
float oracle_median(const float *a, long n, float *out2) {
    float *t = (float *)malloc((size_t)n * sizeof(float));
    memcpy(t, a, (size_t)n * sizeof(float));
    qsort(t, (size_t)n, sizeof(float), cmp_float);
    float lo = (n % 2 == 0) ? t[n / 2 - 1] : t[n / 2], hi = t[n / 2];
    float median = (n % 2 == 0) ? (lo + hi) / 2.0f : hi;
    if (out2) { out2[0] = lo; out2[1] = hi; }
    free(t);
    return median;
}

/* calculate_quantile (statistics.c:14-50): index = (float)(n-1) * q; lower = (int)index; upper = lower+1;
 * weight = index - (float)lower; (1 - weight) * t[lower] + weight * t[upper].  The reference reads
 * t[n] (one past its malloc) when q == 1, and multiplies it by weight 0: restated with the last
 * element there, which gives the same value whenever that stray float is finite.  The arithmetic
 * is left to the compiler exactly as in the reference (gcc -mfma contracts it). */
float oracle_quantile(const float *a, long n, float quantile, float *out2) {
    float *t = (float *)malloc((size_t)n * sizeof(float));
    memcpy(t, a, (size_t)n * sizeof(float));
    qsort(t, (size_t)n, sizeof(float), cmp_float);
    float index = (float)(n - 1) * quantile;
    int lower_index = (int)index;
    int upper_index = lower_index + 1;
    if (upper_index > n - 1) upper_index = (int)(n - 1);
    float weight = index - (float)lower_index;
    float lower_value = t[lower_index];
    float upper_value = t[upper_index];
    float quantile_value = (1 - weight) * lower_value + weight * upper_value;
    if (out2) { out2[0] = lower_value; out2[1] = upper_value; }
    free(t);
    return quantile_value;
}

/* NDArray::mean without axis: NDArray_Sum_Float(nda) / NDArray_NUMELEMENTS(nda) (numpower.c:2659) */
float oracle_mean(const float *a, long n) { return oracle_sum(a, n) / n; }

/* ------------------------------------------------------------------------------------------
 * axis reductions: reduce(array, &axis, NDArray_Add_Float | NDArray_Multiply_Float)
 * ------------------------------------------------------------------------------------------ */

/* _reduce (ndarray.c:394-429): for every index of the axes in front of `axis`, copy the first
 * slice, then fold each further slice with apply_reduce = operation(rtn, slice) + memcpy back
 * (ndarray.c:358-368).  A slice has `inner` elements and ndim - axis - 1 dimensions; 0-d slices
 * take the 0-d x 0-d short cut of Multiply_Float.  op: R_SUM / R_PROD / R_MEAN
 * (mean = reduce(Add) then Divide_Float by CreateFromLongScalar(shape[axis]), numpower.c:2660-2670). */
int oracle_reduce_axis(int op, const float *in, const int *shape, int ndim, int axis, float *out) {
    if (axis >= ndim || axis < 0) {
        snprintf(g_error, sizeof(g_error), "axis %d is out of bounds for array of dimension %d", axis, ndim);
        return -1;
    }
    long outer = 1, inner = 1;
    for (int i = 0; i < axis; i++) outer *= shape[i];
    for (int i = axis + 1; i < ndim; i++) inner *= shape[i];
    const long len = shape[axis];
    const int slice_ndim = ndim - axis - 1;
    const int bop = (op == R_PROD) ? O_MULTIPLY : O_ADD;
    float *tmp = (float *) malloc(sizeof(float) * (inner > 0 ? inner : 1));
    for (long o = 0; o < outer; o++) {
        float *rtn = out + o * inner;
        const float *base = in + o * len * inner;
        memcpy(rtn, base, sizeof(float) * inner);
        for (long s = 1; s < len; s++) {
            const float *slice = base + s * inner;
            if (slice_ndim == 0 && bop == O_MULTIPLY)
                tmp[0] = rtn[0] * slice[0];
            else
                binary_loop(bop, rtn, slice, tmp, inner, inner);
            memcpy(rtn, tmp, sizeof(float) * inner);
        }
    }
    free(tmp);
    if (op == R_MEAN) {
        /* Divide_Float(sum, scalar): scalar expand then AVX div body + tail = IEEE division */
        const float d = (float) (long) len;
        for (long i = 0; i < outer * inner; i++) out[i] = out[i] / d;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * matmul / matrix.vector: linalg.c:44-82, 367-386
 * ------------------------------------------------------------------------------------------ */

typedef void (*sgemm_fn)(int, int, int, int, int, int, float, const float *, int, const float *, int,
                         float, float *, int);
typedef void (*sgemv_fn)(int, int, int, int, float, const float *, int, const float *, int, float,
                         float *, int);
typedef void (*setthreads_fn)(int);
static sgemm_fn g_sgemm = NULL;
static sgemv_fn g_sgemv = NULL;
static void *g_blas = NULL;

/* Point the oracle at a CBLAS shared library (OpenBLAS).  `prefix` is the symbol prefix
 * ("" for a system libopenblas, "scipy_" for the wheels' bundled one).  Returns 0 on success. */
int oracle_set_blas(const char *path, const char *prefix, int threads) {
    char name[128];
    void *h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(dlerror());
    snprintf(name, sizeof(name), "%scblas_sgemm", prefix);
    sgemm_fn f = (sgemm_fn) dlsym(h, name);
    snprintf(name, sizeof(name), "%scblas_sgemv", prefix);
    sgemv_fn v = (sgemv_fn) dlsym(h, name);
    if (!f || !v) return fail("cblas_sgemm/cblas_sgemv not found in BLAS library");
    if (threads > 0) {
        snprintf(name, sizeof(name), "%sopenblas_set_num_threads", prefix);
        setthreads_fn st = (setthreads_fn) dlsym(h, name);
        if (st) st(threads);
    }
    g_blas = h;
    g_sgemm = f;
    g_sgemv = v;
    return 0;
}

/* 1 = external CBLAS (the reference's own back end), 0 = built-in plain sgemm */
int oracle_blas_kind(void) { return g_sgemm != NULL; }

static void plain_sgemm(int M, int N, int K, const float *A, const float *B, float *C) {
    /* i-k-j loop, row-major, fp32 accumulation; only used when no CBLAS could be found */
    for (long i = 0; i < M; i++) {
        float *c = C + i * N;
        for (long j = 0; j < N; j++) c[j] = 0.0f;
        for (long k = 0; k < K; k++) {
            const float a = A[i * K + k];
            const float *b = B + k * N;
            for (long j = 0; j < N; j++) c[j] += a * b[j];
        }
    }
}

/* NDArray_FMatmul CPU branch (linalg.c:75-79):
 * cblas_sgemm(CblasRowMajor, CblasNoTrans, CblasNoTrans, M, N, K, 1, A, K, B, N, 0, C, N) */
void oracle_matmul(int M, int N, int K, const float *A, const float *B, float *C) {
    if (g_sgemm)
        g_sgemm(101 /*RowMajor*/, 111 /*NoTrans*/, 111, M, N, K, 1.0f, A, K, B, N, 0.0f, C, N);
    else
        plain_sgemm(M, N, K, A, B, C);
}

/* NDArray_Dot matrix.vector CPU branch (linalg.c:382-383): cblas_sgemv(RowMajor, NoTrans, …) */
void oracle_matvec(int M, int N, const float *A, const float *x, float *y) {
    if (g_sgemv) {
        g_sgemv(101, 111, M, N, 1.0f, A, N, x, 1, 0.0f, y, 1);
        return;
    }
    for (long i = 0; i < M; i++) {
        float acc = 0.0f;
        for (long j = 0; j < N; j++) acc += A[i * N + j] * x[j];
        y[i] = acc;
    }
}

/* NDArray_Matmul argument checks (linalg.c:216-245); returns 0 or -1 with the reference's message */
int oracle_matmul_check(const int *as, int an, const int *bs, int bn) {
    if (an != bn) return fail("Arrays must have the same shape. Broadcasting not implemented.");
    if (an == 0 || an == 1) return 0;
    if (as[an - 1] != bs[bn - 2]) return fail("Shape mismatch for matmul. cols(a) != rows(b)");
    if (an > 2 && bn > 2) return fail("Stack of matrices not allowed");
    return 0;
}
