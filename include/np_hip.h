/*
 * np_hip.h — C ABI of the MI355X (gfx950) device back end for NumPower's fp32 hot path.
 *
 * This is the drop-in boundary: plain C, raw device pointers, size_t counts, int status.
 * It replaces the reference's inner (C <-> device back end) interface, i.e.
 *   - src/gpu_alloc.h:8-15            (vmalloc / vfree / vmemcpy* / vmemcheck / NDArray_VFLOAT)
 *   - src/ndmath/cuda/cuda_math.h:14-79 (cuda_*_float, cuda_float_*, cuda_sum/prod/min/max_float,
 *                                        cuda_fill_float, cuda_float_multiply_matrix_vector)
 *   - the cublasSgemm call in src/ndmath/linalg.c:55-71
 * Every entry point cites the reference symbol it stands in for.  All paths are relative to
 * the reference tree (NumPower/numpower @ 2024_08_07).
 *
 * Conventions
 *   - every function returns NP_OK (0) or a negative np_status; np_last_error() returns the
 *     message of the last failure on the calling thread.
 *   - pointers are DEVICE pointers unless the parameter name starts with `host`.
 *   - all arrays are contiguous C-order fp32 (the reference's only dtype on this path,
 *     src/types.h:5); the ops never look at strides (same contract as arithmetics.c).
 *   - work is enqueued on the library stream (np_get_stream/np_set_stream) and is asynchronous;
 *     np_sync(), np_memcpy_d2h() and the host_out reductions are the only blocking calls.
 *     (The reference synchronises after every launch, cuda_math.cu:1104-1109; results are
 *     identical, only the blocking point moves to the read-back.)
 *   - no torch / PHP / Zend types anywhere in this file.
 *   - threads: the library is one device + one stream per process, like the reference (PHP NTS, one
 *     request = one thread).  Calls may come from several host threads: the pool and the host-result
 *     entry points (np_reduce_all, np_all, np_count_mismatch, np_moments, np_weighted_sums,
 *     np_order_stat, np_fused_chain_reduce) serialise internally, np_last_error() is per thread.
 *     Every np_* call leaves the CALLING THREAD's current HIP device set to the library's device
 *     (np_init / np_set_device) — the effect cudaSetDevice has in the reference (numpower.c:633);
 *     a host that keeps another device current for its own work must re-select it afterwards.
 */
#ifndef NUMPOWER_AMD_NP_HIP_H
#define NUMPOWER_AMD_NP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum np_status {
    NP_OK = 0,
    NP_ERR_INVALID = -1,   /* bad argument (null pointer, unknown op, bad shape)          */
    NP_ERR_ALLOC = -2,     /* "device memory allocation failed" (gpu_alloc.c:15)          */
    NP_ERR_DEVICE = -3,    /* HIP runtime error (message in np_last_error)                */
    NP_ERR_NODEVICE = -4   /* "No GPU device available" (numpower.c:525)                  */
} np_status;

/* ---- runtime ------------------------------------------------------------------------ */

/* Select the device and create the library stream.  Replaces the implicit CUDA context
 * creation of the reference; np_set_device mirrors NDArray::setDevice -> cudaSetDevice
 * (numpower.c:615-635): it changes where NEW buffers and NEW work go and tears nothing down — the
 * device used before keeps its stream (work in flight completes), its cached blocks and its live
 * buffers, which can be freed at any time and used again after switching back. */
int np_init(int device);
int np_set_device(int device);
int np_device_count(int *host_count);
int np_sync(void);
/* Device errors.  A kernel whose device-side wait gives up (see "Device-side waits" below) raises a bit in the error word of
 * THE DEVICE IT RUNS ON.  From then on np_sync, np_memcpy_d2h, every host-result call and every np_comm_* call on that device
 * return NP_ERR_DEVICE — every time, whichever thread asks, so nobody's sync "eats" the error of the launch's owner — and the
 * same calls on another device (np_set_device) are not affected.  np_clear_device_error() is the acknowledgement: it waits
 * for everything the current device was given (all streams), puts the library's device-side bookkeeping back in order,
 * clears the word and hands back the bits that were up (1 = a stream-ordering wait of np_comm, 2 = a GEMM workgroup waiting
 * for its siblings' partial tiles; 0 = there was nothing to acknowledge).  Results produced between the failed launch and
 * the acknowledgement must be discarded.  While the device's communicator still has a transfer in flight that waits for a peer
 * (after 10 s of grace, or np_comm_set_wait_limit's seconds if fewer) it returns NP_ERR_DEVICE and clears nothing — the collective library's kernel does not end by itself
 * when a rank has died, and waiting for the whole device would never return: np_comm_destroy() first (it aborts the
 * communicator), then acknowledge.  (The reference ignores device errors altogether: cuda_math.cu never checks a launch.) */
int np_clear_device_error(unsigned *host_bits /* may be NULL */);
const char *np_last_error(void);
/* Library version string, e.g. "numpower_amd 0.1 gfx950". */
const char *np_version(void);

/* Stream plumbing: the library launches on one stream.  By default it owns a non-blocking
 * stream; a caller that already has one (e.g. torch's current stream, passed as the raw
 * hipStream_t) can hand it in.  Pass NULL to go back to the library-owned stream. */
int np_set_stream(void *hip_stream);
void *np_get_stream(void);

/* Event timing on the library stream (used by bench.py; hipEvent based). */
int np_timer_create(void **timer);
int np_timer_start(void *timer);
int np_timer_stop(void *timer);
int np_timer_elapsed_ms(void *timer, float *host_ms);   /* blocks until stop event done */
int np_timer_destroy(void *timer);

/* Launch-bound sequences as HIP graphs: capture what the library enqueues between begin and end,
 * replay it with one launch.  The reference synchronises after every kernel (cuda_math.cu:1104-1109),
 * so a chain of small ops costs a launch + sync each; a captured chain costs one graph launch.
 * Inside the captured region: no host-result calls (np_reduce_all, np_read_float, np_memcpy_*2h/h2d
 * ...), and warm the sequence up once first so the pool does not have to hipMalloc. */
int np_graph_begin(void);
int np_graph_end(void **graph_exec);
int np_graph_launch(void *graph_exec);
int np_graph_destroy(void *graph_exec);

/* ---- device-buffer layer (replaces src/gpu_alloc.c) ---------------------------------- */

/* vmalloc (gpu_alloc.c:11-17).  size_t instead of the reference's `unsigned int` (4 GiB cap).
 * Served from a caching sub-allocator (freed blocks are kept per size class and reused, so the
 * per-op result allocation of arithmetics.c:211-231 does not reach hipMalloc in steady state).
 * Bumps the live-allocation counter that np_live_allocs() reports. */
int np_malloc(void **dev_ptr, size_t bytes);
/* vfree (gpu_alloc.c:30-33). */
int np_free(void *dev_ptr);
/* vmemcheck (gpu_alloc.c:36-40): number of live np_malloc blocks (NDARRAY_VCHECK parity). */
long np_live_allocs(void);
/* Return all cached (free) blocks to the driver; returns bytes released through *host_bytes. */
int np_pool_trim(size_t *host_bytes);
/* Bytes currently held by the pool (live + cached). */
size_t np_pool_reserved_bytes(void);

/* vmemcpyh2d (gpu_alloc.c:25-27) / the cudaMemcpy H2D of NDArray_ToGPU (ndarray.c:1054). */
int np_memcpy_h2d(void *dev_dst, const void *host_src, size_t bytes);
/* The cudaMemcpy D2H of NDArray_ToCPU (ndarray.c:1090); blocks until the data is on the host. */
int np_memcpy_d2h(void *host_dst, const void *dev_src, size_t bytes);
/* vmemcpyd2d (gpu_alloc.c:20-22).  NOTE argument order here is (dst, src). */
int np_memcpy_d2d(void *dev_dst, const void *dev_src, size_t bytes);
/* cudaMemset(…, 0, …) of NDArray_Zeros (initializers.c:437-446). */
int np_memset0(void *dev_ptr, size_t bytes);
/* cuda_fill_float (cuda_math.cu:829,912). */
int np_fill(float *dev_ptr, float value, size_t n);
/* NDArray_VFLOAT / NDArray_VFLOATF_I (gpu_alloc.c:43-54): one float back to the host. */
int np_read_float(const float *dev_ptr, size_t index, float *host_out);

/* ---- binary elementwise with fused broadcast ------------------------------------------ */

/* Ops of NDArray_{Add,Subtract,Multiply,Divide,Mod,Pow}_Float (arithmetics.c:160-926) and
 * float_arctan2 (double_math.c:259-261, cuda_float_arctan2 cuda_math.cu:489,1224). */
typedef enum np_binary_op {
    NP_ADD = 0, NP_SUBTRACT = 1, NP_MULTIPLY = 2, NP_DIVIDE = 3, NP_MOD = 4, NP_POW = 5,
    NP_ARCTAN2 = 6,
    /* comparison ops of src/logic.c:67-670: 1.0f where true, 0.0f elsewhere.
     * Ordered compares (NaN -> 0).  EQUAL / NOT_EQUAL: the reference's AVX2 body compares exactly
     * (_CMP_EQ_OQ / _CMP_NEQ_OQ, logic.c:541,642), its scalar tail and its CUDA kernels use
     * |a-b| <= 1e-7 (logic.c:552,655, cuda_math.cu:243); NP_QUIRK_AVX_BODY selects the CPU
     * body/tail split, flags = 0 the tolerance form everywhere. */
    NP_EQUAL = 7, NP_NOT_EQUAL = 8, NP_GREATER = 9, NP_GREATER_EQUAL = 10, NP_LESS = 11,
    NP_LESS_EQUAL = 12,
    /* NDArray_Maximum / NDArray_Minimum (ndarray.c:853-931): fmaxf / fminf per element, with glibc's
     * rules — a NaN loses to a number, and when the operands compare equal (+0 / -0) the FIRST one is
     * returned.  (The reference refuses GPU arrays here and its CPU loop stops at numel(a).) */
    NP_MAXIMUM = 13, NP_MINIMUM = 14,
    NP_BINARY_OP_COUNT
} np_binary_op;

/* How an operand maps onto the rows x cols output.  These are exactly the patterns
 * NDArray_Broadcast materialises (ndarray.c:1196-1291); here they are index arithmetic
 * inside the kernel, no temporary is written. */
typedef enum np_operand_kind {
    NP_FULL = 0,    /* rows*cols elements                                                   */
    NP_SCALAR = 1,  /* 1 element        (0-d scalar expand, arithmetics.c:169-181)          */
    NP_ROW = 2,     /* cols elements    (1-D / 1xC -> every row, ndarray.c:1202-1223,1273)  */
    NP_COL = 3,     /* rows elements    (Rx1 -> every column, ndarray.c:1226-1272)          */
    NP_HOST_SCALAR = 4  /* like NP_SCALAR but the pointer is a HOST pointer to one float: the
                         * `$gpu_array + 2.0` case, where the 0-d operand lives on the CPU
                         * (arithmetics.c:163 exempts 0-d operands from the device check)      */
} np_operand_kind;

/* Result-visible CPU-path quirks of the reference that the kernel can reproduce so that GPU
 * results equal the reference's AVX2 CPU results element for element:
 *   NP_QUIRK_AVX_BODY  — elements with index < body_end behave like the AVX2 loop body,
 *                        the rest like the scalar tail:
 *       multiply: body turns every zero product into -0.0f (fix_negative_zero,
 *                 arithmetics.c:280-284,403), tail turns -0.0f into +0.0f (:410-412)
 *       mod:      body is a - floor(a/b)*b (arithmetics.c:794), tail is fmodf (:800)
 *   body_end is what the loop `for (i = 0; i < numel(a) - 7; i += 8)` covers, see
 *   np_avx_body_end().  flags = 0 gives plain IEEE multiply / C fmodf for every element
 *   (what the reference's own CUDA kernels do, cuda_math.cu:617-631). */
#define NP_QUIRK_AVX_BODY 1u

/* Number of leading elements covered by the reference's 8-wide AVX2 loop when the loop bound
 * is `numel_a - 7` (arithmetics.c:251): 0 if numel_a < 8, else 8*ceil((numel_a-7)/8). */
size_t np_avx_body_end(size_t numel_a);

/* out[r*cols + c] = a[...] op b[...];  replaces cuda_{add,subtract,multiply,divide,mod,pow}_float
 * (cuda_math.cu:1064-1109) + the Zeros/Fill/Broadcast temporaries in front of them. */
int np_binary(int op, const float *a, int a_kind, const float *b, int b_kind, float *out,
              size_t rows, size_t cols, unsigned flags, size_t body_end);

/* ---- unary elementwise ------------------------------------------------------------------ */

/* The float_* kernels of src/ndmath/double_math.c:9-265 (order follows double_math.h:7-44). */
typedef enum np_unary_op {
    NP_ABS = 0, NP_SQRT, NP_EXP, NP_EXP2, NP_EXPM1, NP_LOG, NP_LOG2, NP_LOG10, NP_LOG1P, NP_LOGB,
    NP_SIN, NP_COS, NP_TAN, NP_ARCSIN, NP_ARCCOS, NP_ARCTAN, NP_DEGREES, NP_RADIANS,
    NP_SINH, NP_COSH, NP_TANH, NP_ARCSINH, NP_ARCCOSH, NP_ARCTANH,
    NP_RINT, NP_FIX, NP_FLOOR, NP_CEIL, NP_TRUNC, NP_SINC, NP_NEGATE, NP_SIGN,
    NP_CLIP,        /* p0 = min, p1 = max            (float_clip, double_math.c:250-252)    */
    NP_ROUND,       /* p0 = decimals                 (float_round, double_math.c:254-257)   */
    NP_RSQRT,       /* 0x5f3759df bit hack + 1 Newton step (float_rsqrt, :111-126)          */
    NP_POSITIVE, NP_RECIPROCAL,
    NP_UNARY_OP_COUNT
} np_unary_op;

/* out[i] = f(in[i]); in == out is allowed.  Replaces NDArrayMathGPU_ElementWise{,1F,2F}
 * (cuda_math.cu:1532-1558) and the unary kernels behind them, without the preceding
 * NDArray_Copy (one 8 B/elem pass instead of copy + in-place = 16 B/elem). */
int np_unary(int op, const float *in, float *out, size_t n, float p0, float p1);

/* ---- fused elementwise chains --------------------------------------- */

/* One pass over HBM for a whole chain of elementwise ops:
 *     acc = inputs[0];  for each op:  acc = f(acc)            (NP_FUSED_UNARY,  op = np_unary_op)
 *                                     acc = acc (op) in[k]    (NP_FUSED_BINARY, op = np_binary_op,
 *                                     acc = in[k] (op) acc     operand = k, swap = 0 / 1)
 *     out[i] = acc
 * e.g. nd::exp($a) * $b + 2 costs 12 B/elem instead of 8 + 12 + 8 = 28 B/elem and two temporaries
 * (one allocation + one full round trip per PHP-level op in the reference, numpower.c:193-229).
 * Every step runs the same arithmetic as np_unary / np_binary (`value ** 2.0f` with a host scalar is
 * the product x * x in both), so the result is bit-identical to the unfused sequence; flags/body_end have the meaning they have in np_binary.  The result is
 * rows x cols; inputs[0] is NP_FULL (rows*cols elements) or NP_HOST_SCALAR, the other inputs have
 * any np_operand_kind: NP_FULL, NP_ROW (cols floats, added to every row), NP_COL (rows floats, one
 * per row), NP_SCALAR (device 0-d) or NP_HOST_SCALAR — the broadcast cases of ndarray.c:1196-1291
 * resolved by index arithmetic in the kernel, as in np_binary.  At most 6 inputs and 12 ops. */
typedef enum np_fused_kind { NP_FUSED_UNARY = 0, NP_FUSED_BINARY = 1 } np_fused_kind;
typedef struct np_fused_op {
    int kind;      /* np_fused_kind */
    int op;        /* np_unary_op or np_binary_op */
    int operand;   /* binary: index into inputs[] of the other operand */
    int swap;      /* binary: 0 = acc (op) operand, 1 = operand (op) acc */
    float p0, p1;  /* unary parameters (clip min/max, round decimals) */
    unsigned flags;     /* NP_QUIRK_AVX_BODY or 0 */
    size_t body_end;    /* see np_binary */
} np_fused_op;
int np_fused_chain(const float *const *inputs, const int *input_kinds, int n_inputs,
                   const np_fused_op *ops, int n_ops, float *out, size_t rows, size_t cols);
/* The same chain with a full reduction (np_reduce_op) as its last step: the chain value never goes
 * to memory — sum(exp(a) * b) reads 8 B/elem instead of writing 4 and reading 4 more.  Per-workgroup
 * partials + the deterministic second pass of np_reduce_all; NP_MEAN divides the sum by rows*cols
 * as NDArray::mean does (numpower.c:2659). */
int np_fused_chain_reduce(const float *const *inputs, const int *input_kinds, int n_inputs,
                          const np_fused_op *ops, int n_ops, int reduce_op, size_t rows, size_t cols,
                          float *host_out);
/* Same, the result left on the device (dev_out: 1 float) — no host round trip, so calls can be issued back to back. */
int np_fused_chain_reduce_dev(const float *const *inputs, const int *input_kinds, int n_inputs,
                              const np_fused_op *ops, int n_ops, int reduce_op, size_t rows, size_t cols,
                              float *dev_out);
/* ... and with a reduction over ONE axis of the rows x cols chain value as the last step: axis 1 (the
 * last axis) -> out[rows], axis 0 -> out[cols]; out is a DEVICE pointer.  sum(exp(X), 1), max(X - c, 1),
 * mean((X - mu) * (X - mu), 0) in one pass over X: 4 B/elem instead of 12.  Last axis: a wave (or, for
 * long or few rows, a workgroup) per row; first axis: a lane per 4-column slot, the workgroup's waves
 * interleaved over a chunk of rows, chunk partials folded by np_reduce_axis.  Shapes those kernels would
 * run mostly idle on (last axis: rows shorter than 16 floats, or fewer than 128 rows of a small array — a
 * handful of very long rows is cut into several workgroups per row; first axis: fewer than 32 rows or 32
 * column slots) take one fused pass into a temporary + np_reduce_axis instead.  Same combine rules as np_reduce_axis. */
int np_fused_chain_reduce_axis(const float *const *inputs, const int *input_kinds, int n_inputs,
                               const np_fused_op *ops, int n_ops, int reduce_op, size_t rows, size_t cols, int axis,
                               float *out);

/* ---- reductions -------------------------------------------------------------------------- */

typedef enum np_reduce_op {
    NP_SUM = 0, NP_PROD = 1, NP_MIN = 2, NP_MAX = 3, NP_MEAN = 4,
    NP_REDUCE_OP_COUNT
} np_reduce_op;

/* Full reduction to one host float.  Replaces cuda_sum_float / cuda_prod_float /
 * cuda_min_float / cuda_max_float (cuda_math.cu:921,934,1032,1016).  Deterministic two-pass
 * (wave64 shuffle + LDS, then one block over the per-block partials); no float atomics. */
int np_reduce_all(int op, const float *in, size_t n, float *host_out);
/* Same, result left on the device (1 float). */
int np_reduce_all_dev(int op, const float *in, size_t n, float *dev_out);

/* Statistics (src/ndmath/statistics.c:88-154).  The reference composes these
 * from Sum / Subtract / Abs / Pow with a full-size temporary per step (32 B/elem for variance);
 * here the array is read ONCE (4 B/elem): per-lane (count, mean, M2) merged pairwise in a fixed order
 * (Chan's formula; means carried as anchor + offset, so the result is as accurate as the two-pass form
 * for data with a large mean and a small spread).  host_m2 = sum (x - mean)^2:
 *   variance = m2 / n (NDArray_Variance), std = sqrtf(m2 / n) (NDArray_Std).  n >= 1. */
int np_moments(const float *in, size_t n, float *host_mean, float *host_m2);
/* Same, results left on the device: dev_out[0] = mean, dev_out[1] = m2 (2 floats). */
int np_moments_dev(const float *in, size_t n, float *dev_out);
/* NDArray_Average with weights (statistics.c:131-154): sum(a*w) and sum(w) from one kernel that reads
 * a and w once each (8 B/elem; no a*w temporary, products rounded before they are added as the
 * reference's Multiply-then-Sum). */
int np_weighted_sums(const float *a, const float *w, size_t n, float *host_sum_aw, float *host_sum_w);

/* argmax (is_max != 0) / argmin along the middle axis of outer x axis_len x inner; out receives
 * outer*inner indices AS FLOATS, like the reference (float_argmax / float_argmin,
 * src/ndmath/calculation.c:9-72: first occurrence wins; argmax: a NaN in position 0 is maximal,
 * later NaNs are never selected; argmin: the first NaN wins).  The reference rejects GPU arrays
 * ("GPU not supported.", calculation.c:75-78). */
int np_argreduce(int is_max, const float *in, size_t outer, size_t axis_len, size_t inner, float *out);

/* NDArray_All (logic.c:25-58): *host_out = 1 if every element is non-zero, else 0.
 * flags = NP_QUIRK_AVX_BODY reproduces what the reference's CPU code actually computes: its AVX2
 * body tests `movemask != 0x0F` on an 8-lane mask (logic.c:36-39), i.e. a full 8-element block
 * passes only if elements 0-3 are non-zero (and not NaN) and elements 4-7 ARE zero or NaN; the
 * scalar tail (`== 0.0`) is the only part that means "all non-zero".  flags = 0 gives the
 * intended meaning for every element. */
int np_all(const float *in, size_t n, unsigned flags, int *host_out);

/* Two-array predicates reduced to one flag (one streaming pass, 8 B/elem, no mask temporary):
 *   NP_MISMATCH_EXACT     *host_any = 1 if a[i] != b[i] for some i     compare_ndarrays, logic.c:686-690
 *                         (NaN != NaN counts, as in the reference's C loop; its CUDA path,
 *                         cuda_equal_float, is a launch + D2H of a flag per call)
 *   NP_MISMATCH_ALLCLOSE  *host_any = 1 if |a[i]-b[i]| > atol + rtol*|b[i]| for some i
 *                         float_allclose, logic.c:719-738 (a NaN on either side compares false,
 *                         i.e. "close", exactly as the reference's expression does; the reference
 *                         rejects GPU arrays: "`allclose` is not compatible with GPU operations.") */
typedef enum np_mismatch_mode { NP_MISMATCH_EXACT = 0, NP_MISMATCH_ALLCLOSE = 1 } np_mismatch_mode;
int np_count_mismatch(int mode, const float *a, const float *b, size_t n, float rtol, float atol,
                      int *host_any);

/* Order statistics: host_out2[0] = the k-th smallest element (k counted from 0), host_out2[1] = the
 * (k+1)-th (= the k-th again when k is the last rank).  Exact, by a three-pass radix select (12 B/elem
 * of reads, no sort, no copy) — replaces the copy + qsort() of calculate_median
 * (src/ndmath/arithmetics.c:111-138) and calculate_quantile (src/ndmath/statistics.c:14-50), whose
 * results depend on exactly these two values.  Ordering: IEEE total order on the bits (-0.0 before
 * +0.0, which the reference's comparator calls equal; NaNs at the ends by sign — the comparator is
 * inconsistent for NaNs, so the reference's own result is unspecified there).
 * np_order_stat_dev leaves the two floats in device memory (no synchronisation). */
int np_order_stat(const float *in, size_t n, size_t k, float *host_out2);
int np_order_stat_dev(const float *in, size_t n, size_t k, float *dev_out2);

/* Reduce the middle axis of a contiguous array viewed as outer x axis_len x inner; out has
 * outer*inner elements.  Replaces the host-side recursion reduce()/_reduce()/apply_reduce()
 * (ndarray.c:523-578,394-429,358-368), which issues one Add_Float + alloc + D2D copy + free
 * per slice, by one (or two) kernels.
 * flags: NP_QUIRK_AVX_BODY with NP_PROD reproduces the sign the reference's repeated
 * NDArray_Multiply_Float leaves on ZERO results (arithmetics.c:403,410-412): -0.0f for inner
 * index < np_avx_body_end(inner), +0.0f after it; pass it only when the reduced slices are
 * at least 1-D (0-d slices take the plain a*b short cut, arithmetics.c:302-316). */
int np_reduce_axis(int op, const float *in, size_t outer, size_t axis_len, size_t inner,
                   float *out, unsigned flags);
/* Scratch the axis reduction may need for its partials, in bytes (0 if none); the library
 * allocates it from the pool itself — exported so callers can size memory budgets. */
size_t np_reduce_axis_workspace(size_t outer, size_t axis_len, size_t inner);

/* ---- matmul --------------------------------------------------------------------------------- */

/* C[MxN] = A[MxK] . B[KxN], row-major, alpha = 1, beta = 0, exact fp32 (v_mfma_f32_32x32x2_f32).
 * Replaces cblas_sgemm(RowMajor,N,N,…) / cublasSgemm in NDArray_FMatmul (linalg.c:44-82). */
int np_sgemm(size_t M, size_t N, size_t K, const float *A, const float *B, float *C);
/* batch independent products; matrix b of X starts at X + b*stride_x (strides in elements).
 * The reference has no batched entry point (linalg.c:239-242 rejects ndim > 2); this is the
 * loop-of-2-D-calls collapsed into one launch (BASELINE config 5). */
int np_sgemm_strided_batched(size_t batch, size_t M, size_t N, size_t K,
                             const float *A, size_t stride_a, const float *B, size_t stride_b,
                             float *C, size_t stride_c);
/* The same for `count` matrices that are a PIECE of a batch of `whole` (a caller that pipelines a batch piece by piece,
 * numpower_amd/parallel.py): the piece runs the kernel configuration np_sgemm_strided_batched would pick for the whole
 * batch, so that the pieces together are bit-identical to the one call — a piece of one matrix would otherwise go to
 * the single-product planner (split-K, stream-K ...), which sums in a different order.  whole >= count. */
int np_sgemm_strided_batched_piece(size_t count, size_t whole, size_t M, size_t N, size_t K,
                                   const float *A, size_t stride_a, const float *B, size_t stride_b,
                                   float *C, size_t stride_c);
/* y[M] = A[MxN] . x[N]; replaces cblas_sgemv / matrixVectorMultiplyFloatKernel
 * (linalg.c:367-386, cuda_math.cu:228,1417). */
int np_sgemv(size_t M, size_t N, const float *A, const float *x, float *y);
/* out[M x N] = a (M) outer b (N), row-major: NDArray_Outer (linalg.c:724-751; cblas_sger on a zeroed
 * matrix / cuda_calculate_outer_product), i.e. out[i][j] = 0 + a[i]*b[j] (zero products are +0.0). */
int np_outer(const float *a, size_t M, const float *b, size_t N, float *out);

/* ---- layout ---------------------------------------------------------- */

/* out[b][c][r] = in[b][r][c] for batch matrices of rows x cols (contiguous); in != out.
 * Replaces cuda_float_transpose / transposeCoalesced (cuda_math.cu:136-150,1288-1294: fixed 16x16
 * grid, only correct up to 256 x 256) with a 64 x 64 LDS-tiled transpose, any size. */
int np_transpose2d(const float *in, float *out, size_t batch, size_t rows, size_t cols);
/* General axis permutation into a new contiguous buffer: out.shape[i] = shape[perm[i]]
 * (NDArray_Transpose + NDArray_ToContiguous, manipulation.c:68-130,381-421).  ndim <= 8;
 * shape/perm are host arrays.  Errors: "axes don't match array", "repeated axis in transpose". */
int np_permute(const float *in, float *out, int ndim, const int *host_shape, const int *host_perm);
/* Gather a strided view into a contiguous buffer: out[i0]...[ik] = in[sum i_d * strides[d]]
 * (strides in ELEMENTS, may be negative or zero).  One launch instead of NDArray_ToContiguous's
 * one 4-byte D2D memcpy per element (manipulation.c:381-421); also what NDArray_Diagonal
 * (indexing.c:21-48: one memcpy per diagonal element) and multi-index NDArray_Slice
 * (manipulation.c:193-283) reduce to. */
int np_strided_copy(const float *in, float *out, int ndim, const int *host_shape, const long long *host_strides);
/* Pitched 2-D copy: rows x width floats, consecutive rows dst_pitch / src_pitch floats apart (both
 * >= width).  One launch per input of NDArray_Concatenate along an inner axis (manipulation.c:894-997,
 * where NDArray_AssignArray copies element by element through the sliding view). */
int np_copy2d(float *dst, size_t dst_pitch, const float *src, size_t src_pitch, size_t width, size_t rows);

/* ---- device-side initializers (the reference's own phpbench suite, benchmarks/initializers/) ------
 * The reference builds every array on the CPU (initializers.c) and a GPU user pays the PCIe copy on
 * ->gpu(); zeros / ones / full are np_memset0 / np_fill above.
 * np_identity: n x n, ones on the diagonal (NDArray_Identity, initializers.c:479-510).
 * np_arange:   x[0] = (float)start, x[i] = (float)((double)x[i-1] + step) — NDArray_Arange's
 *              recurrence (initializers.c:836-839), bit for bit, evaluated as per-binade arithmetic
 *              segments (see np_layout.hip). */
int np_identity(float *out, size_t n);
int np_arange(float *out, double start, double step, size_t n);

/* ---- the path's one collective (BASELINE config 5) --------------------------------------
 * One process per GPU (the reference's device model: NDArray::setDevice, numpower.c:615-635; it has no
 * multi-device code).  A batch-parallel workload keeps contiguous slabs per rank — e.g. batch b of a batched
 * matmul on rank b / (batch / world) — computes them with np_sgemm_strided_batched, and replicates the
 * result with ONE all-gather over xGMI (RCCL, loaded on demand).  All calls are enqueued on the library stream,
 * ordered with the kernels; nothing here is needed — or loaded — by a single-GPU process.
 *
 * np_comm_init   rank in [0, world); endpoint = "tcp://host:port" (rank 0 serves the RCCL id on that port
 *                to peers that introduce themselves; use 127.0.0.1 and a free port on one node) or a file path
 *                (rank 0 publishes the id there — replacing a stale file — peers poll for it; removed by
 *                np_comm_destroy and on a failed init).  Blocks until every rank has joined (120 s limit).  The
 *                device is the one np_init selected.
 * np_allgather   recv[r * bytes_per_rank ...] = rank r's send buffer, for every r; in place when
 *                send == recv + rank * bytes_per_rank.  Enqueued on the LIBRARY stream: ordered behind the kernels,
 *                no overlap with them.  Asynchronous for the host (np_sync / a read-back waits for it).
 * np_comm_max    max of one host float over the ranks (timing: the slowest rank); blocks.
 * np_comm_barrier returns once every rank's streams (library and communication) have reached the call. */
int np_comm_init(int rank, int world, const char *endpoint);
int np_comm_rank(void);    /* -1 without a communicator */
/* The collective library behind np_comm_*: ncclGetVersion() of the librccl.so.1 the library loads (e.g. 22707 = 2.27.7);
 * needs no communicator and no device — what bench.py prints next to a multi-GPU line. */
int np_comm_rccl_version(int *host_version);
int np_comm_world(void);   /*  0 without a communicator */
int np_allgather(const void *dev_send, void *dev_recv, size_t bytes_per_rank);
int np_comm_max(float value, float *host_max);
int np_comm_barrier(void);
int np_comm_destroy(void);

/* The overlapped form.  The communicator owns a second, high-priority stream.  np_allgather_async makes that stream
 * wait (an event, on the device) for everything the library stream has been given so far, then enqueues the gather
 * THERE and returns: kernels launched afterwards on the library stream run while the data travels.
 *   dev_recv_base + r * recv_stride_bytes  <-  rank r's `bytes` at dev_send, for every r  (recv_stride_bytes >= bytes;
 *   in place when dev_send is this rank's destination).
 * How the bytes travel:
 *   NP_GATHER_COLLECTIVE  one ncclAllGather; needs recv_stride_bytes == bytes
 *   NP_GATHER_P2P         one grouped exchange of ncclSend / ncclRecv pairs: each peer's piece crosses that peer's own
 *                         xGMI link straight into place (any stride; no staging buffer, no scatter copy)
 *   NP_GATHER_AUTO        COLLECTIVE when the destinations are contiguous, else P2P
 * np_comm_wait makes the library stream wait (on the device; the host does not block) for everything given to the
 * communication stream so far: call it before the gathered data is read by a kernel, np_sync or a copy.  Until then
 * the send region must not be rewritten and the destination not read. */
typedef enum np_gather_mode { NP_GATHER_AUTO = 0, NP_GATHER_COLLECTIVE = 1, NP_GATHER_P2P = 2 } np_gather_mode;
int np_allgather_async(const void *dev_send, void *dev_recv_base, size_t bytes, size_t recv_stride_bytes, int mode);
int np_comm_wait(void);
void *np_comm_stream(void);   /* the communication stream as a raw hipStream_t (NULL without a communicator) */

/* BASELINE config 5 as one call: C_full is the replicated result, world * slab matrices of M x N, contiguous.  This
 * rank multiplies its slab (A, B: `slab` matrices, strides in elements) straight into its window
 * C_full + rank * slab * M * N, in `chunks` pieces (as equal as they come; clipped to slab); the gather of piece c is
 * handed to the communication stream as soon as piece c's GEMM is enqueued and travels while piece c + 1 computes.
 * Ends with np_comm_wait(), so the next call on the library stream (or np_sync) sees the whole result.  chunks = 0
 * leaves the piece count to the library: a step model of the pipeline (GEMM rate, one xGMI link per peer, the start delay
 * a transfer pays while the GEMM holds the CUs) picks it from 1, 2, 4, 8, 16 — one piece when nothing travels (a one-rank
 * communicator) or under NP_GATHER_COLLECTIVE; this is what a caller without a reason of its own should pass.  chunks = 1
 * is "compute, then one all-gather" on two streams (mode picks the transport); chunks > 1 always travels P2P (the
 * pieces of a slab are one slab apart across ranks) and mode NP_GATHER_COLLECTIVE is refused.  The result is
 * bit-identical for every chunks / mode: the same GEMM kernel computes every matrix.  The reference has no batched
 * or multi-device matmul (linalg.c:239-242 rejects ndim > 2, numpower.c:615-635): the oracle for this call is the
 * loop of 2-D NDArray_Matmul over the batch. */
/* How the call above cuts a slab: piece c of `chunks` starts at item *host_lo and holds *host_count items (the first
 * slab % chunks pieces hold one more).  Pure arithmetic; needs no device and no communicator. */
int np_comm_piece(size_t slab, int chunks, int c, size_t *host_lo, size_t *host_count);
int np_sgemm_strided_batched_allgather(size_t slab, size_t M, size_t N, size_t K, const float *A, size_t stride_a,
                                       const float *B, size_t stride_b, float *C_full, int chunks, int mode);

/* Device-side waits.  A wait for this GPU's own work (a stream-ordering wait of the sharded matmul whose producer never ran,
 * a GEMM workgroup that never saw its siblings' partial tiles) gives up after a bounded number of polls and raises the
 * device's error word: np_sync / np_memcpy_d2h / host-result calls / np_comm_* calls on that device return NP_ERR_DEVICE until
 * np_clear_device_error() acknowledges it.
 * A wait for TRANSFERS (which depend on other ranks) gives up after np_comm_set_wait_limit seconds (default 600; 0 = never,
 * like the RCCL kernel it waits for) and raises the same error; np_comm_destroy releases whatever still waits
 * (after a grace period of 30 s, or of the wait limit if that is shorter, for transfers that are merely still travelling).
 * With peers (world > 1) the sharded GEMM is one launch per piece; see np_hip_debug.h (np_comm_set_variant) for the
 * forms kept for A/B measurements. */
int np_comm_set_wait_limit(double seconds);

/* Everything that only tests, sweeps and A/B measurements need — kernel-variant switches (process-global state),
 * planner / rendezvous / transport probes, error injection — is declared in np_hip_debug.h.  Same library; a binding
 * needs none of it. */

#ifdef __cplusplus
}
#endif

#endif /* NUMPOWER_AMD_NP_HIP_H */
