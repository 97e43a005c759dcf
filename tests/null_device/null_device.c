/*
 * null_device.c — TEST INFRASTRUCTURE for sanitizer runs of the HOST code on the CPU build (tests/test_host_sanitizers_cpu.py).
 *
 * The entry points of include/np_hip.h that libnumpower_host.so and the ext/ glue call, over malloc — a "device" that COMPUTES
 * NOTHING: every compute call counts one launch, reads every byte of the ranges a real kernel would read and zero-fills the
 * ranges it would write.  Nothing is compared with anything and no result leaves this harness: it is not an oracle, not a
 * fallback and never part of the product (the product fails loudly without libnp_hip.so and a GPU).  What it is for: the
 * host-side code of this repository — numpower_host.cpp, ext/hip_fast.c, ext/hip_lazy.c (pending chains: reference counts,
 * the side table, flushes on write, lifetimes), ext/hip_math_drivers.c, ext/gpu_alloc_hip.c and the generated programs
 * method_bodies / fast_path_bodies / lazy_bodies — compiled with -fsanitize=address,undefined and run end to end without a GPU
 * (GPU AddressSanitizer is not available on the pool; "run sanitizers on the CPU build only").  Because the ranges are touched,
 * AddressSanitizer checks every extent the host code passes down: a result buffer allocated too small, a chain flushed after
 * its input was freed, an operand kind that does not match the array's size all end as a report.
 *
 * Only memory is real: np_malloc / np_free (live count as the pool's), the copies, np_fill / np_memset0 / np_identity /
 * np_arange (initialisers: a value is a value), np_read_float.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "np_hip.h"
#include "np_hip_debug.h"

static unsigned long long g_launches;
static long g_live;
static char g_error[256];

static int fail(int code, const char *msg) {
    snprintf(g_error, sizeof g_error, "%s", msg);
    return code;
}

static volatile unsigned char g_sink;
static void reads(const void *p, size_t bytes) {
    const unsigned char *q = (const unsigned char *) p;
    unsigned char acc = 0;
    for (size_t i = 0; i < bytes; i++) acc ^= q[i];
    g_sink ^= acc;
}
static void writes(void *p, size_t bytes) { memset(p, 0, bytes); }
#define F(n) ((size_t) (n) * sizeof(float))

static size_t kind_elems(int kind, size_t rows, size_t cols) {
    switch (kind) {
        case NP_FULL: return rows * cols;
        case NP_ROW: return cols;
        case NP_COL: return rows;
        default: return 1;                 /* NP_SCALAR (device 0-d) and NP_HOST_SCALAR (host pointer): one float */
    }
}

/* ---- runtime ---- */
int np_init(int device) { return device == 0 ? NP_OK : fail(NP_ERR_INVALID, "null device: only device 0"); }
int np_set_device(int device) { return np_init(device); }
int np_device_count(int *host_count) { if (host_count) *host_count = 1; return NP_OK; }
int np_sync(void) { return NP_OK; }
int np_clear_device_error(unsigned *host_bits) { if (host_bits) *host_bits = 0; return NP_OK; }
const char *np_last_error(void) { return g_error; }
const char *np_version(void) { return "null device (tests/null_device: computes nothing)"; }
int np_debug_launch_count(unsigned long long *host_count) { if (host_count) *host_count = g_launches; return NP_OK; }

int np_malloc(void **dev_ptr, size_t bytes) {
    if (!dev_ptr) return fail(NP_ERR_INVALID, "np_malloc: null output");
    *dev_ptr = calloc(bytes ? bytes : 1, 1);
    if (!*dev_ptr) return fail(NP_ERR_ALLOC, "np_malloc: out of memory");
    g_live++;
    return NP_OK;
}
int np_free(void *dev_ptr) {
    if (dev_ptr) {
        free(dev_ptr);
        g_live--;
    }
    return NP_OK;
}
long np_live_allocs(void) { return g_live; }
int np_memcpy_h2d(void *dst, const void *src, size_t bytes) { memcpy(dst, src, bytes); return NP_OK; }
int np_memcpy_d2h(void *dst, const void *src, size_t bytes) { memcpy(dst, src, bytes); return NP_OK; }
int np_memcpy_d2d(void *dst, const void *src, size_t bytes) { memmove(dst, src, bytes); return NP_OK; }
int np_memset0(void *dev_ptr, size_t bytes) { memset(dev_ptr, 0, bytes); return NP_OK; }
int np_fill(float *dev_ptr, float value, size_t n) { for (size_t i = 0; i < n; i++) dev_ptr[i] = value; return NP_OK; }
int np_read_float(const float *dev_ptr, size_t index, float *host_out) { *host_out = dev_ptr[index]; return NP_OK; }
int np_identity(float *out, size_t n) {
    memset(out, 0, F(n * n));
    for (size_t i = 0; i < n; i++) out[i * n + i] = 1.0f;
    return NP_OK;
}
int np_arange(float *out, double start, double step, size_t n) {
    for (size_t i = 0; i < n; i++) out[i] = (float) (start + step * (double) i);
    return NP_OK;
}
size_t np_avx_body_end(size_t numel_a) { return numel_a - numel_a % 8; }

/* ---- elementwise ---- */
int np_binary(int op, const float *a, int a_kind, const float *b, int b_kind, float *out, size_t rows, size_t cols, unsigned flags,
              size_t body_end) {
    (void) op; (void) flags; (void) body_end;
    g_launches++;
    if (rows == 0 || cols == 0) return NP_OK;
    reads(a, F(kind_elems(a_kind, rows, cols)));
    reads(b, F(kind_elems(b_kind, rows, cols)));
    writes(out, F(rows * cols));
    return NP_OK;
}
int np_unary(int op, const float *in, float *out, size_t n, float p0, float p1) {
    (void) op; (void) p0; (void) p1;
    g_launches++;
    reads(in, F(n));
    if (in != out) writes(out, F(n));
    return NP_OK;
}
static int chain_reads(const float *const *inputs, const int *kinds, int n_inputs, const np_fused_op *ops, int n_ops, size_t rows, size_t cols) {
    if (n_inputs < 1 || n_inputs > 6 || n_ops < 0 || n_ops > 12) return fail(NP_ERR_INVALID, "null device: chain too long");
    for (int i = 0; i < n_inputs; i++) reads(inputs[i], F(kind_elems(kinds[i], rows, cols)));
    for (int k = 0; k < n_ops; k++)
        if (ops[k].kind == NP_FUSED_BINARY && (ops[k].operand < 0 || ops[k].operand >= n_inputs))
            return fail(NP_ERR_INVALID, "null device: operand index out of range");
    return NP_OK;
}
int np_fused_chain(const float *const *inputs, const int *input_kinds, int n_inputs, const np_fused_op *ops, int n_ops, float *out,
                   size_t rows, size_t cols) {
    g_launches++;
    if (chain_reads(inputs, input_kinds, n_inputs, ops, n_ops, rows, cols)) return NP_ERR_INVALID;
    writes(out, F(rows * cols));
    return NP_OK;
}
int np_fused_chain_reduce(const float *const *inputs, const int *input_kinds, int n_inputs, const np_fused_op *ops, int n_ops,
                          int reduce_op, size_t rows, size_t cols, float *host_out) {
    (void) reduce_op;
    g_launches++;
    if (chain_reads(inputs, input_kinds, n_inputs, ops, n_ops, rows, cols)) return NP_ERR_INVALID;
    *host_out = 0.0f;
    return NP_OK;
}
int np_fused_chain_reduce_axis(const float *const *inputs, const int *input_kinds, int n_inputs, const np_fused_op *ops, int n_ops,
                               int reduce_op, size_t rows, size_t cols, int axis, float *out) {
    (void) reduce_op;
    g_launches++;
    if (chain_reads(inputs, input_kinds, n_inputs, ops, n_ops, rows, cols)) return NP_ERR_INVALID;
    writes(out, F(axis == 0 ? cols : rows));
    return NP_OK;
}

/* ---- reductions, statistics, predicates ---- */
int np_reduce_all(int op, const float *in, size_t n, float *host_out) {
    (void) op;
    g_launches++;
    reads(in, F(n));
    *host_out = 0.0f;
    return NP_OK;
}
int np_reduce_axis(int op, const float *in, size_t outer, size_t axis_len, size_t inner, float *out, unsigned flags) {
    (void) op; (void) flags;
    g_launches++;
    reads(in, F(outer * axis_len * inner));
    writes(out, F(outer * inner));
    return NP_OK;
}
int np_argreduce(int is_max, const float *in, size_t outer, size_t axis_len, size_t inner, float *out) {
    (void) is_max;
    g_launches++;
    reads(in, F(outer * axis_len * inner));
    writes(out, F(outer * inner));
    return NP_OK;
}
int np_all(const float *in, size_t n, unsigned flags, int *host_out) {
    (void) flags;
    g_launches++;
    reads(in, F(n));
    *host_out = 0;
    return NP_OK;
}
int np_count_mismatch(int mode, const float *a, const float *b, size_t n, float rtol, float atol, int *host_any) {
    (void) mode; (void) rtol; (void) atol;
    g_launches++;
    reads(a, F(n));
    reads(b, F(n));
    *host_any = 0;
    return NP_OK;
}
int np_moments(const float *in, size_t n, float *host_mean, float *host_m2) {
    g_launches++;
    reads(in, F(n));
    *host_mean = 0.0f;
    *host_m2 = 0.0f;
    return NP_OK;
}
int np_weighted_sums(const float *a, const float *w, size_t n, float *host_sum_aw, float *host_sum_w) {
    g_launches++;
    reads(a, F(n));
    reads(w, F(n));
    *host_sum_aw = 0.0f;
    *host_sum_w = 1.0f;
    return NP_OK;
}
int np_order_stat(const float *in, size_t n, size_t k, float *host_out2) {
    if (k >= n) return fail(NP_ERR_INVALID, "np_order_stat: k out of range");
    g_launches++;
    reads(in, F(n));
    host_out2[0] = host_out2[1] = 0.0f;
    return NP_OK;
}

/* ---- products ---- */
int np_sgemm(size_t M, size_t N, size_t K, const float *A, const float *B, float *C) {
    g_launches++;
    reads(A, F(M * K));
    reads(B, F(K * N));
    writes(C, F(M * N));
    return NP_OK;
}
int np_sgemm_strided_batched(size_t batch, size_t M, size_t N, size_t K, const float *A, size_t stride_a, const float *B, size_t stride_b,
                             float *C, size_t stride_c) {
    g_launches++;
    for (size_t i = 0; i < batch; i++) {
        reads(A + i * stride_a, F(M * K));
        reads(B + i * stride_b, F(K * N));
        writes(C + i * stride_c, F(M * N));
    }
    return NP_OK;
}
int np_sgemv(size_t M, size_t N, const float *A, const float *x, float *y) {
    g_launches++;
    reads(A, F(M * N));
    reads(x, F(N));
    writes(y, F(M));
    return NP_OK;
}
int np_outer(const float *a, size_t M, const float *b, size_t N, float *out) {
    g_launches++;
    reads(a, F(M));
    reads(b, F(N));
    writes(out, F(M * N));
    return NP_OK;
}

/* ---- layout ---- */
int np_transpose2d(const float *in, float *out, size_t batch, size_t rows, size_t cols) {
    g_launches++;
    reads(in, F(batch * rows * cols));
    writes(out, F(batch * rows * cols));
    return NP_OK;
}
int np_permute(const float *in, float *out, int ndim, const int *host_shape, const int *host_perm) {
    size_t n = 1;
    for (int d = 0; d < ndim; d++) {
        if (host_perm[d] < 0 || host_perm[d] >= ndim) return fail(NP_ERR_INVALID, "np_permute: bad permutation");
        n *= (size_t) host_shape[d];
    }
    g_launches++;
    reads(in, F(n));
    writes(out, F(n));
    return NP_OK;
}
int np_strided_copy(const float *in, float *out, int ndim, const int *host_shape, const long long *host_strides) {
    /* every element of the view is read where it lies */
    size_t n = 1;
    for (int d = 0; d < ndim; d++) n *= (size_t) host_shape[d];
    g_launches++;
    for (size_t i = 0; i < n; i++) {
        size_t rest = i;
        long long off = 0;
        for (int d = ndim - 1; d >= 0; d--) {
            off += (long long) (rest % (size_t) host_shape[d]) * host_strides[d];
            rest /= (size_t) host_shape[d];
        }
        reads(in + off, sizeof(float));
    }
    writes(out, F(n));
    return NP_OK;
}
int np_copy2d(float *dst, size_t dst_pitch, const float *src, size_t src_pitch, size_t width, size_t rows) {
    g_launches++;
    for (size_t r = 0; r < rows; r++) memmove(dst + r * dst_pitch, src + r * src_pitch, F(width));
    return NP_OK;
}

/* ---- the collective: a one-rank communicator (nothing travels) ---- */
static int g_comm;
int np_comm_init(int rank, int world, const char *endpoint) {
    (void) endpoint;
    if (rank != 0 || world != 1) return fail(NP_ERR_DEVICE, "null device: a one-rank communicator only");
    g_comm = 1;
    return NP_OK;
}
int np_comm_rank(void) { return g_comm ? 0 : -1; }
int np_comm_world(void) { return g_comm ? 1 : 0; }
int np_comm_destroy(void) { g_comm = 0; return NP_OK; }
int np_sgemm_strided_batched_allgather(size_t slab, size_t M, size_t N, size_t K, const float *A, size_t stride_a, const float *B,
                                       size_t stride_b, float *C_full, int chunks, int mode) {
    (void) chunks; (void) mode;
    if (!g_comm) return fail(NP_ERR_INVALID, "np_sgemm_strided_batched_allgather: no communicator (np_comm_init first)");
    return np_sgemm_strided_batched(slab, M, N, K, A, stride_a, B, stride_b, C_full, M * N);
}
