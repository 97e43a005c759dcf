// Internal helpers shared by the HIP translation units of libnp_hip.so (not part of the ABI).
#ifndef NUMPOWER_AMD_NP_INTERNAL_H
#define NUMPOWER_AMD_NP_INTERNAL_H

#include <hip/hip_runtime.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "np_hip.h"
#include "np_hip_debug.h"

namespace np {

// Records msg as the calling thread's last error and returns code.
int fail(int code, const char *fmt, ...);
// Library stream (never the null stream unless the caller installed it).
hipStream_t stream();
// Makes sure np_init ran (lazily initialises device 0).
int ensure_init();
// Device properties cached at init.
int num_cus();
// Host-result slots: 64 floats of pinned, device-visible host memory.  A reduction's last kernel writes its value
// straight into a slot and the caller only waits for the stream (wait()) and reads it — no 4-byte D2H copy call
// behind every nd::sum() / allclose() / median() (that call alone is ~10 us of host + driver time).  One set per
// process, handed out to ONE host-result call at a time: the guard holds a process-wide mutex from arming the slots
// to the end of the call, so concurrent callers (threads of a ctypes host, ZTS PHP) queue up instead of re-arming
// and reading each other's slots.  Do not nest two guards in one scope.
struct ResultCall {
    explicit ResultCall(int count = 1);   // `count` = how many slots the call's kernels will write
    ~ResultCall();
    ResultCall(const ResultCall &) = delete;
    ResultCall &operator=(const ResultCall &) = delete;
    float *slot = nullptr;                // nullptr + error set on failure
    int wait();                           // wait for the library stream (spinning on the slots, np_runtime.hip)
};
// A zeroed device counter for one launch of a "last workgroup folds" kernel (np::dev::fold_in_last_workgroup): taken
// from a ring of 1024, and put back to zero by the workgroup that used it up, so a slot is clean again long before
// the ring comes round (nullptr + error set on failure).
unsigned *next_ticket();
// `count` consecutive zeroed counters (count <= 256), one per independent fold of the same launch
unsigned *next_tickets(unsigned count);
// Stream-K's flag array of the current device (np_sgemm.hip): kStreamKFlagCount zeroed counters allocated by np_init — never
// lazily, the first stream-K launch may sit inside a stream capture (nullptr + error set on failure).
constexpr unsigned kStreamKFlagCount = 1024;
unsigned *streamk_flags();
// The CURRENT device's error word (pinned host memory, device-visible, one pair of words per device; nullptr before
// np_init).  Word [0]: a kernel whose device-side wait gives up ORs one of the kErr* bits in with a system-scope atomic;
// np_sync, np_memcpy_d2h, the host-result calls and every np_comm_* entry point turn a non-zero word OF THE DEVICE THEY RUN
// ON into NP_ERR_DEVICE (check_device_error) — every time, until np_clear_device_error() acknowledges it: the first reader
// does not eat the error of whoever owns the failed launch, and another device's np_sync does not see it at all.
// Word [1]: the host's ABORT request — an unbounded device-side wait (np_comm.hip: the library stream waiting for transfers
// that depend on other ranks) polls it and returns when it is non-zero (np_comm_destroy sets it).
constexpr unsigned kErrCommWait = 1u, kErrStreamK = 2u;
unsigned *device_error_word();
int check_device_error(const char *who);
// kernel launches issued by the library since it was loaded (one per NP_LAUNCH_CHECK; np_debug_launch_count)
extern unsigned long long g_launch_count;
// Largest first-pass grid whose partials are folded by its own last workgroup rather than by a second kernel.  The
// ticket is one hot address (~20 ns per workgroup at the memory side) and every workgroup waits a round trip for its
// own: with 2049 workgroups the in-kernel fold LOST 5 us on a 63 us sum of 10^8 floats; on small grids it saves the
// second launch — nd::sum() of 1024 floats 13.2 -> 10.3 us, of 10^5 (98 workgroups) 15.9 -> 13.0 (tools/latency_ab.py).
// (Cutting mid-size grids down to 511 fatter workgroups so that they qualify was tried: 10^6 floats 17.0 -> 18.1 us.)
constexpr size_t kFoldInKernelMaxBlocks = 256;
// Round 5 tried to lift that limit with tickets taken in 32 GROUPS (workgroup b draws from counter 1 + b % 32, the last of a
// group from counter 0): correct and bit-identical, but nd::sum() of 10^6 floats (977 workgroups) took 16.1 us that way against
// 12.3 us with the second launch — two dependent memory-side round trips for the tickets and one more for the partials cost
// more than a 4 us launch (profiles/r05/reduce_small_ab.log) — and was removed.  What did help a little: np_reduce_all of up to
// 2^20 elements on at most g_small_reduce_blocks fat workgroups, which then fold behind ONE ticket (10^6: 12.0 us).
extern size_t g_fold_in_kernel_max;   // largest grid that folds in-kernel, <= kFoldInKernelMaxBlocks (np_reduce_set_variant(2000000 + N): A/B, tests)
extern size_t g_small_reduce_blocks;  // np_reduce_all of up to 2^20 elements on at most this many workgroups (0 = off; np_reduce_set_variant(3000000 + N))
// the ticket for a first pass of `blocks` workgroups, or nullptr (a second launch folds)
inline unsigned *fold_ticket(size_t blocks) {
    if (blocks <= kFoldInKernelMaxBlocks && blocks <= g_fold_in_kernel_max) return next_ticket();
    return nullptr;
}

// Block count of a capped grid-stride kernel.  With `cap` a power of two (CUs x 8 ...) every lane's accesses
// sit a power-of-two number of bytes apart — 2048 workgroups x 4 KiB = 8 MiB — and land on the same HBM channel:
// the full sum of 1e8 floats ran 5.66 TB/s on 2048 workgroups and 6.39 on 2049 (profiles/r02/reduce_cap_ab_2.log).
// A capped grid is therefore made odd.
inline size_t capped_grid(size_t wanted, size_t cap) {
    if (wanted <= cap) return wanted < 1 ? 1 : wanted;
    return cap | 1;
}
// One-block fold of per-workgroup partials into one device float (np_reduce.hip); op = NP_SUM / PROD / MIN / MAX.
int fold_partials(int op, const float *partials, size_t n, float *dev_out);
// Batched GEMM in one launch with per-piece progress counters (np_sgemm.hip; used by np_comm.hip's overlapped pipeline).
int sgemm_batched_with_progress(size_t batch, size_t M, size_t N, size_t K, const float *A, size_t stride_a, const float *B,
                                size_t stride_b, float *C, size_t stride_c, unsigned *counters, int chunks,
                                unsigned *tiles_per_matrix);
// np_sgemm_strided_batched for `count` matrices that are a piece of a batch of `whole`: planned as the whole batch (np_sgemm.hip)
int sgemm_batched_piece(size_t count, size_t whole, size_t M, size_t N, size_t K, const float *A, size_t stride_a, const float *B,
                        size_t stride_b, float *C, size_t stride_c);
// Compiled chains (np_fused_static.hip): what np_elementwise.hip's fused_chain_impl hands over when a chain of 1-3 steps
// might be on the menu of straight-line kernels.  operand[k] == nullptr: a scalar operand (or a unary step); idx[k]: how an
// array operand is indexed by the flat element index e of the rows x cols result — 0 full, 1 row (e % cols), 2 column
// (e / cols), 3 zero-d.  bcast_cols: the row length when some operand is broadcast (then a multiple of 4), else 0.
extern int g_bcast2d_off;   // np_elementwise_set_variant(8100): the 2-D broadcast kernels off (A/B)
struct FusedStaticDesc {
    int n_ops;
    const float *in0;
    int kind[3], op[3], swap[3], idx[3];
    const float *operand[3];
    float scalar[3], p0[3], p1[3];
    int quirk[3];            // a multiply step with NP_QUIRK_AVX_BODY (flat kernels only) ...
    unsigned body_end[3];    // ... whose AVX2 body ends at this flat index
    unsigned bcast_cols, div_m, div_s1, div_s2;
};
// sink < 0: the chain value is stored; else NP_SUM ... ; axis_mode -1 flat, 0 first axis, 1 last axis
bool fused_static_covers(const FusedStaticDesc &d, int sink, int axis_mode);
// flat: out[e] = chain(e) (sink < 0) or one partial per workgroup in out[] under the interpreter's protocol (ticket / result)
int fused_static_flat(const FusedStaticDesc &d, float *out, size_t n, int sink, unsigned grid, unsigned *ticket, float *result);
// first axis: out[chunk][c] = sum over the chunk's rows (the geometry fused_chain_impl computed for the interpreter's kernel)
int fused_static_cols(const FusedStaticDesc &d, float *out, size_t rows, size_t cols, size_t rows_per_chunk, float mean_div,
                      unsigned col_blocks, unsigned chunks, int rows_in_flight);
// last axis: out[r] = sum over the row, lane groups of L lanes per row (the interpreter's wave mode); cols % 4 == 0
int fused_static_rows(const FusedStaticDesc &d, float *out, size_t rows, size_t cols, unsigned L, float mean_div, unsigned grid);
// Copy kernel for large word-aligned device-to-device copies (np_elementwise.hip); bytes % 4 == 0.
int device_copy(void *dst, const void *src, size_t bytes);

// Scratch from the caching pool, released back to the pool when the object dies.  Freed blocks
// are only reused by later launches on the same stream, so releasing right after the launch
// that uses them is safe (stream order).
struct Scratch {
    void *ptr = nullptr;
    int alloc(size_t bytes);
    ~Scratch();
};

// np_comm.hip: true when a communicator lives on `device` and its communication stream still has unfinished work after
// `grace_s` seconds — a transfer that waits for a peer (late, or gone: the collective library's kernel then never ends and
// anything that synchronises the whole device would never return).  np_comm_destroy() is what ends such a transfer.
bool comm_transfers_stuck(int device, double grace_s);

}  // namespace np

// ---- device-side reduction helpers shared by np_reduce.hip and the fused chain kernel ----
namespace np {
namespace dev {

template <int OP>
__device__ __forceinline__ float r_identity() {
    if constexpr (OP == NP_SUM || OP == NP_MEAN) return 0.0f;
    if constexpr (OP == NP_PROD) return 1.0f;
    if constexpr (OP == NP_MIN) return INFINITY;
    return -INFINITY;
}

template <int OP>
__device__ __forceinline__ float r_combine(float a, float b) {
    if constexpr (OP == NP_SUM || OP == NP_MEAN) return a + b;
    if constexpr (OP == NP_PROD) return a * b;
    // same comparison the reference uses (ndarray.c:764, :951): NaN never replaces
    if constexpr (OP == NP_MIN) return (b < a) ? b : a;
    return (b > a) ? b : a;
}

template <int OP>
__device__ __forceinline__ float wave_reduce(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = r_combine<OP>(v, __shfl_down(v, off, 64));
    return v;   // lane 0 holds the wave's result
}

// ---- handing a workgroup's result to the LAST workgroup of the same kernel ----
// A second, one-workgroup kernel behind a streaming pass costs 4-5 us however little it folds.  Instead every workgroup
// stores its partial, takes a ticket, and the one that draws the last ticket folds them all.  Everything that crosses
// workgroups travels by device-scope atomics (store, RMW, load): these are performed at the memory side, past the
// eight XCD L2s, which are not coherent with each other.  A wave takes its ticket only after its own stores have been
// acknowledged (s_waitcnt 0).  Device-scope FENCES are deliberately not used: each writes back / invalidates an L2,
// and a thousand workgroups doing that tripled the time of the streaming pass they ended (np_select.hip: 73 -> 240 us).
// The ticket is one address, ~20 ns per workgroup at the memory side, and every workgroup waits a round trip for its own
// before it retires: the full reductions with their 2049 workgroups lost 5 us to it (sum of 10^8: 63 -> 68 us) and keep
// the one-workgroup second kernel at that size; up to kFoldInKernelMaxBlocks workgroups they fold in-kernel, as do
// the selection passes (np_select.hip), whose second step walks a histogram.
template <typename T>
__device__ __forceinline__ T coherent_load(const T *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ void coherent_store(T *p, T v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// true in every thread of the last of `participants` workgroups to get here (all threads of a workgroup must call it)
__device__ __forceinline__ bool last_workgroup_done(unsigned *ticket, unsigned participants) {
    __shared__ unsigned s_last;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == participants - 1 ? 1u : 0u;
    __syncthreads();
    return s_last != 0;
}

// Workgroup reduce of one value per thread (256 threads = 4 waves).  Result valid in thread 0.
template <int OP>
__device__ __forceinline__ float block_reduce(float v, float *lds4) {
    v = wave_reduce<OP>(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) lds4[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        v = lds4[0];
        const int nw = (blockDim.x + 63) >> 6;
        for (int w = 1; w < nw; ++w) v = r_combine<OP>(v, lds4[w]);
    }
    return v;
}

// The end of a streaming first pass: `r` is this workgroup's partial (valid in thread 0).  With a ticket, the last
// workgroup to finish folds all gridDim.x partials — thread t takes partials t, t + blockDim, ... and block_reduce joins
// them, the order the separate one-workgroup kernel uses, whichever workgroup happens to be last — and writes out[0]
// (divided by mean_count for NP_MEAN).  Without one (ticket == nullptr) the partial is just stored for that kernel.
// All threads of every workgroup call it.
template <int OP>
__device__ __forceinline__ void fold_in_last_workgroup(float r, float *partials, unsigned *ticket, float *out,
                                                       float mean_count, float *lds4) {
    if (!ticket) {
        if (threadIdx.x == 0) partials[blockIdx.x] = r;
        return;
    }
    if (gridDim.x == 1) {
        if (threadIdx.x == 0) {
            if constexpr (OP == NP_MEAN) r = __fdiv_rn(r, mean_count);
            out[0] = r;
        }
        return;
    }
    if (threadIdx.x == 0) coherent_store(&partials[blockIdx.x], r);
    if (!last_workgroup_done(ticket, gridDim.x)) return;
    // thread t folds partials t, t + blockDim, ... in that order (the order of the one-workgroup second kernel).  The loads
    // are memory-side atomics, a microsecond or two each: eight are issued before the first is used (lanes past the end load
    // nothing and fold the identity, which changes no bit: x + 0, x * 1, min(x, +inf), max(x, -inf))
    float f = r_identity<OP>();
    for (unsigned base = 0; base < gridDim.x; base += 8 * blockDim.x) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned i = base + u * blockDim.x + threadIdx.x;
            v[u] = i < gridDim.x ? coherent_load(&partials[i]) : r_identity<OP>();
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) f = r_combine<OP>(f, v[u]);
    }
    f = block_reduce<OP>(f, lds4);
    if (threadIdx.x == 0) {
        if constexpr (OP == NP_MEAN) f = __fdiv_rn(f, mean_count);
        out[0] = f;
        coherent_store(ticket, 0u);   // clean for the launch that draws this slot next
    }
}

}  // namespace dev
}  // namespace np

#define NP_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess)                                                           \
            return np::fail(NP_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

#define NP_LAUNCH_CHECK(name)                                                           \
    do {                                                                                \
        __atomic_fetch_add(&np::g_launch_count, 1ull, __ATOMIC_RELAXED);                \
        hipError_t _e = hipGetLastError();                                              \
        if (_e != hipSuccess)                                                           \
            return np::fail(NP_ERR_DEVICE, "launch of %s failed: %s", name,             \
                            hipGetErrorString(_e));                                     \
    } while (0)

#endif
