// Order statistics on the device: the k-th and (k+1)-th smallest of an fp32 array, exact, by radix
// select — what `NDArray::median` / `NDArray::quantile` need.  The reference copies the array and
// qsort()s it on the host (src/ndmath/arithmetics.c:111-138 calculate_median,
// src/ndmath/statistics.c:14-50 calculate_quantile; both refuse device arrays), i.e. O(n log n)
// single-threaded; the two order statistics it then reads are all the result depends on.
//
// Keys: the usual order-preserving map of IEEE bits to unsigned (negative: ~bits, else bits | sign), so
// -0.0 sorts before +0.0 (the reference's comparator calls them equal and leaves their order to qsort)
// and NaNs sort to the ends by sign (the comparator returns 0 for any NaN pair: qsort's result is then
// unspecified, nothing to match).
//
// Three passes over the array, most significant digit first (11 + 11 + 10 bits): each pass histograms
// the digit of the elements that match the prefix chosen so far — per-workgroup LDS histograms, four
// copies selected by lane so that the few hot bins of real data (half of U[0,1) shares one exponent)
// do not serialise a whole wave on one LDS address — then one small workgroup picks the bin holding
// rank k.  The successor (rank k+1) costs no extra pass: it is either in the same bin all the way
// down (then the last histogram names it), or it is the smallest key of the next non-empty bin at the
// level where the two ranks part, which the following pass finds with one compare + min per element
// (or names directly, at the last level).  12 B/elem of HBM reads in total; no sort, no copy.
#include "np_internal.h"

namespace {

constexpr int BINS = 2048;
constexpr int COPIES = 4;

struct SelectState {
    unsigned long long k;        // rank still to find inside the current prefix group
    unsigned prefix;             // key bits fixed so far (the rest zero)
    unsigned succ_prefix;        // mode 1: prefix of the group whose minimum is the successor
    unsigned succ_min;           // running minimum key of that group (atomicMin)
    unsigned succ_key;           // mode 2: the successor's key
    int mode;                    // 0: successor still in the same bin; 1: group known; 2: key known; 3: none
};

__device__ __forceinline__ unsigned to_key(float x) {
    const unsigned u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float from_key(unsigned key) {
    return __uint_as_float((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
}

// PASS 0: bits 31..21, PASS 1: bits 20..10, PASS 2: bits 9..0
template <int PASS>
struct Digit {
    static constexpr int shift = PASS == 0 ? 21 : PASS == 1 ? 10 : 0;
    static constexpr unsigned bins = PASS == 2 ? 1024u : 2048u;
    static constexpr unsigned above = PASS == 0 ? 0u : PASS == 1 ? 0xffe00000u : 0xfffffc00u;   // bits already fixed
};

template <int PASS, typename I>
__global__ __launch_bounds__(256) void select_hist_kernel(const float *__restrict__ in, I n, SelectState *__restrict__ st,
                                                          unsigned long long *__restrict__ hist) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef v4f v4f_u __attribute__((aligned(4)));
    __shared__ unsigned h[COPIES][BINS];
    __shared__ unsigned smin[4];
    for (unsigned i = threadIdx.x; i < COPIES * BINS; i += 256) (&h[0][0])[i] = 0;
    __syncthreads();
    const unsigned prefix = st->prefix;
    const bool want_succ = PASS > 0 && st->mode == 1;
    const unsigned succ_prefix = st->succ_prefix;
    unsigned *mine = h[threadIdx.x & (COPIES - 1)];
    unsigned local_min = 0xffffffffu;
    const unsigned lane = threadIdx.x & 63;
    // Duplicate-heavy data (half of a ReLU output is one value; a constant array is the extreme) sends most
    // of a wave to ONE LDS address, which the LDS serialises (constant 10^8: 0.55 ms).  Voting — the first
    // hitting lane's bin is counted once for all lanes that share it — fixes that (0.30 ms) but costs random
    // data 10 %, so every trip votes on its first element only and lets the result (>= 24 lanes agreed)
    // decide how the trip's other elements are counted.
    bool vote = false;   // wave-uniform
    auto take = [&](float x, bool probe) {
        const unsigned key = to_key(x);
        const bool hit = PASS == 0 || (key & Digit<PASS>::above) == prefix;
        const unsigned bin = (key >> Digit<PASS>::shift) & (Digit<PASS>::bins - 1);
        if (probe || vote) {
            const unsigned long long hits = __ballot(hit);
            if (hits) {
                const int leader = __ffsll((long long)hits) - 1;
                const unsigned hot = (unsigned)__shfl((int)bin, leader, 64);
                const unsigned long long same = __ballot(hit && bin == hot);
                if ((int)lane == leader) atomicAdd(&mine[hot], (unsigned)__popcll(same));
                else if (hit && bin != hot) atomicAdd(&mine[bin], 1u);
                if (probe) vote = __popcll(same) >= 24;
            } else if (probe) {
                vote = false;
            }
        } else if (hit) {
            atomicAdd(&mine[bin], 1u);
        }
        if (PASS > 0 && want_succ && (key & Digit<PASS>::above) == succ_prefix) local_min = min(local_min, key);
    };
    const I nvec = n / 4;
    const I stride = (I)gridDim.x * 256;
    I v = (I)blockIdx.x * 256 + threadIdx.x;
    for (; v + 3 * stride < nvec; v += 4 * stride) {   // four loads in flight per lane
        v4f x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) x[u] = __builtin_nontemporal_load((const v4f_u *)(in + (size_t)(v + u * stride) * 4));
#pragma unroll
        for (int u = 0; u < 4; ++u) { take(x[u][0], u == 0); take(x[u][1], false); take(x[u][2], false); take(x[u][3], false); }
    }
    for (; v < nvec; v += stride) {
        const v4f x = __builtin_nontemporal_load((const v4f_u *)(in + (size_t)v * 4));
        take(x[0], true); take(x[1], false); take(x[2], false); take(x[3], false);
    }
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)(n - nvec * 4)) take(in[(size_t)nvec * 4 + threadIdx.x], true);
    __syncthreads();
    for (unsigned b = threadIdx.x; b < Digit<PASS>::bins; b += 256) {
        const unsigned c = h[0][b] + h[1][b] + h[2][b] + h[3][b];
        if (c) atomicAdd(&hist[b], (unsigned long long)c);
    }
    if (PASS > 0 && want_succ) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) local_min = min(local_min, (unsigned)__shfl_down((int)local_min, off, 64));
        if ((threadIdx.x & 63) == 0) smin[threadIdx.x >> 6] = local_min;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned m = min(min(smin[0], smin[1]), min(smin[2], smin[3]));
            if (m != 0xffffffffu) atomicMin(&st->succ_min, m);
        }
    }
}

// One workgroup: find the bin holding rank k, descend into it, settle what is known about rank k+1,
// and clear the histogram for the next pass.
template <int PASS>
__global__ __launch_bounds__(256) void select_scan_kernel(SelectState *__restrict__ st, unsigned long long *__restrict__ hist,
                                                          float *__restrict__ out2) {
    constexpr unsigned bins = Digit<PASS>::bins, per = bins / 256;
    __shared__ unsigned long long part[2][256];
    __shared__ unsigned long long cnt[BINS];
    unsigned long long sum = 0;
    for (unsigned j = 0; j < per; ++j) {
        const unsigned b = threadIdx.x * per + j;
        cnt[b] = hist[b];
        hist[b] = 0;
        sum += cnt[b];
    }
    // inclusive scan of the 256 per-thread sums (Hillis-Steele, double buffered); the thread whose
    // range of ranks holds k carries on alone (a serial walk by thread 0 cost 15-35 us per level)
    int cur = 0;
    part[0][threadIdx.x] = sum;
    __syncthreads();
    for (unsigned off = 1; off < 256; off <<= 1) {
        unsigned long long v = part[cur][threadIdx.x];
        if (threadIdx.x >= off) v += part[cur][threadIdx.x - off];
        part[cur ^ 1][threadIdx.x] = v;
        __syncthreads();
        cur ^= 1;
    }
    const unsigned long long k = st->k;
    const unsigned long long incl = part[cur][threadIdx.x];
    unsigned long long below = incl - sum;
    if (!(below <= k && k < incl)) return;   // exactly one thread stays: the group holds more than k elements
    unsigned b = threadIdx.x * per;
    while (b < bins - 1 && below + cnt[b] <= k) below += cnt[b++];
    const unsigned long long in_bin = cnt[b];
    const unsigned old_prefix = st->prefix;
    const unsigned prefix = old_prefix | (b << Digit<PASS>::shift);
    st->prefix = prefix;
    st->k = k - below;
    int mode = st->mode;
    unsigned succ_key = st->succ_key;
    if (PASS > 0 && mode == 1) {   // the group named one level up was scanned by this pass: its minimum is the successor
        succ_key = st->succ_min;
        mode = 2;
    }
    if (mode == 0 && k + 1 >= below + in_bin) {   // ranks k and k+1 part at this level
        unsigned nb = b + 1;
        while (nb < bins && cnt[nb] == 0) ++nb;
        if (nb == bins) mode = 3;   // first level only (deeper, mode 0 says the successor shares the parent bin): k is the maximum
        else if (PASS == 2) {
            succ_key = old_prefix | nb;   // last level: the bin IS the key
            mode = 2;
        } else {
            st->succ_prefix = old_prefix | (nb << Digit<PASS>::shift);
            st->succ_min = 0xffffffffu;
            mode = 1;
        }
    }
    st->mode = mode;
    st->succ_key = succ_key;
    if (PASS == 2) {
        if (mode != 2) succ_key = prefix;   // same value again (duplicates), or no successor at all
        out2[0] = from_key(prefix);
        out2[1] = from_key(succ_key);
    }
}

__global__ void select_init_kernel(SelectState *st, unsigned long long *hist, unsigned long long k) {
    for (unsigned b = threadIdx.x; b < BINS; b += blockDim.x) hist[b] = 0;
    if (threadIdx.x == 0) {
        st->k = k;
        st->prefix = 0;
        st->succ_prefix = 0;
        st->succ_min = 0xffffffffu;
        st->succ_key = 0;
        st->mode = 0;
    }
}

template <typename I>
int run_select(const float *in, size_t n, size_t k, float *dev_out2) {
    hipStream_t s = np::stream();
    np::Scratch buf;
    if (int rc = buf.alloc(256 + BINS * sizeof(unsigned long long))) return rc;
    SelectState *st = (SelectState *)buf.ptr;
    unsigned long long *hist = (unsigned long long *)((char *)buf.ptr + 256);
    size_t blocks = (n / 4 + 255) / 256;
    const size_t cap = (size_t)np::num_cus() * 4;   // 32 KB of LDS histograms per workgroup
    if (blocks > cap) blocks = cap;                 // (LDS-atomic bound: an odd count, np::capped_grid, buys nothing here)
    if (blocks < 1) blocks = 1;
    select_init_kernel<<<1, 256, 0, s>>>(st, hist, (unsigned long long)k);
    NP_LAUNCH_CHECK("select_init_kernel");
    select_hist_kernel<0, I><<<(unsigned)blocks, 256, 0, s>>>(in, (I)n, st, hist);
    select_scan_kernel<0><<<1, 256, 0, s>>>(st, hist, dev_out2);
    select_hist_kernel<1, I><<<(unsigned)blocks, 256, 0, s>>>(in, (I)n, st, hist);
    select_scan_kernel<1><<<1, 256, 0, s>>>(st, hist, dev_out2);
    select_hist_kernel<2, I><<<(unsigned)blocks, 256, 0, s>>>(in, (I)n, st, hist);
    select_scan_kernel<2><<<1, 256, 0, s>>>(st, hist, dev_out2);
    NP_LAUNCH_CHECK("select kernels");
    return NP_OK;
}

}  // namespace

extern "C" {

int np_order_stat_dev(const float *in, size_t n, size_t k, float *dev_out2) {
    if (n == 0) return np::fail(NP_ERR_INVALID, "np_order_stat: empty array");
    if (k >= n) return np::fail(NP_ERR_INVALID, "np_order_stat: rank %zu out of range for %zu elements", k, n);
    if (!in || !dev_out2) return np::fail(NP_ERR_INVALID, "np_order_stat: null pointer");
    if (int rc = np::ensure_init()) return rc;
    if (n < (size_t(1) << 31)) return run_select<uint32_t>(in, n, k, dev_out2);
    return run_select<uint64_t>(in, n, k, dev_out2);
}

int np_order_stat(const float *in, size_t n, size_t k, float *host_out2) {
    if (!host_out2) return np::fail(NP_ERR_INVALID, "np_order_stat: null output");
    if (int rc = np::ensure_init()) return rc;
    float *slot = np::result_slots();
    if (!slot) return NP_ERR_ALLOC;
    if (int rc = np_order_stat_dev(in, n, k, slot)) return rc;
    if (int rc = np::result_wait()) return rc;
    host_out2[0] = slot[0];
    host_out2[1] = slot[1];
    return NP_OK;
}

}  // extern "C"
