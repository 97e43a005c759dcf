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
//
// Three paths, by size:
//   n <= 32768      select_small_kernel: one workgroup, one launch, the array read once into registers.
//   n <  2^26       the three passes above (each pass's bin-picking step runs in the pass's last workgroup, so a
//                   selection is an init kernel + three launches).
//   n >= 2^26       the "bracket" path in front of them.  An evenly spaced 2^20-element SAMPLE (1024 runs of 4 KiB) is
//                   histogrammed at two levels (22 key bits) and names the narrow key range that can hold rank k (the
//                   sample ranks k*m/n -+ 4096, eight standard deviations of a binomial rank: about 1 % of the data).
//                   ONE pass over the array then counts the keys below that bracket and copies the keys inside it into
//                   per-WAVE segments of a scratch buffer (fill count in a scalar register: no atomics, no LDS).  If
//                   rank k and its successor did land inside the bracket — checked exactly, on the device — the radix
//                   passes run over the copied keys instead of the array: 4 B/elem + small instead of 12 B/elem.  If
//                   not (adversarial order, or a value so often repeated that the bracket would be more than a quarter
//                   of the data, in which case the copy is skipped altogether), the three passes read the array as
//                   before: the result never depends on the sample, only the time does.  The path's launches cost
//                   ~125 us before the first byte, against ~35 us + 12 B/elem for the plain passes: measured crossover
//                   5-6 * 10^7 elements (profiles/r02/select_threshold.log).
#include "np_internal.h"

namespace {

constexpr int BINS = 2048;
constexpr unsigned SMALL_MAX = 32768;   // arrays up to here: select_small_kernel, one workgroup, one launch
constexpr int COPIES = 4;
// Threads per workgroup of the histogram passes: one 16-wave workgroup per CU keeps as many waves resident as four
// 4-wave ones but adds a quarter as many LDS histograms to the global one — up to 2048 device-scope atomics per
// workgroup per pass, which at 1024 workgroups (2 M atomics) cost a pass 10-20 us.
constexpr int HT = 1024;
// Workgroups add their LDS histogram to one of HCOPIES global copies (blockIdx % HCOPIES).  Device-scope atomics on
// ONE address serialise at the memory side (the eight XCD L2s are not coherent; ~20 ns each: 1024 workgroups that
// finish together and add to the same hot bin wait ~20 us), but the streaming passes' workgroups do not finish
// together, and 8 copies cost every scan kernel 3-4 us: measured a wash, so one copy.  The kernels that DID suffer
// (the sample pass, the passes over a thin bracket) now run on few workgroups instead.
constexpr int HCOPIES = 1;

__device__ int g_last_path;   // the last selection: 0 read the array, 1 the bracket's copied keys, 2 ran in select_small_kernel (np_select_last_path)

struct SelectState {
    unsigned long long k;        // rank still to find inside the current prefix group
    unsigned prefix;             // key bits fixed so far (the rest zero)
    unsigned succ_prefix;        // mode 1: prefix of the group whose minimum is the successor
    unsigned succ_min;           // running minimum key of that group (atomicMin)
    unsigned succ_key;           // mode 2: the successor's key
    int mode;                    // 0: successor still in the same bin; 1: group known; 2: key known; 3: none
    // bracket path
    unsigned lo_key;             // bracket = keys with key - lo_key <= width
    unsigned width;
    int bracket;                 // the sample produced a usable bracket: the filter pass runs
    int overflow;                // a wave's segment filled up: the copy is incomplete
    int use_compact;             // verdict of decide(): the radix passes read the segments
    unsigned group;              // radix passes over the copied keys: segments per wave
    unsigned nbits;              // ... which are ranked by (key - lo_key) << (32 - nbits), nbits = bit length of width:
                                 //     pass p is needed only while nbits > 11 * p (32 and a plain key when reading the array)
    unsigned shift;              // second sample level: bin = (key - lo_key) >> shift
    unsigned long long sample_below;   // sample keys below the first-level bracket
    unsigned ticket[6];          // workgroups that have finished: sample 0, sample 1, filter, radix pass 0, 1, 2
};

// The single-workgroup step that follows every pass (pick the bin, settle the bracket, give the verdict) runs in the
// LAST workgroup of that pass to finish instead of in a kernel of its own (np::dev::last_workgroup_done, np_internal.h:
// device-scope atomics carry what crosses workgroups, no device-scope fences): a selection had nine such kernels.
using np::dev::coherent_load;
using np::dev::coherent_store;
using np::dev::last_workgroup_done;

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

// The last workgroup of a radix pass: find the bin holding rank k, descend into it, settle what is known about rank
// k+1, and clear the histogram for the next pass.  Threads 0..255 work, every thread of the workgroup keeps the barriers.
template <int PASS>
__device__ void scan_level(SelectState *__restrict__ st, unsigned long long *__restrict__ hist, float *__restrict__ out2,
                           unsigned long long *__restrict__ lds /* 512 + BINS words */) {
    constexpr unsigned bins = Digit<PASS>::bins, per = bins / 256;
    unsigned long long *part = lds;            // [2][256]
    unsigned long long *cnt = lds + 512;       // [BINS]
    const unsigned t = threadIdx.x;
    const bool worker = t < 256;
    // the state first, so that its round trip overlaps the histogram's
    const unsigned long long k = st->k;
    const unsigned old_prefix = st->prefix, nbits = st->nbits, lo = st->lo_key;
    const int compact = st->use_compact;
    int mode = st->mode;
    unsigned succ_key = st->succ_key;
    const bool final_level = PASS == 2 || (compact && nbits <= 11u * (PASS + 1));   // no key bits left below this digit
    unsigned long long sum = 0;
    if (worker) {
        for (unsigned j = 0; j < per; ++j) {
            const unsigned b = t * per + j;
            unsigned long long c = 0;
#pragma unroll
            for (int h = 0; h < HCOPIES; ++h) {
                c += coherent_load(&hist[h * BINS + b]);
                hist[h * BINS + b] = 0;
            }
            cnt[b] = c;
            sum += c;
        }
        part[t] = sum;
    }
    __syncthreads();
    // inclusive scan of the 256 per-thread sums (Hillis-Steele, double buffered); the thread whose
    // range of ranks holds k carries on alone (a serial walk by thread 0 cost 15-35 us per level)
    int cur = 0;
    for (unsigned off = 1; off < 256; off <<= 1) {
        if (worker) {
            unsigned long long v = part[cur * 256 + t];
            if (t >= off) v += part[cur * 256 + t - off];
            part[(cur ^ 1) * 256 + t] = v;
        }
        __syncthreads();
        cur ^= 1;
    }
    if (!worker) return;
    const unsigned long long incl = part[cur * 256 + t];
    unsigned long long below = incl - sum;
    if (!(below <= k && k < incl)) return;   // exactly one thread stays: the group holds more than k elements
    unsigned b = t * per;
    while (b < bins - 1 && below + cnt[b] <= k) below += cnt[b++];
    const unsigned long long in_bin = cnt[b];
    const unsigned prefix = old_prefix | (b << Digit<PASS>::shift);
    st->prefix = prefix;
    st->k = k - below;
    if (PASS > 0 && mode == 1) {   // the group named one level up was scanned by this pass: its minimum is the successor
        succ_key = coherent_load(&st->succ_min);
        mode = 2;
    }
    if (mode == 0 && k + 1 >= below + in_bin) {   // ranks k and k+1 part at this level
        unsigned nb = b + 1;
        while (nb < bins && cnt[nb] == 0) ++nb;
        if (nb == bins) mode = 3;   // first level only (deeper, mode 0 says the successor shares the parent bin): k is the maximum
        else if (final_level) {
            succ_key = old_prefix | (nb << Digit<PASS>::shift);   // last level: the bin IS the key
            mode = 2;
        } else {
            st->succ_prefix = old_prefix | (nb << Digit<PASS>::shift);
            st->succ_min = 0xffffffffu;
            mode = 1;
        }
    }
    st->mode = mode;
    st->succ_key = succ_key;
    if (final_level) {
        g_last_path = compact;
        if (mode != 2) succ_key = prefix;   // same value again (duplicates), or no successor at all
        const unsigned lshift = 32u - nbits;   // 0 and lo = 0 unless the keys are the bracket's
        out2[0] = from_key(compact ? lo + (prefix >> lshift) : prefix);
        out2[1] = from_key(compact ? lo + (succ_key >> lshift) : succ_key);
    }
}

template <int PASS, typename I>
__global__ __launch_bounds__(HT) void select_hist_kernel(const float *__restrict__ in, I n, SelectState *__restrict__ st,
                                                          unsigned long long *__restrict__ hist, const float *__restrict__ cbuf,
                                                          const unsigned *__restrict__ ccount, unsigned seg_cap, unsigned nseg,
                                                          float *__restrict__ out2) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef v4f v4f_u __attribute__((aligned(4)));
    __shared__ __attribute__((aligned(16))) unsigned h[COPIES][BINS];   // later the scan's 512 + BINS 64-bit words
    static_assert(sizeof(unsigned) * COPIES * BINS >= sizeof(unsigned long long) * (512 + BINS), "scan_level reuses h");
    __shared__ unsigned smin[HT / 64];
    // everything the pass needs from the state in one round trip
    const bool compact = st->use_compact != 0;
    const unsigned group = st->group, nbits = st->nbits, lo = st->lo_key;
    const unsigned prefix = st->prefix, succ_prefix = st->succ_prefix;
    const bool want_succ = PASS > 0 && st->mode == 1;
    if (compact && PASS > 0 && nbits <= 11u * PASS) return;          // the copied keys' range was settled one pass up
    if (compact && blockIdx.x * (HT / 64) * group >= nseg) return;   // this workgroup's waves have no segments
    const unsigned participants = compact ? (nseg + (HT / 64) * group - 1) / ((HT / 64) * group) : gridDim.x;
    const unsigned lshift = 32u - nbits;
    for (unsigned i = threadIdx.x; i < COPIES * BINS; i += HT) (&h[0][0])[i] = 0;
    __syncthreads();
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
        unsigned key = to_key(x);
        if (compact) key = (key - lo) << lshift;   // rank inside the bracket, left-aligned: its top digit is the first pass's
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
    if (compact) {   // bracket path: wave w of the grid reads segments [w * group, (w + 1) * group) of the copied keys
        const unsigned first = (blockIdx.x * (HT / 64) + (threadIdx.x >> 6)) * group;
        const unsigned last = min(first + group, nseg);
        // Every read is a whole float4 (segments are 16-byte aligned and end in slack, so the up to three floats
        // past a segment's count are readable; they are just not counted): a scalar tail would be a chain of
        // dependent loads, ~1 us each, on a pass that has nothing to hide them behind.
        auto take4 = [&](const v4f &x, unsigned off, unsigned cnt) {
            if (off < cnt) take(x[0], true);
            if (off + 1 < cnt) take(x[1], false);
            if (off + 2 < cnt) take(x[2], false);
            if (off + 3 < cnt) take(x[3], false);
        };
        for (unsigned s0 = first; s0 < last; s0 += 8) {   // the first 512 keys of eight segments in flight at once
            unsigned cnt[8];
            v4f x[8][2];
#pragma unroll
            for (unsigned u = 0; u < 8; ++u) {
                cnt[u] = s0 + u < last ? ccount[s0 + u] : 0u;
                const float *seg = cbuf + (size_t)(s0 + u) * seg_cap;
#pragma unroll
                for (unsigned c = 0; c < 2; ++c)
                    if (c * 256 + lane * 4 < cnt[u]) x[u][c] = *(const v4f *)(seg + c * 256 + lane * 4);
            }
#pragma unroll
            for (unsigned u = 0; u < 8; ++u) {
                const float *seg = cbuf + (size_t)(s0 + u) * seg_cap;
#pragma unroll
                for (unsigned c = 0; c < 2; ++c)
                    if (c * 256 + lane * 4 < cnt[u]) take4(x[u][c], c * 256 + lane * 4, cnt[u]);
                for (unsigned off = 512 + lane * 4; off < cnt[u]; off += 256) {   // the rest of a long segment
                    const v4f y = *(const v4f *)(seg + off);
                    take4(y, off, cnt[u]);
                }
            }
        }
    } else {
        const I nvec = n / 4;
        const I stride = (I)gridDim.x * HT;
        I v = (I)blockIdx.x * HT + threadIdx.x;
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
    }
    __syncthreads();
    for (unsigned b = threadIdx.x; b < Digit<PASS>::bins; b += HT) {
        const unsigned c = h[0][b] + h[1][b] + h[2][b] + h[3][b];
        if (c) atomicAdd(&hist[(blockIdx.x % HCOPIES) * BINS + b], (unsigned long long)c);
    }
    if (PASS > 0 && want_succ) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) local_min = min(local_min, (unsigned)__shfl_down((int)local_min, off, 64));
        if ((threadIdx.x & 63) == 0) smin[threadIdx.x >> 6] = local_min;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned m = smin[0];
            for (unsigned w = 1; w < HT / 64; ++w) m = min(m, smin[w]);
            if (m != 0xffffffffu) atomicMin(&st->succ_min, m);
        }
    }
    if (last_workgroup_done(&st->ticket[3 + PASS], participants)) scan_level<PASS>(st, hist, out2, (unsigned long long *)&h[0][0]);
}


// ---- bracket path ------------------------------------------------------------------------------------
constexpr unsigned SAMPLE_RUNS = 1024;                    // 4 KiB runs of the sample, evenly spaced
constexpr unsigned SAMPLE_WGS = 64;                       // workgroups of a sample pass, SAMPLE_RUNS / SAMPLE_WGS runs each
constexpr unsigned long long SAMPLE_M = SAMPLE_RUNS * 1024ull;
constexpr unsigned SAMPLE_DELTA = 4096;                   // 8 sigma of a binomial rank at m = 2^20

// The last workgroup of a sample pass (256 threads): the bins holding sample ranks ks - delta and ks + delta bound the bracket (LEVEL 0: top-level
// bins; LEVEL 1: the same ranks again inside the first bracket, 2048 times finer).
template <int LEVEL>
__device__ void bracket_level(SelectState *__restrict__ st, unsigned long long *__restrict__ hist, unsigned long long n) {
    constexpr unsigned per = BINS / 256;
    __shared__ unsigned long long part[2][256];
    __shared__ unsigned long long edge[4];   // b_lo, keys below b_lo, b_hi, keys up to and including b_hi
    unsigned long long cnt[per], sum = 0;
#pragma unroll
    for (unsigned j = 0; j < per; ++j) {
        unsigned long long c = 0;
#pragma unroll
        for (int h = 0; h < HCOPIES; ++h) {
            c += coherent_load(&hist[h * BINS + threadIdx.x * per + j]);
            hist[h * BINS + threadIdx.x * per + j] = 0;
        }
        cnt[j] = c;
        sum += cnt[j];
    }
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
    // k * m / n without overflow: k < 2^44 for any array that fits the device
    const unsigned long long ks = (unsigned long long)((unsigned __int128)st->k * SAMPLE_M / n);
    const unsigned long long skip = LEVEL ? st->sample_below : 0ull;   // ranks are counted from the first bracket's start
    const unsigned long long r_lo = (ks > SAMPLE_DELTA ? ks - SAMPLE_DELTA : 0) - skip;
    const unsigned long long r_hi = (ks + SAMPLE_DELTA < SAMPLE_M ? ks + SAMPLE_DELTA : SAMPLE_M - 1) - skip;
    // A rank within delta of either end of the sample: the array may hold keys beyond the sample's extremes.  The two
    // levels still narrow the bracket down inside the sample's range (sample rank 0 / m-1 = its first / last occupied
    // bin); the finished bracket is then opened to key 0 / 0xffffffff, which adds only what the sample never saw.
    const bool lo_open = ks <= SAMPLE_DELTA, hi_open = ks + SAMPLE_DELTA >= SAMPLE_M - 1;
    const unsigned long long incl = part[cur][threadIdx.x], excl = incl - sum;
    if (excl <= r_lo && r_lo < incl) {
        unsigned long long below = excl;
        unsigned j = 0;
#pragma unroll
        for (unsigned t = 0; t < per - 1; ++t)
            if (j == t && below + cnt[t] <= r_lo) { below += cnt[t]; j = t + 1; }
        edge[0] = threadIdx.x * per + j;
        edge[1] = below;
    }
    if (excl <= r_hi && r_hi < incl) {
        unsigned long long upto = excl;
        unsigned j = 0;
#pragma unroll
        for (unsigned t = 0; t < per - 1; ++t)
            if (j == t && upto + cnt[t] <= r_hi) { upto += cnt[t]; j = t + 1; }
        unsigned long long last = 0;
#pragma unroll
        for (unsigned t = 0; t < per; ++t) last = j == t ? cnt[t] : last;
        edge[2] = threadIdx.x * per + j;
        edge[3] = upto + last;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned b_lo = (unsigned)edge[0], b_hi = (unsigned)edge[2], span = b_hi - b_lo + 1;
        if (LEVEL == 0) {
            unsigned up = 0;                       // ceil(log2(span)): the second level spreads the bracket over <= 2048 bins
            while ((1u << up) < span) ++up;
            st->lo_key = b_lo << 21;
            st->width = (unsigned)(((unsigned long long)span << 21) - 1ull);   // ends at 0xffffffff at most (b_hi <= 2047)
            st->shift = 10 + up;
            st->sample_below = edge[1];
        } else {
            const unsigned shift = st->shift;
            const unsigned long long inside = edge[3] - edge[1];
            // the finer bracket never reaches past the first one (whose bins need not fill all 2048 second-level slots)
            const unsigned long long old_lo = st->lo_key, old_end = old_lo + st->width;
            const unsigned long long new_lo = old_lo + ((unsigned long long)b_lo << shift);
            unsigned long long new_end = new_lo + ((unsigned long long)span << shift) - 1ull;
            if (new_end > old_end) new_end = old_end;
            const unsigned long long open_lo = lo_open ? 0ull : new_lo, open_end = hi_open ? 0xffffffffull : new_end;
            st->lo_key = (unsigned)open_lo;
            st->width = (unsigned)(open_end - open_lo);
            // the copy pays only while the bracket is a small part of the data (one value repeated over a quarter
            // of the array cannot be bracketed any tighter: the plain passes take over)
            st->bracket = inside * 4 <= SAMPLE_M ? 1 : 0;
        }
    }
}

// LEVEL 0: the 11 top key bits of every sample element.  LEVEL 1: the sample elements inside the first-level
// bracket, 11 bits further down (bin = (key - lo_key) >> shift): 1/2048 of a top-level bin instead of a whole one.
template <int LEVEL, typename I>
__global__ __launch_bounds__(256) void select_sample_kernel(const float *__restrict__ in, I n, SelectState *__restrict__ st,
                                                            unsigned long long *__restrict__ hist) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef v4f v4f_u __attribute__((aligned(4)));
    __shared__ unsigned h[BINS];
    for (unsigned i = threadIdx.x; i < BINS; i += 256) h[i] = 0;
    __syncthreads();
    const unsigned lo = LEVEL ? st->lo_key : 0u, width = LEVEL ? st->width : 0xffffffffu, shift = LEVEL ? st->shift : 21u;
    const unsigned long long nvec = n / 4;
    constexpr unsigned R = SAMPLE_RUNS / SAMPLE_WGS;
    v4f x[R];
#pragma unroll
    for (unsigned r = 0; r < R; ++r) {
        const unsigned long long start = (unsigned long long)(blockIdx.x * R + r) * (nvec - 256) / (SAMPLE_RUNS - 1);   // last run ends at the end
        x[r] = __builtin_nontemporal_load((const v4f_u *)(in + (size_t)(start + threadIdx.x) * 4));
    }
#pragma unroll
    for (unsigned r = 0; r < R; ++r) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // the first hitting lane's bin is counted once for every lane that shares it
            const unsigned d = to_key(x[r][j]) - lo;
            const bool hit = d <= width;
            const unsigned bin = d >> shift;
            const unsigned long long hits = __ballot(hit);
            if (hits) {
                const int leader = __ffsll((long long)hits) - 1;
                const unsigned hot = (unsigned)__shfl((int)bin, leader, 64);
                const unsigned long long same = __ballot(hit && bin == hot);
                if ((int)(threadIdx.x & 63) == leader) atomicAdd(&h[hot], (unsigned)__popcll(same));
                else if (hit && bin != hot) atomicAdd(&h[bin], 1u);
            }
        }
    }
    __syncthreads();
    for (unsigned b = threadIdx.x; b < BINS; b += 256)
        if (h[b]) atomicAdd(&hist[(blockIdx.x % HCOPIES) * BINS + b], (unsigned long long)h[b]);
    if (last_workgroup_done(&st->ticket[LEVEL], gridDim.x)) bracket_level<LEVEL>(st, hist, (unsigned long long)n);
}

// The last workgroup of the filter pass (256 threads): did rank k and its successor land in the copied keys?  Then the
// radix passes read those.
__device__ void decide(SelectState *__restrict__ st, const unsigned *__restrict__ ccount, const unsigned *__restrict__ cbelow,
                       unsigned waves, unsigned long long n) {
    __shared__ unsigned long long part[2][256];
    unsigned long long in_sum = 0, below_sum = 0;
    for (unsigned w = threadIdx.x; w < waves; w += 256) {
        in_sum += coherent_load(&ccount[w]);
        below_sum += coherent_load(&cbelow[w]);
    }
    part[0][threadIdx.x] = in_sum;
    part[1][threadIdx.x] = below_sum;
    __syncthreads();
    for (unsigned off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            part[0][threadIdx.x] += part[0][threadIdx.x + off];
            part[1][threadIdx.x] += part[1][threadIdx.x + off];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const unsigned long long k = st->k, inside = part[0][0], below = part[1][0], end = below + inside;
        const bool ok = !coherent_load(&st->overflow) && below <= k && k < end && (k + 1 < end || k + 1 == n);
        if (ok) {
            st->k = k - below;
            st->use_compact = 1;
            st->nbits = 32u - (unsigned)__clz((int)(st->width | 1u));
            // segments per wave of the radix passes: about 2048 keys per wave, so that a thin bracket is not spread
            // over every workgroup (each pays the pass's fixed costs and adds its few counts to the same bins)
            const unsigned long long per_seg = inside / waves + 1;
            unsigned group = 1;
            while (group < 32 && per_seg * group < 2048) group *= 2;
            st->group = group;
        }
    }
}

// One pass over the array: count the keys below the bracket, copy the values inside it into this WAVE's segment.
// A wave's fill count is a scalar register — no atomics, no LDS — so every trip is executed by the whole
// wave (validity is a per-lane predicate, never a per-lane loop exit).
template <typename I>
__global__ __launch_bounds__(256) void select_filter_kernel(const float *__restrict__ in, I n, SelectState *__restrict__ st,
                                                            float *__restrict__ cbuf, unsigned *__restrict__ ccount,
                                                            unsigned *__restrict__ cbelow, unsigned seg_cap) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef v4f v4f_u __attribute__((aligned(4)));
    if (!st->bracket) return;   // every workgroup alike: nobody takes a ticket, nobody decides
    const unsigned lo = st->lo_key, width = st->width;
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    float *seg = cbuf + (size_t)wave * seg_cap;
    const unsigned lane = threadIdx.x & 63;
    unsigned below = 0;          // per lane
    unsigned fill = 0;           // wave-uniform
    bool over = false;           // wave-uniform
    // E elements per lane at a time: their in-bracket masks first (wave-uniform), then the stores behind the fill count
    auto trip = [&](const float *x, bool valid, auto count) {
        constexpr int E = decltype(count)::value;
        unsigned long long mask[E];
        bool inb[E];
        unsigned total = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const unsigned key = to_key(x[e]);
            below += (valid && key < lo) ? 1u : 0u;
            inb[e] = valid && key - lo <= width;
            mask[e] = __ballot(inb[e]);
            total += (unsigned)__popcll(mask[e]);
        }
        if (total == 0) return;
        unsigned base = fill;
        fill += total;
        if (fill > seg_cap) { over = true; return; }
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (mask[e] == 0) continue;
            const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mask[e] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask[e], 0u));
            if (inb[e]) seg[base + rank] = x[e];
            base += (unsigned)__popcll(mask[e]);
        }
    };
    const I nvec = n / 4;
    const I stride = (I)gridDim.x * 256;
    I vb = (I)blockIdx.x * 256 + (threadIdx.x & ~63u);   // the wave's first vector: wave-uniform loop control
    if (vb + 63 + 3 * stride < nvec) {   // the next four loads are in flight while this trip's sixteen elements are handled
        float x[16], y[16];
#pragma unroll
        for (int u = 0; u < 4; ++u) *(v4f *)(x + 4 * u) = __builtin_nontemporal_load((const v4f_u *)(in + (size_t)(vb + lane + u * stride) * 4));
        for (;;) {
            vb += 4 * stride;
            const bool more = vb + 63 + 3 * stride < nvec;
            if (more) {
#pragma unroll
                for (int u = 0; u < 4; ++u) *(v4f *)(y + 4 * u) = __builtin_nontemporal_load((const v4f_u *)(in + (size_t)(vb + lane + u * stride) * 4));
            }
            trip(x, true, std::integral_constant<int, 16>());
            if (!more) break;
#pragma unroll
            for (int e = 0; e < 16; ++e) x[e] = y[e];
        }
    }
    for (; vb < nvec; vb += stride) {
        const bool valid = vb + lane < nvec;
        float x[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (valid) *(v4f *)x = __builtin_nontemporal_load((const v4f_u *)(in + (size_t)(vb + lane) * 4));
        trip(x, valid, std::integral_constant<int, 4>());
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {   // the array's last n % 4 elements
        const bool valid = lane < (unsigned)(n - nvec * 4);
        const float x = valid ? in[(size_t)nvec * 4 + lane] : 0.0f;
        trip(&x, valid, std::integral_constant<int, 1>());
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) below += (unsigned)__shfl_down((int)below, off, 64);
    if (lane == 0) {   // per-wave results; select_decide_kernel adds them up (4096 atomics on st->below cost 80 us)
        coherent_store(&ccount[wave], over ? 0u : fill);
        coherent_store(&cbelow[wave], below);
        if (over) coherent_store(&st->overflow, 1);
    }
    // (a verdict kernel of its own behind this pass measured the same: 0.148 vs 0.149 ms for the whole selection)
    if (last_workgroup_done(&st->ticket[2], gridDim.x)) decide(st, ccount, cbelow, gridDim.x * 4, (unsigned long long)n);
}

__global__ void select_init_kernel(SelectState *st, unsigned long long *hist, unsigned long long k) {
    for (unsigned b = threadIdx.x; b < BINS * HCOPIES; b += blockDim.x) hist[b] = 0;
    if (threadIdx.x == 0) {
        st->k = k;
        st->prefix = 0;
        st->succ_prefix = 0;
        st->succ_min = 0xffffffffu;
        st->succ_key = 0;
        st->mode = 0;
        st->lo_key = 0;
        st->width = 0;
        st->bracket = 0;
        st->overflow = 0;
        st->use_compact = 0;
        st->group = 1;
        st->nbits = 32;
        for (int i = 0; i < 6; ++i) st->ticket[i] = 0;
        st->shift = 0;
        st->sample_below = 0;
    }
}

size_t g_small_max = SMALL_MAX;             // np_select_set_variant(0) also switches the one-workgroup kernel off
size_t g_filter_blocks = 0;                 // np_select_set_variant(2..2047): workgroups of the streaming passes (0 = 4 per CU)
size_t g_bracket_min_n = size_t(1) << 26;   // np_select_set_variant: 0 switches the bracket path off


// ---- small arrays: one workgroup, one launch ---------------------------------------------------------
// Up to 32768 elements the selection is a chain of launches and round trips, not of bytes (four launches: 36 us for
// 1024 floats).  One workgroup of 1024 threads reads the array ONCE into registers (<= 32 keys per thread), then
// selects byte by byte, most significant first: a 256-bin LDS histogram of the keys that match the prefix so far, one
// wave scans it (4 bins per lane + a shuffle scan).  After four bytes the prefix IS the k-th key; its successor is the
// same key again if the last bin holds rank k+1 too, else the smallest key above it (one min-reduce over the
// registers), else — k is the maximum — the key itself, as on the large path.

__global__ __launch_bounds__(1024) void select_small_kernel(const float *__restrict__ in, unsigned n, unsigned k,
                                                            float *__restrict__ out2) {
    constexpr unsigned PER = SMALL_MAX / 1024;
    __shared__ unsigned h[256];
    __shared__ unsigned s_bin, s_below, s_in_bin, s_min[16];
    unsigned key[PER];
#pragma unroll
    for (unsigned j = 0; j < PER; ++j) {
        const unsigned i = j * 1024 + threadIdx.x;
        key[j] = i < n ? to_key(in[i]) : 0u;
    }
    const unsigned lane = threadIdx.x & 63;
    unsigned prefix = 0, mask = 0, rank = k, in_bin = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (threadIdx.x < 256) h[threadIdx.x] = 0;
        __syncthreads();
#pragma unroll
        for (unsigned j = 0; j < PER; ++j)
            if (j * 1024 + threadIdx.x < n && (key[j] & mask) == prefix) atomicAdd(&h[(key[j] >> shift) & 255u], 1u);
        __syncthreads();
        if (threadIdx.x < 64) {   // wave 0: lane l owns bins 4l .. 4l+3
            const unsigned c0 = h[4 * lane], c1 = h[4 * lane + 1], c2 = h[4 * lane + 2], c3 = h[4 * lane + 3];
            const unsigned sum = c0 + c1 + c2 + c3;
            unsigned incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned up = (unsigned)__shfl_up((int)incl, off, 64);
                if ((int)lane >= off) incl += up;
            }
            const unsigned excl = incl - sum;
            if (excl <= rank && rank < incl) {   // exactly one lane
                unsigned below = excl, b = 0, c = c0;
                if (below + c0 <= rank) { below += c0; b = 1; c = c1;
                    if (below + c1 <= rank) { below += c1; b = 2; c = c2;
                        if (below + c2 <= rank) { below += c2; b = 3; c = c3; } } }
                s_bin = 4 * lane + b;
                s_below = below;
                s_in_bin = c;
            }
        }
        __syncthreads();
        prefix |= s_bin << shift;
        mask |= 0xffu << shift;
        rank -= s_below;
        in_bin = s_in_bin;
    }
    // prefix = the k-th key; `rank` = its index among the in_bin keys equal to it; k - rank keys are smaller
    unsigned succ = prefix;   // the same key again (duplicates), or no successor at all (k is the maximum)
    if (rank + 1 >= in_bin && (k - rank) + in_bin < n) {   // wave-uniform: the successor is the smallest key above
        unsigned m = 0xffffffffu;
#pragma unroll
        for (unsigned j = 0; j < PER; ++j)
            if (j * 1024 + threadIdx.x < n && key[j] > prefix) m = min(m, key[j]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = min(m, (unsigned)__shfl_down((int)m, off, 64));
        if (lane == 0) s_min[threadIdx.x >> 6] = m;
        __syncthreads();
        succ = s_min[0];
        for (int w = 1; w < 16; ++w) succ = min(succ, s_min[w]);
    }
    if (threadIdx.x == 0) {
        g_last_path = 2;
        out2[0] = from_key(prefix);
        out2[1] = from_key(succ);
    }
}

template <typename I>
int run_select(const float *in, size_t n, size_t k, float *dev_out2) {
    hipStream_t s = np::stream();
    np::Scratch buf;
    if (int rc = buf.alloc(256 + BINS * HCOPIES * sizeof(unsigned long long))) return rc;
    SelectState *st = (SelectState *)buf.ptr;
    unsigned long long *hist = (unsigned long long *)((char *)buf.ptr + 256);
    size_t blocks = (n / 4 + 255) / 256;
    // Workgroups (of four waves) of the filter pass = a quarter of the copied keys' segments.  3 per CU less one: measured
    // on 10^8 floats against 255 ... 2047 (tools/select_ab.py): 1023 / 1024 / 1025 are 8 % slower (1024 is also a
    // power-of-two stride between a lane's loads), 511 as slow as 1024, 2047 25 % slower.
    const size_t cap = g_filter_blocks ? g_filter_blocks : (size_t)np::num_cus() * 3 - 1;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    select_init_kernel<<<1, 256, 0, s>>>(st, hist, (unsigned long long)k);
    NP_LAUNCH_CHECK("select_init_kernel");
    // bracket path: sample -> bracket -> one filtering pass -> verdict; every decision stays on the device
    np::Scratch copy;
    float *cbuf = nullptr;
    unsigned *ccount = nullptr, *cbelow = nullptr;
    unsigned seg_cap = 0;
    if (g_bracket_min_n && n >= g_bracket_min_n) {
        // a wave sees at most ceil(nvec / (64 * waves)) + 1 vectors; its segment holds 3/8 of that plus slack (the
        // bracket is at most 1/4 of the SAMPLE; a fuller segment raises `overflow` and the path is dropped)
        const size_t waves = blocks * 4;
        const size_t share = ((n / 4 + waves * 64 - 1) / (waves * 64) + 1) * 4 * 64;
        seg_cap = (unsigned)((share * 3 / 8 + 64 + 3) & ~size_t(3));
        if (copy.alloc(2 * waves * sizeof(unsigned) + 256 + waves * (size_t)seg_cap * sizeof(float)) == NP_OK) {
            ccount = (unsigned *)copy.ptr;
            cbelow = ccount + waves;
            cbuf = (float *)((char *)copy.ptr + ((2 * waves * sizeof(unsigned) + 255) & ~size_t(255)));
            select_sample_kernel<0, I><<<SAMPLE_WGS, 256, 0, s>>>(in, (I)n, st, hist);
            select_sample_kernel<1, I><<<SAMPLE_WGS, 256, 0, s>>>(in, (I)n, st, hist);
            select_filter_kernel<I><<<(unsigned)blocks, 256, 0, s>>>(in, (I)n, st, cbuf, ccount, cbelow, seg_cap);
            NP_LAUNCH_CHECK("select bracket kernels");
        }   // else: no room for the copy — the plain three passes need none
    }
    const unsigned nseg = (unsigned)blocks * 4;                         // the filter pass's waves
    // the radix passes: one 16-wave workgroup per CU when they read the array, and at least a wave per segment
    size_t hblocks = (n / 4 + HT - 1) / HT;
    if (hblocks > (size_t)np::num_cus()) hblocks = (size_t)np::num_cus();
    if (hblocks < (nseg + HT / 64 - 1) / (HT / 64)) hblocks = (nseg + HT / 64 - 1) / (HT / 64);
    select_hist_kernel<0, I><<<(unsigned)hblocks, HT, 0, s>>>(in, (I)n, st, hist, cbuf, ccount, seg_cap, nseg, dev_out2);
    select_hist_kernel<1, I><<<(unsigned)hblocks, HT, 0, s>>>(in, (I)n, st, hist, cbuf, ccount, seg_cap, nseg, dev_out2);
    select_hist_kernel<2, I><<<(unsigned)hblocks, HT, 0, s>>>(in, (I)n, st, hist, cbuf, ccount, seg_cap, nseg, dev_out2);
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
    if (n <= g_small_max) {
        select_small_kernel<<<1, 1024, 0, np::stream()>>>(in, (unsigned)n, (unsigned)k, dev_out2);
        NP_LAUNCH_CHECK("select_small_kernel");
        return NP_OK;
    }
    if (n < (size_t(1) << 31)) return run_select<uint32_t>(in, n, k, dev_out2);
    return run_select<uint64_t>(in, n, k, dev_out2);
}

int np_select_set_variant(int variant) {
    if (variant < 0) return np::fail(NP_ERR_INVALID, "np_select_set_variant: 0 = no bracket path, 1 = default, 2..2047 = workgroups, else the smallest n that takes the path");
    if (variant >= 2 && variant < 2048) {   // tuning: grid of the streaming passes
        g_filter_blocks = (size_t)variant;
        return NP_OK;
    }
    if (variant == 1) g_filter_blocks = 0;
    g_small_max = variant == 0 ? 0 : SMALL_MAX;
    g_bracket_min_n = variant == 0 ? 0 : variant == 1 ? size_t(1) << 26 : (size_t)variant;
    if (g_bracket_min_n && g_bracket_min_n < 2048) g_bracket_min_n = 2048;   // a sample run is 1024 floats
    return NP_OK;
}

int np_select_last_path(int *path) {
    if (!path) return np::fail(NP_ERR_INVALID, "np_select_last_path: null output");
    if (int rc = np::ensure_init()) return rc;
    NP_HIP_CHECK(hipStreamSynchronize(np::stream()));
    NP_HIP_CHECK(hipMemcpyFromSymbol(path, HIP_SYMBOL(g_last_path), sizeof(int)));
    return NP_OK;
}

int np_order_stat(const float *in, size_t n, size_t k, float *host_out2) {
    if (!host_out2) return np::fail(NP_ERR_INVALID, "np_order_stat: null output");
    if (int rc = np::ensure_init()) return rc;
    np::ResultCall call(2);
    float *slot = call.slot;
    if (!slot) return NP_ERR_ALLOC;
    if (int rc = np_order_stat_dev(in, n, k, slot)) return rc;
    if (int rc = call.wait()) return rc;
    host_out2[0] = slot[0];
    host_out2[1] = slot[1];
    return NP_OK;
}

}  // extern "C"
