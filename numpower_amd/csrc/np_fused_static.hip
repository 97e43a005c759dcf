// Compiled chains — straight-line kernels for the short, cheap elementwise chains.
//
// The chain interpreter of np_elementwise.hip takes ANY chain of up to 12 steps: a loop over step descriptors, a uniform
// switch per step.  rocprofv3 (profiles/r03/pmc_sq_final.txt) showed what that costs where the chain itself is cheap:
// sum(exp(X), axis 0) issued 19.5 VALU instructions per element for ~11 of arithmetic, 85 % VALU-active, 0.61 of the HBM
// roofline — the switch's joins copy the chain value between register sets, every trip re-derives its live masks, zero-fills
// the slots it is about to load, and folds into one accumulator.  For chains of 1-3 steps taken from a small menu the
// sequence is instead a TEMPLATE PARAMETER here: no descriptor loads, no switch, every load of a trip issued up front, full
// trips without masks (the ragged end is peeled), four independent accumulators per lane in front of a reduction.
//
// What is static: the step kind (unary / binary with an array operand / binary with a scalar operand), the op, the operand
// order (swap).  What stays a run-time uniform: which array an operand is and how it is indexed (full / row / column
// broadcast / 0-d) — that only steers address arithmetic in front of a load.  The op bodies are the SAME unary_apply /
// binary_apply as everywhere else (np_elementwise_ops.h), so a compiled chain that stores its value is bit-identical to the
// interpreter and to the op-by-op sequence (tests/test_gpu_fusion.py); a chain that ends in a reduction folds in a
// different order than the interpreter (more accumulators) and is held to the same 1e-5 bar.
//
// Chains that are not on the menu (np_fused_static_covers), that carry the AVX-body quirk flag on an op other than multiply (or
// on a multiply under a first-axis sum), whose broadcast rows are not float4-divisible, or that are larger than 2^31 elements go
// to the interpreter as before.  Multiply's quirk — what every chain of the PHP binding carries — is known here (CArgs::quirk).
// np_elementwise_set_variant(7000) sends everything there (A/B, tools/fused_static_ab.py).
#include <type_traits>

#include "np_elementwise_ops.h"

#pragma clang fp contract(off)

namespace {

constexpr int CK_UNARY = 0, CK_ARRAY = 1, CK_SCALAR = 2;
constexpr int cstep(int ck, int op, int swap = 0) { return op | (ck << 8) | (swap << 10); }
constexpr int cs_op(int s) { return s & 255; }
constexpr int cs_kind(int s) { return (s >> 8) & 3; }
constexpr bool cs_swap(int s) { return ((s >> 10) & 1) != 0; }
constexpr int kNoStep = -1;

template <int S0, int S1 = kNoStep, int S2 = kNoStep>
struct CChain {
    static constexpr int n = S2 != kNoStep ? 3 : S1 != kNoStep ? 2 : 1;
    static constexpr int at(int k) { return k == 0 ? S0 : k == 1 ? S1 : S2; }
    static constexpr bool has_array() {
        return cs_kind(S0) == CK_ARRAY || (S1 != kNoStep && cs_kind(S1) == CK_ARRAY) || (S2 != kNoStep && cs_kind(S2) == CK_ARRAY);
    }
};

// operand indexing (np::FusedStaticDesc::idx): 0 full, 1 row (e % cols), 2 column (e / cols), 3 zero-d
struct CArgs {
    const float *in0;
    const float *operand[3];
    int idx[3];
    float scalar[3], p0[3], p1[3];
    // a multiply step that carries NP_QUIRK_AVX_BODY (what the PHP binding's chains do: the reference's AVX2 body writes every zero
    // product as -0.0f, its scalar tail as +0.0f, arithmetics.c:403,410-412): quirk[k] != 0, body = flat index < body_end[k].
    // Flat kernels only (the axis kernels are not given such chains: fused_static_covers).
    int quirk[3];
    unsigned body_end[3];
    unsigned cols, div_m, div_s1, div_s2;   // row length of the result when an operand is broadcast, else 0
    unsigned *ticket;                       // full reduction on a small grid: the last workgroup folds (np_internal.h)
    float *result;
};

template <int SINK>
__device__ __forceinline__ float c_identity() {
    return SINK == NP_SUM ? 0.0f : SINK == NP_PROD ? 1.0f : SINK == NP_MIN ? INFINITY : -INFINITY;
}
template <int SINK>
__device__ __forceinline__ float c_combine(float a, float b) {
    if constexpr (SINK < 0) return a;
    else return np::dev::r_combine<SINK>(a, b);
}

__device__ __forceinline__ v4f c_fetch(const float *p, int idx, unsigned first, unsigned row, unsigned col) {
    if (idx == 0) return __builtin_nontemporal_load((const v4f_u *)(p + (size_t)first));
    if (idx == 1) return *(const v4f_u *)(p + (size_t)col);   // one row, re-read by every result row: cache-resident
    const float t = p[idx == 2 ? (size_t)row : (size_t)0];
    return v4f{t, t, t, t};
}

// step K of the chain on the N values a lane holds
// `firsts`: the flat index of the first element of each float4 slot (N == 1: of the element) — only read by a multiply step with
// the AVX-body quirk; the axis kernels pass nullptr
template <class CH, int K, int N>
__device__ __forceinline__ void c_step(const CArgs &a, float (&acc)[N], const float (&oth)[3][N], const unsigned *firsts = nullptr) {
    if constexpr (K < CH::n) {
        constexpr int s = CH::at(K), OP = cs_op(s);
        if constexpr (OP == NP_MULTIPLY && cs_kind(s) != CK_UNARY) {
            if (a.quirk[K]) {   // uniform
                const unsigned be = a.body_end[K];
                const float c = a.scalar[K];
#pragma unroll
                for (int e = 0; e < N; ++e) {
                    const bool body = (N == 1 ? firsts[0] : firsts[e / 4] + (unsigned)(e & 3)) < be;
                    const float o = cs_kind(s) == CK_SCALAR ? c : oth[K][e];
                    acc[e] = cs_swap(s) ? binary_apply<NP_MULTIPLY, true>(o, acc[e], body) : binary_apply<NP_MULTIPLY, true>(acc[e], o, body);
                }
                return;
            }
        }
        if constexpr (cs_kind(s) == CK_UNARY) {
            const float p0 = a.p0[K], p1 = a.p1[K];
#pragma unroll
            for (int e = 0; e < N; ++e) acc[e] = unary_apply<OP>(acc[e], p0, p1);
        } else if constexpr (cs_kind(s) == CK_SCALAR) {
            const float c = a.scalar[K];
#pragma unroll
            for (int e = 0; e < N; ++e)
                acc[e] = cs_swap(s) ? binary_apply<OP, false>(c, acc[e], false) : binary_apply<OP, false>(acc[e], c, false);
        } else {
#pragma unroll
            for (int e = 0; e < N; ++e)
                acc[e] = cs_swap(s) ? binary_apply<OP, false>(oth[K][e], acc[e], false) : binary_apply<OP, false>(acc[e], oth[K][e], false);
        }
    }
}

// The chain value of U float4 slots: first[u] = flat index of a slot's first element, (row[u], col[u]) its position in the
// rows x cols result (only read when an operand is broadcast).  Every load first, then the steps.
template <class CH, int U>
__device__ __forceinline__ void c_trip_at(const CArgs &a, const unsigned (&first)[U], const unsigned (&row)[U], const unsigned (&col)[U],
                                          float (&acc)[U * 4]) {
    v4f x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load((const v4f_u *)(a.in0 + (size_t)first[u]));
    float oth[3][U * 4];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (k < CH::n && cs_kind(CH::at(k)) == CK_ARRAY) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const v4f t = c_fetch(a.operand[k], a.idx[k], first[u], row[u], col[u]);
#pragma unroll
                for (int e = 0; e < 4; ++e) oth[k][u * 4 + e] = t[e];
            }
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[u * 4 + e] = x[u][e];
    c_step<CH, 0, U * 4>(a, acc, oth, first);
    c_step<CH, 1, U * 4>(a, acc, oth, first);
    c_step<CH, 2, U * 4>(a, acc, oth, first);
}

// ... with (row, col) derived from the flat index (the flat kernels)
template <class CH, int U>
__device__ __forceinline__ void c_trip(const CArgs &a, const unsigned (&first)[U], float (&acc)[U * 4]) {
    unsigned row[U], col[U];
#pragma unroll
    for (int u = 0; u < U; ++u) row[u] = col[u] = 0;
    if constexpr (CH::has_array()) {
        if (a.cols) {   // uniform: only a broadcast operand needs (row, col)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                row[u] = fast_div(first[u], a.div_m, a.div_s1, a.div_s2);
                col[u] = first[u] - row[u] * a.cols;
            }
        }
    }
    c_trip_at<CH, U>(a, first, row, col, acc);
}

// one element (the < 4 elements behind the last float4 slot)
template <class CH>
__device__ __forceinline__ float c_element(const CArgs &a, unsigned e) {
    float acc[1] = {a.in0[e]};
    float oth[3][1] = {{0.0f}, {0.0f}, {0.0f}};
    const unsigned row = a.cols ? e / a.cols : 0, col = a.cols ? e - row * a.cols : 0;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (k < CH::n && cs_kind(CH::at(k)) == CK_ARRAY)
            oth[k][0] = a.operand[k][a.idx[k] == 0 ? (size_t)e : a.idx[k] == 1 ? (size_t)col : a.idx[k] == 2 ? (size_t)row : (size_t)0];
    c_step<CH, 0, 1>(a, acc, oth, &e);
    c_step<CH, 1, 1>(a, acc, oth, &e);
    c_step<CH, 2, 1>(a, acc, oth, &e);
    return acc[0];
}

// SINK < 0: out[e] = chain(e).  SINK = NP_SUM ...: one partial per workgroup in out[], folded by the last workgroup
// (a.ticket) or by np::fold_partials behind the launch — the protocol of the interpreter's kernel.
template <class CH, int SINK>
__global__ __launch_bounds__(256) void cchain_flat_kernel(CArgs a, float *__restrict__ out, unsigned n) {
    const unsigned nvec = n / 4, stride = gridDim.x * 256u, tid = blockIdx.x * 256u + threadIdx.x;
    float r4[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) r4[e] = c_identity<SINK>();
    unsigned v = tid;
    // full trips: two slots a grid apart, no masks (nvec < 2^29: v + stride cannot wrap)
    for (; v + stride < nvec; v += 2 * stride) {
        const unsigned first[2] = {v * 4, (v + stride) * 4};
        float acc[8];
        c_trip<CH, 2>(a, first, acc);
        if constexpr (SINK < 0) {
            __builtin_nontemporal_store(v4f{acc[0], acc[1], acc[2], acc[3]}, (v4f_u *)(out + (size_t)first[0]));
            __builtin_nontemporal_store(v4f{acc[4], acc[5], acc[6], acc[7]}, (v4f_u *)(out + (size_t)first[1]));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) r4[e] = c_combine<SINK>(c_combine<SINK>(r4[e], acc[e]), acc[4 + e]);
        }
    }
    if (v < nvec) {   // the one slot left to this lane
        const unsigned first[1] = {v * 4};
        float acc[4];
        c_trip<CH, 1>(a, first, acc);
        if constexpr (SINK < 0) {
            __builtin_nontemporal_store(v4f{acc[0], acc[1], acc[2], acc[3]}, (v4f_u *)(out + (size_t)first[0]));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) r4[e] = c_combine<SINK>(r4[e], acc[e]);
        }
    }
    if (tid == 0) {
        for (unsigned e = nvec * 4; e < n; ++e) {
            const float t = c_element<CH>(a, e);
            if constexpr (SINK < 0) out[e] = t;
            else r4[0] = c_combine<SINK>(r4[0], t);
        }
    }
    if constexpr (SINK >= 0) {
        __shared__ float lds4[4];
        const float r = c_combine<SINK>(c_combine<SINK>(r4[0], r4[1]), c_combine<SINK>(r4[2], r4[3]));
        np::dev::fold_in_last_workgroup<SINK>(np::dev::block_reduce<SINK>(r, lds4), out, a.ticket, a.result, 1.0f, lds4);
    }
}

// Chain ending in a reduction over the FIRST axis: out[chunk][c] = reduce over the chunk's rows of chain(r, c).  The
// interpreter's geometry (fused_chain_cols_kernel): a workgroup owns 64 float4 column slots, its four waves take every
// fourth row of the block's row chunk, combined through LDS; several chunks are folded by np_reduce_axis afterwards.
// Here a lane keeps CU rows in flight with an accumulator set per row slot, a ROW-broadcast operand is loaded ONCE per lane
// (it is the same four columns for every row), and the ragged end of the chunk is peeled instead of masked.
template <class CH, int SINK, int CU>
__global__ __launch_bounds__(256) void cchain_cols_kernel(CArgs a, float *__restrict__ out, unsigned rows, unsigned cols,
                                                          unsigned rows_per_chunk, float mean_div) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned slots_per_row = cols / 4;
    const unsigned r0 = blockIdx.y * rows_per_chunk;
    const unsigned r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
    const unsigned slot = blockIdx.x * 64 + lane;
    float rv[CU][4];
#pragma unroll
    for (int u = 0; u < CU; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) rv[u][e] = c_identity<SINK>();
    if (slot < slots_per_row) {
        const unsigned c0 = slot * 4;
        float rowop[3][4];   // ROW-broadcast operands: this lane's four columns, loaded once
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k < CH::n && cs_kind(CH::at(k)) == CK_ARRAY) {
                v4f t = v4f{0.0f, 0.0f, 0.0f, 0.0f};
                if (a.idx[k] == 1) t = *(const v4f_u *)(a.operand[k] + c0);
                else if (a.idx[k] == 3) t = v4f{a.operand[k][0], a.operand[k][0], a.operand[k][0], a.operand[k][0]};
#pragma unroll
                for (int e = 0; e < 4; ++e) rowop[k][e] = t[e];
            }
        const auto rows_at = [&](auto count, unsigned r) {   // `count` rows r, r + 4, ... of this lane's slot
            constexpr int U = decltype(count)::value;
            v4f x[U];
#pragma unroll
            for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load((const v4f_u *)(a.in0 + (size_t)(r + 4 * u) * cols + c0));
            float oth[3][U * 4];
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (k < CH::n && cs_kind(CH::at(k)) == CK_ARRAY) {
                    if (a.idx[k] == 0) {   // uniform
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const v4f t = __builtin_nontemporal_load((const v4f_u *)(a.operand[k] + (size_t)(r + 4 * u) * cols + c0));
#pragma unroll
                            for (int e = 0; e < 4; ++e) oth[k][u * 4 + e] = t[e];
                        }
                    } else if (a.idx[k] == 2) {   // one value per row (the same address in every lane: one transaction)
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const float t = a.operand[k][r + 4 * u];
#pragma unroll
                            for (int e = 0; e < 4; ++e) oth[k][u * 4 + e] = t;
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < U; ++u)
#pragma unroll
                            for (int e = 0; e < 4; ++e) oth[k][u * 4 + e] = rowop[k][e];
                    }
                }
            float acc[U * 4];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[u * 4 + e] = x[u][e];
            c_step<CH, 0, U * 4>(a, acc, oth);
            c_step<CH, 1, U * 4>(a, acc, oth);
            c_step<CH, 2, U * 4>(a, acc, oth);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) rv[u][e] = c_combine<SINK>(rv[u][e], acc[u * 4 + e]);
        };
        unsigned r = r0 + wave;
        for (; r + 4 * (CU - 1) < r1; r += 4 * CU) rows_at(std::integral_constant<int, CU>{}, r);
        for (; r < r1; r += 4) rows_at(std::integral_constant<int, 1>{}, r);
    }
    // cross-wave combine through LDS, lane fastest (a wave's 64 accesses of one element land on 64 different banks)
    __shared__ float part[4][4][64];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float v = rv[0][e];
#pragma unroll
        for (int u = 1; u < CU; ++u) v = c_combine<SINK>(v, rv[u][e]);
        part[wave][e][lane] = v;
    }
    __syncthreads();
    if (wave == 0 && slot < slots_per_row) {
        v4f v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = part[0][e][lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) t = c_combine<SINK>(t, part[w][e][lane]);
            v[e] = mean_div != 0.0f ? t / mean_div : t;
        }
        *(v4f_u *)(out + (size_t)blockIdx.y * cols + (size_t)slot * 4) = v;
    }
}

// out[r][c] = chain(r, c) with broadcast operands, on whole float4 columns (cols % 4 == 0, at least one workgroup of them): a
// lane owns ONE float4 column of a block of RPT rows — the geometry of binary_rows2d_kernel (np_elementwise.hip).  A ROW
// operand's four values are loaded once, a COL operand is one scalar per row (the same address in every lane), no division
// to find (row, col), RPT independent loads of every full operand in flight.  The flat kernel pays a fast_div and a second
// (cached) load per float4 for the same values: exp(X) + col 6.1 -> 6.4 TB/s, x * x + col + row +12-14 %, exp(X) + row +-0
// (profiles/r04/bcast2d_ab.log, in alternation).  Two rows per lane: fatter lanes stream slower on this machine (four: -3 %).
template <class CH, int RPT>
__global__ __launch_bounds__(256) void cchain_tile2d_kernel(CArgs a, float *__restrict__ out, unsigned rows, unsigned cols, unsigned items) {
    // work item = (row block, float4 column), column fastest; a.div_* divide by cols / 4 here
    const unsigned id = blockIdx.x * 256u + threadIdx.x;
    if (id >= items) return;
    const unsigned rb = fast_div(id, a.div_m, a.div_s1, a.div_s2), c0 = (id - rb * (cols / 4)) * 4, r0 = rb * RPT;
    float rowop[3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k)
        if (k < CH::n && cs_kind(CH::at(k)) == CK_ARRAY) {
            v4f t = v4f{0.0f, 0.0f, 0.0f, 0.0f};
            if (a.idx[k] == 1) t = *(const v4f_u *)(a.operand[k] + c0);
            else if (a.idx[k] == 3) t = v4f{a.operand[k][0], a.operand[k][0], a.operand[k][0], a.operand[k][0]};
#pragma unroll
            for (int e = 0; e < 4; ++e) rowop[k][e] = t[e];
        }
    const auto rows_at = [&](auto count, unsigned r) {   // `count` consecutive rows from r
        constexpr int U = decltype(count)::value;
        v4f x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = __builtin_nontemporal_load((const v4f_u *)(a.in0 + (size_t)(r + u) * cols + c0));
        float oth[3][U * 4];
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (k < CH::n && cs_kind(CH::at(k)) == CK_ARRAY) {
                if (a.idx[k] == 0) {   // uniform
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const v4f t = __builtin_nontemporal_load((const v4f_u *)(a.operand[k] + (size_t)(r + u) * cols + c0));
#pragma unroll
                        for (int e = 0; e < 4; ++e) oth[k][u * 4 + e] = t[e];
                    }
                } else if (a.idx[k] == 2) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const float t = a.operand[k][r + u];
#pragma unroll
                        for (int e = 0; e < 4; ++e) oth[k][u * 4 + e] = t;
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int e = 0; e < 4; ++e) oth[k][u * 4 + e] = rowop[k][e];
                }
            }
        float acc[U * 4];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[u * 4 + e] = x[u][e];
        unsigned firsts[U];
#pragma unroll
        for (int u = 0; u < U; ++u) firsts[u] = (r + u) * cols + c0;   // (only a quirk-carrying multiply reads them)
        c_step<CH, 0, U * 4>(a, acc, oth, firsts);
        c_step<CH, 1, U * 4>(a, acc, oth, firsts);
        c_step<CH, 2, U * 4>(a, acc, oth, firsts);
#pragma unroll
        for (int u = 0; u < U; ++u)
            __builtin_nontemporal_store(v4f{acc[u * 4], acc[u * 4 + 1], acc[u * 4 + 2], acc[u * 4 + 3]}, (v4f_u *)(out + (size_t)(r + u) * cols + c0));
    };
    if (r0 + RPT <= rows) rows_at(std::integral_constant<int, RPT>{}, r0);
    else
        for (unsigned r = r0; r < rows; ++r) rows_at(std::integral_constant<int, 1>{}, r);
}

// Chain ending in a reduction over the LAST axis: out[r] = reduce over c of chain(r, c).  The interpreter's wave mode
// (fused_chain_rows_kernel): groups of L lanes (a power of two <= 64) own a row each, so short rows still fill the wave; no
// barrier anywhere.  cols % 4 == 0; the row index is known (no division), two slots per trip without masks, the ragged
// end of the row peeled.
template <class CH, int SINK>
__global__ __launch_bounds__(256) void cchain_rows_kernel(CArgs a, float *__restrict__ out, unsigned rows, unsigned cols, unsigned L,
                                                          float mean_div) {
    const unsigned lane = threadIdx.x & (L - 1), groups = 64 / L;
    const unsigned first_row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * groups + (threadIdx.x & 63) / L;
    const unsigned row_stride = gridDim.x * 4 * groups;
    const unsigned slots = cols / 4;
    for (unsigned r = first_row; r < rows; r += row_stride) {
        float r4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) r4[e] = c_identity<SINK>();
        const unsigned base = r * cols;
        unsigned s = lane;
        for (; s + L < slots; s += 2 * L) {
            const unsigned first[2] = {base + s * 4, base + (s + L) * 4}, row[2] = {r, r}, col[2] = {s * 4, (s + L) * 4};
            float acc[8];
            c_trip_at<CH, 2>(a, first, row, col, acc);
#pragma unroll
            for (int e = 0; e < 4; ++e) r4[e] = c_combine<SINK>(c_combine<SINK>(r4[e], acc[e]), acc[4 + e]);
        }
        if (s < slots) {
            const unsigned first[1] = {base + s * 4}, row[1] = {r}, col[1] = {s * 4};
            float acc[4];
            c_trip_at<CH, 1>(a, first, row, col, acc);
#pragma unroll
            for (int e = 0; e < 4; ++e) r4[e] = c_combine<SINK>(r4[e], acc[e]);
        }
        float v = c_combine<SINK>(c_combine<SINK>(r4[0], r4[1]), c_combine<SINK>(r4[2], r4[3]));
        for (unsigned off = L >> 1; off > 0; off >>= 1) v = c_combine<SINK>(v, __shfl_xor(v, (int)off, 64));
        if (lane == 0) out[r] = mean_div != 0.0f ? v / mean_div : v;
    }
}

// ---- the menu ----
#define U_(OP) cstep(CK_UNARY, OP)
#define A_(OP) cstep(CK_ARRAY, OP)
#define AR_(OP) cstep(CK_ARRAY, OP, 1)    // operand (op) value
#define S_(OP) cstep(CK_SCALAR, OP)
#define SR_(OP) cstep(CK_SCALAR, OP, 1)

// X(chain): every chain gets the flat store kernel (len >= 2 only: a single step IS np_unary / np_binary), the flat sum /
// min / max kernels' sum form, and the column-sum kernel.
#define NP_CCHAINS_1(X)                                                                                                  \
    X(U_(NP_EXP)) X(U_(NP_LOG)) X(U_(NP_SQRT)) X(U_(NP_ABS)) X(U_(NP_NEGATE))                                           \
    X(A_(NP_ADD)) X(A_(NP_SUBTRACT)) X(A_(NP_MULTIPLY)) X(A_(NP_DIVIDE))                                                 \
    X(S_(NP_ADD)) X(S_(NP_SUBTRACT)) X(S_(NP_MULTIPLY)) X(S_(NP_DIVIDE))

#define NP_CCHAINS_2(X)                                                                                                  \
    /* f(x) (op) operand */                                                                                              \
    X(U_(NP_EXP), A_(NP_ADD)) X(U_(NP_EXP), A_(NP_MULTIPLY)) X(U_(NP_EXP), A_(NP_SUBTRACT)) X(U_(NP_EXP), A_(NP_DIVIDE)) \
    X(U_(NP_EXP), S_(NP_ADD)) X(U_(NP_EXP), S_(NP_MULTIPLY)) X(U_(NP_EXP), S_(NP_SUBTRACT))                             \
    X(U_(NP_LOG), A_(NP_ADD)) X(U_(NP_LOG), A_(NP_MULTIPLY)) X(U_(NP_LOG), S_(NP_ADD)) X(U_(NP_LOG), S_(NP_MULTIPLY))   \
    X(U_(NP_SQRT), A_(NP_ADD)) X(U_(NP_SQRT), A_(NP_MULTIPLY)) X(U_(NP_SQRT), S_(NP_ADD)) X(U_(NP_SQRT), S_(NP_MULTIPLY)) \
    X(U_(NP_ABS), A_(NP_ADD)) X(U_(NP_ABS), A_(NP_MULTIPLY)) X(U_(NP_ABS), S_(NP_ADD)) X(U_(NP_ABS), S_(NP_MULTIPLY))   \
    /* f(x (op) operand) */                                                                                              \
    X(A_(NP_ADD), U_(NP_EXP)) X(A_(NP_SUBTRACT), U_(NP_EXP)) X(A_(NP_MULTIPLY), U_(NP_EXP)) X(S_(NP_ADD), U_(NP_EXP))   \
    X(S_(NP_SUBTRACT), U_(NP_EXP)) X(S_(NP_MULTIPLY), U_(NP_EXP))                                                        \
    X(A_(NP_ADD), U_(NP_LOG)) X(A_(NP_MULTIPLY), U_(NP_LOG)) X(A_(NP_DIVIDE), U_(NP_LOG)) X(S_(NP_ADD), U_(NP_LOG))     \
    X(S_(NP_MULTIPLY), U_(NP_LOG))                                                                                       \
    X(A_(NP_ADD), U_(NP_SQRT)) X(A_(NP_MULTIPLY), U_(NP_SQRT)) X(S_(NP_ADD), U_(NP_SQRT)) X(S_(NP_MULTIPLY), U_(NP_SQRT)) \
    X(A_(NP_SUBTRACT), U_(NP_ABS)) X(S_(NP_SUBTRACT), U_(NP_ABS)) X(A_(NP_ADD), U_(NP_ABS))                             \
    /* two arithmetic steps */                                                                                           \
    X(A_(NP_MULTIPLY), A_(NP_ADD)) X(A_(NP_MULTIPLY), S_(NP_ADD)) X(S_(NP_MULTIPLY), A_(NP_ADD)) X(S_(NP_MULTIPLY), S_(NP_ADD)) \
    X(A_(NP_MULTIPLY), A_(NP_SUBTRACT)) X(S_(NP_MULTIPLY), A_(NP_SUBTRACT)) X(A_(NP_MULTIPLY), A_(NP_MULTIPLY))          \
    X(A_(NP_MULTIPLY), S_(NP_MULTIPLY)) X(A_(NP_ADD), A_(NP_MULTIPLY)) X(A_(NP_ADD), S_(NP_MULTIPLY))                   \
    X(A_(NP_SUBTRACT), A_(NP_MULTIPLY)) X(A_(NP_SUBTRACT), S_(NP_MULTIPLY)) X(A_(NP_SUBTRACT), A_(NP_DIVIDE))           \
    X(A_(NP_SUBTRACT), S_(NP_DIVIDE)) X(A_(NP_ADD), A_(NP_ADD)) X(A_(NP_ADD), S_(NP_ADD)) X(A_(NP_ADD), S_(NP_DIVIDE))  \
    X(S_(NP_ADD), A_(NP_MULTIPLY)) X(S_(NP_SUBTRACT), A_(NP_MULTIPLY)) X(S_(NP_SUBTRACT), A_(NP_DIVIDE))

#define NP_CCHAINS_3(X)                                                                                                  \
    X(U_(NP_EXP), A_(NP_MULTIPLY), S_(NP_ADD)) X(U_(NP_EXP), A_(NP_MULTIPLY), A_(NP_ADD))                               \
    X(U_(NP_EXP), S_(NP_MULTIPLY), S_(NP_ADD)) X(U_(NP_EXP), S_(NP_MULTIPLY), A_(NP_ADD))                               \
    X(U_(NP_EXP), A_(NP_ADD), A_(NP_MULTIPLY)) X(U_(NP_EXP), S_(NP_ADD), A_(NP_MULTIPLY))                               \
    X(A_(NP_SUBTRACT), U_(NP_EXP), A_(NP_MULTIPLY)) X(A_(NP_SUBTRACT), U_(NP_EXP), S_(NP_MULTIPLY))                     \
    X(S_(NP_SUBTRACT), U_(NP_EXP), A_(NP_MULTIPLY)) X(A_(NP_SUBTRACT), U_(NP_EXP), A_(NP_DIVIDE))                       \
    X(A_(NP_SUBTRACT), U_(NP_ABS), A_(NP_MULTIPLY)) X(A_(NP_SUBTRACT), U_(NP_ABS), S_(NP_MULTIPLY))                     \
    X(A_(NP_MULTIPLY), A_(NP_ADD), U_(NP_EXP)) X(A_(NP_MULTIPLY), S_(NP_ADD), U_(NP_EXP))                               \
    X(S_(NP_MULTIPLY), A_(NP_ADD), U_(NP_EXP)) X(S_(NP_MULTIPLY), S_(NP_ADD), U_(NP_EXP))                               \
    X(A_(NP_MULTIPLY), A_(NP_ADD), U_(NP_SQRT)) X(A_(NP_MULTIPLY), A_(NP_ADD), U_(NP_ABS))                              \
    X(A_(NP_MULTIPLY), A_(NP_MULTIPLY), A_(NP_ADD)) X(A_(NP_MULTIPLY), S_(NP_MULTIPLY), A_(NP_ADD))                     \
    X(A_(NP_SUBTRACT), A_(NP_MULTIPLY), A_(NP_ADD)) X(A_(NP_SUBTRACT), S_(NP_MULTIPLY), A_(NP_ADD))                     \
    X(A_(NP_SUBTRACT), A_(NP_DIVIDE), U_(NP_EXP)) X(A_(NP_SUBTRACT), S_(NP_DIVIDE), U_(NP_EXP))                         \
    /* distance / affine forms: x * s + col + row (examples/kmeans.py), (x + a) * b + c */                               \
    X(S_(NP_MULTIPLY), A_(NP_ADD), A_(NP_ADD)) X(A_(NP_MULTIPLY), A_(NP_ADD), A_(NP_ADD)) X(A_(NP_ADD), A_(NP_ADD), A_(NP_ADD))    \
    X(A_(NP_ADD), A_(NP_MULTIPLY), A_(NP_ADD)) X(A_(NP_SUBTRACT), A_(NP_MULTIPLY), A_(NP_MULTIPLY))

struct Launchers {
    int key[3];
    void (*tile2d_store)(const CArgs &, float *, unsigned, unsigned, unsigned, hipStream_t);   // chains with an array operand, len >= 2
    void (*flat_store)(const CArgs &, float *, unsigned, unsigned, hipStream_t);
    void (*flat_sum)(const CArgs &, float *, unsigned, unsigned, hipStream_t);
    void (*cols_sum)(const CArgs &, float *, unsigned, unsigned, unsigned, float, dim3, hipStream_t, int);
    void (*rows_sum)(const CArgs &, float *, unsigned, unsigned, unsigned, float, unsigned, hipStream_t);
};

template <class CH, int SINK>
void launch_flat(const CArgs &a, float *out, unsigned n, unsigned grid, hipStream_t s) {
    cchain_flat_kernel<CH, SINK><<<grid, 256, 0, s>>>(a, out, n);
}
template <class CH>
void launch_tile2d(const CArgs &a, float *out, unsigned rows, unsigned cols, unsigned items, hipStream_t s) {
    cchain_tile2d_kernel<CH, 2><<<(items + 255) / 256, 256, 0, s>>>(a, out, rows, cols, items);
}
template <class CH>
constexpr auto tile2d_or_null() -> void (*)(const CArgs &, float *, unsigned, unsigned, unsigned, hipStream_t) {
    if constexpr (CH::has_array()) return launch_tile2d<CH>;
    else return nullptr;
}
template <class CH>
void launch_cols(const CArgs &a, float *out, unsigned rows, unsigned cols, unsigned rows_per_chunk, float mean_div, dim3 grid,
                 hipStream_t s, int rows_in_flight) {
    if (rows_in_flight == 4)
        cchain_cols_kernel<CH, NP_SUM, 4><<<grid, 256, 0, s>>>(a, out, rows, cols, rows_per_chunk, mean_div);
    else
        cchain_cols_kernel<CH, NP_SUM, 2><<<grid, 256, 0, s>>>(a, out, rows, cols, rows_per_chunk, mean_div);
}

template <class CH>
void launch_rows(const CArgs &a, float *out, unsigned rows, unsigned cols, unsigned L, float mean_div, unsigned grid, hipStream_t s) {
    cchain_rows_kernel<CH, NP_SUM><<<grid, 256, 0, s>>>(a, out, rows, cols, L, mean_div);
}

#define NP_ROW1(S0) {{S0, kNoStep, kNoStep}, nullptr, nullptr, launch_flat<CChain<S0>, NP_SUM>, launch_cols<CChain<S0>>, launch_rows<CChain<S0>>},
#define NP_ROW2(S0, S1) {{S0, S1, kNoStep}, tile2d_or_null<CChain<S0, S1>>(), launch_flat<CChain<S0, S1>, -1>, launch_flat<CChain<S0, S1>, NP_SUM>, launch_cols<CChain<S0, S1>>, launch_rows<CChain<S0, S1>>},
#define NP_ROW3(S0, S1, S2) {{S0, S1, S2}, tile2d_or_null<CChain<S0, S1, S2>>(), launch_flat<CChain<S0, S1, S2>, -1>, launch_flat<CChain<S0, S1, S2>, NP_SUM>, launch_cols<CChain<S0, S1, S2>>, launch_rows<CChain<S0, S1, S2>>},
const Launchers kMenu[] = {NP_CCHAINS_1(NP_ROW1) NP_CCHAINS_2(NP_ROW2) NP_CCHAINS_3(NP_ROW3)};
#undef NP_ROW1
#undef NP_ROW2
#undef NP_ROW3

const Launchers *find_chain(const np::FusedStaticDesc &d) {
    if (d.n_ops < 1 || d.n_ops > 3 || !d.in0) return nullptr;
    int key[3] = {kNoStep, kNoStep, kNoStep};
    for (int k = 0; k < d.n_ops; ++k) {
        const int ck = d.kind[k] == NP_FUSED_UNARY ? CK_UNARY : d.operand[k] ? CK_ARRAY : CK_SCALAR;
        if (d.op[k] < 0 || d.op[k] > 255) return nullptr;
        key[k] = cstep(ck, d.op[k], ck == CK_UNARY ? 0 : (d.swap[k] ? 1 : 0));
    }
    for (const Launchers &l : kMenu)
        if (l.key[0] == key[0] && l.key[1] == key[1] && l.key[2] == key[2]) return &l;
    // commutative steps: (operand + value) is (value + operand) bit for bit for finite and infinite values; only which NaN
    // payload survives could differ, and no caller of this path depends on payloads (np_binary's own kernels make the same
    // choice for scalar operands) — still, keep the swapped forms with the interpreter: nothing to gain here
    return nullptr;
}

void fill_args(const np::FusedStaticDesc &d, CArgs &a) {
    a.in0 = d.in0;
    for (int k = 0; k < 3; ++k) {
        a.operand[k] = k < d.n_ops ? d.operand[k] : nullptr;
        a.idx[k] = k < d.n_ops ? d.idx[k] : 0;
        a.scalar[k] = k < d.n_ops ? d.scalar[k] : 0.0f;
        a.p0[k] = k < d.n_ops ? d.p0[k] : 0.0f;
        a.p1[k] = k < d.n_ops ? d.p1[k] : 0.0f;
        a.quirk[k] = k < d.n_ops ? d.quirk[k] : 0;
        a.body_end[k] = k < d.n_ops ? d.body_end[k] : 0u;
    }
    a.cols = d.bcast_cols;
    a.div_m = d.div_m;
    a.div_s1 = d.div_s1;
    a.div_s2 = d.div_s2;
    a.ticket = nullptr;
    a.result = nullptr;
}

}  // namespace

namespace np {

bool fused_static_covers(const FusedStaticDesc &d, int sink, int axis_mode) {
    const Launchers *l = find_chain(d);
    if (!l) return false;
    const bool quirk = d.quirk[0] || d.quirk[1] || d.quirk[2];
    if (axis_mode == 0) return sink == NP_SUM && !quirk;          // the column kernel has no flat index at hand for the AVX-body line
    if (axis_mode == 1) return sink == NP_SUM;
    return sink < 0 ? l->flat_store != nullptr : sink == NP_SUM;
}

int fused_static_flat(const FusedStaticDesc &d, float *out, size_t n, int sink, unsigned grid, unsigned *ticket, float *result) {
    const Launchers *l = find_chain(d);
    if (!l || n >= (size_t(1) << 31) || (sink >= 0 && sink != NP_SUM) || (sink < 0 && !l->flat_store))
        return np::fail(NP_ERR_INVALID, "internal: fused_static_flat called for a chain it does not cover");
    CArgs a;
    fill_args(d, a);
    a.ticket = ticket;
    a.result = result;
    // stored chains with a broadcast operand on whole float4 columns: the 2-D form (cchain_tile2d_kernel)
    if (sink < 0 && l->tile2d_store && d.bcast_cols >= 1024 && d.bcast_cols % 32 == 0 && n % d.bcast_cols == 0 && n / d.bcast_cols >= 2 &&
        !np::g_bcast2d_off) {
        constexpr unsigned RPT = 2;   // (the template argument of launch_tile2d)
        const unsigned rows = (unsigned)(n / d.bcast_cols), row_blocks = (rows + RPT - 1) / RPT;
        fast_div_magic(d.bcast_cols / 4, a.div_m, a.div_s1, a.div_s2);   // (the flat kernel divides by cols, this one by cols / 4)
        l->tile2d_store(a, out, rows, d.bcast_cols, row_blocks * (d.bcast_cols / 4), np::stream());
        NP_LAUNCH_CHECK("cchain_tile2d_kernel");
        return NP_OK;
    }
    (sink < 0 ? l->flat_store : l->flat_sum)(a, out, (unsigned)n, grid, np::stream());
    NP_LAUNCH_CHECK("cchain_flat_kernel");
    return NP_OK;
}

int fused_static_cols(const FusedStaticDesc &d, float *out, size_t rows, size_t cols, size_t rows_per_chunk, float mean_div,
                      unsigned col_blocks, unsigned chunks, int rows_in_flight) {
    const Launchers *l = find_chain(d);
    if (!l || cols % 4 != 0 || rows * cols >= (size_t(1) << 31))
        return np::fail(NP_ERR_INVALID, "internal: fused_static_cols called for a chain it does not cover");
    CArgs a;
    fill_args(d, a);
    l->cols_sum(a, out, (unsigned)rows, (unsigned)cols, (unsigned)rows_per_chunk, mean_div, dim3(col_blocks, chunks), np::stream(),
                rows_in_flight);
    NP_LAUNCH_CHECK("cchain_cols_kernel");
    return NP_OK;
}

int fused_static_rows(const FusedStaticDesc &d, float *out, size_t rows, size_t cols, unsigned L, float mean_div, unsigned grid) {
    const Launchers *l = find_chain(d);
    if (!l || cols % 4 != 0 || rows * cols >= (size_t(1) << 31) || L < 1 || L > 64 || (L & (L - 1)))
        return np::fail(NP_ERR_INVALID, "internal: fused_static_rows called for a chain it does not cover");
    CArgs a;
    fill_args(d, a);
    l->rows_sum(a, out, (unsigned)rows, (unsigned)cols, L, mean_div, grid, np::stream());
    NP_LAUNCH_CHECK("cchain_rows_kernel");
    return NP_OK;
}

}  // namespace np
