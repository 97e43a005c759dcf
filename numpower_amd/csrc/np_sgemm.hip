// fp32 GEMM / GEMV of libnp_hip.so — the kernel behind NDArray::matmul / NDArray::dot.
//
// Replaces cblas_sgemm(RowMajor, NoTrans, NoTrans, M, N, K, 1, A, K, B, N, 0, C, N) and the
// cublasSgemm call of NDArray_FMatmul (src/ndmath/linalg.c:44-82), cblas_sgemv /
// matrixVectorMultiplyFloatKernel of NDArray_Dot (linalg.c:367-386, cuda_math.cu:228,1417).
//
// Design (gfx950): exact-fp32 matrix cores, v_mfma_f32_32x32x2_f32 (64 cycles/instruction/SIMD,
// bit-identical to an fmaf chain, no reduced-precision path).  One workgroup = 4 waves computes
// a BM x BN tile of C; each wave owns a (BM/2) x (BN/2) sub-tile as 32x32 MFMA blocks held in
// accumulator registers.  A and B tiles are staged global -> registers -> LDS with the next
// K-tile's global loads in flight under the current tile's MFMAs, two LDS buffers, one barrier
// per K-tile.  LDS layout:
//   As[buf][m][k]  row-major, rows padded to BK+4 floats: a lane reads 4 consecutive k of its
//                  row with one ds_read_b128 (conflict-free with the +4 pad), which feeds 4
//                  consecutive MFMAs.  The MFMA k index is permuted accordingly: in step s the
//                  two lane halves use k = kb + 4*half + s; A and B use the same permutation,
//                  and a dot product does not care in which order its terms are visited.
//   Bs[buf][k][n]  row-major as in memory; a B fragment is one ds_read_b32 per lane, 32
//                  consecutive floats per half-wave (conflict-free).
// Row-major A.B needs no transposes on either operand with this mapping.
#include <type_traits>

#include <algorithm>

#include "np_internal.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
// float4 that is only dword-aligned: rows of a matrix whose leading dimension is not a multiple
// of 4 (K = 4097, N = 1001 ...).  gfx950 global loads handle it (hipcc emits global_load_dwordx4).
struct __attribute__((packed, aligned(4))) U4 {
    v4f v;
};

struct GemmArgs {
    const float *A, *B;
    float *C;
    unsigned M, N, K;
    unsigned K_last;                        // != 0: inner length of the LAST batch entry (split-K remainder chunk)
    unsigned n_store;                       // != 0: C has only this many columns (B was padded to N, see pad_operands)
    unsigned lda, ldb, ldc;
    size_t stride_a, stride_b, stride_c;   // batch strides (elements)
    unsigned tiles_m, tiles_n;
    unsigned swizzle;                       // 0 = row-major tile order, else XCD-aware grouping
    unsigned prio_period;                   // sgemm_dma_kernel<.., PRIO>: K-tiles between priority flips (see there)
    unsigned k_chunks;                      // sgemm_kq_kernel: != 0: the batch is this many K-chunks of ONE product, launched chunk-major over the XCDs (see there)
    unsigned long long *probe;              // optional per-workgroup timing record (debug), else null
    // Progress reporting (np::sgemm_batched_with_progress): when non-null, every workgroup adds 1 to progress[c] once its
    // C tile is visible device-wide, c = the piece (np_comm_piece split: piece_extra pieces of piece_base + 1 batch
    // entries, then pieces of piece_base) its batch entry belongs to.  A wait kernel on another stream releases the
    // transfer of piece c the moment its last tile has landed — no second launch, no host round trip.
    unsigned *progress;
    unsigned piece_base, piece_extra;
};

// Last statement of sgemm_dma_kernel (all threads of the workgroup call it).  In progress mode the kernel stores C with
// device-scope (memory-side) stores, so a tile is visible to every XCD once its stores are acknowledged: each wave
// waits for its own (s_waitcnt 0), the barrier collects the four waves, one memory-side atomic reports the tile.  No
// device-scope FENCE: that is an L2 write-back per workgroup, and 2048 of them cost the 64 x 1024^3 slab half its rate
// (0.96 -> 1.43 ms, profiles/r03/chunk_overhead.log, first version) — the same lesson as np_internal.h's ticket.
__device__ __forceinline__ void progress_signal(const GemmArgs &g) {
    if (!g.progress) return;   // uniform
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned b = blockIdx.z, big = g.piece_extra * (g.piece_base + 1);
        const unsigned c = b < big ? b / (g.piece_base + 1) : g.piece_extra + (b - big) / g.piece_base;
        __hip_atomic_fetch_add(g.progress + c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Debug instrumentation: per workgroup {shader-clock start, end, 100 MHz wall start, end, XCC id}.
// Used by tools/gemm_probe.py to separate "cycles lost to stalls" from "clock lowered by DVFS".
__device__ __forceinline__ void probe_begin(const GemmArgs &g, unsigned long long &c0,
                                            unsigned long long &w0) {
    if (g.probe) {
        c0 = __builtin_readcyclecounter();
        w0 = wall_clock64();
    }
}
__device__ __forceinline__ void probe_end(const GemmArgs &g, unsigned long long c0,
                                          unsigned long long w0) {
    if (g.probe && threadIdx.x == 0) {
        unsigned long long *p = g.probe + 8 * ((size_t)blockIdx.z * gridDim.x + blockIdx.x);
        p[0] = c0;
        p[1] = __builtin_readcyclecounter();
        p[2] = w0;
        p[3] = wall_clock64();
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        p[4] = xcc;
    }
}

// Tile id -> (tile_m, tile_n).  Workgroup b is placed on XCD b % 8 (observed dispatch order);
// with `swizzle` each XCD gets a contiguous run of tiles, and runs are walked in GROUP-row
// bands so that the workgroups resident on one XCD share A row panels and B column panels in
// that XCD's private L2.  Pure speed choice; correctness never depends on placement.
__device__ __forceinline__ void tile_coords(const GemmArgs &g, unsigned bid, unsigned &tm,
                                            unsigned &tn) {
    const unsigned nwg = g.tiles_m * g.tiles_n;
    unsigned id = bid;
    if (g.swizzle) {
        const unsigned q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // bijective
        const unsigned GROUP = g.swizzle;   // tile rows per band
        const unsigned band = GROUP * g.tiles_n;
        const unsigned gid = id / band;
        const unsigned first_m = gid * GROUP;
        const unsigned rows = (g.tiles_m - first_m < GROUP) ? g.tiles_m - first_m : GROUP;
        tm = first_m + (id % band) % rows;
        tn = (id % band) / rows;
        return;
    }
    tm = id / g.tiles_n;
    tn = id % g.tiles_n;
}

// VEC : K % 4 == 0, N % 4 == 0, lda/ldb multiples of 4 and 16-byte aligned bases -> float4 loads
// EDGE: tile may stick out of the matrix -> bounds-checked loads (zero fill) and stores
template <int BM, int BN, int BK, int MINW, bool VEC, bool EDGE>
__global__ __launch_bounds__(256, MINW) void sgemm_kernel(GemmArgs g) {
    if (g.K_last && blockIdx.z + 1 == gridDim.z) g.K = g.K_last;   // uniform: split-K remainder chunk
    constexpr int LDA_S = BK + 4;               // padded A row (floats)
    constexpr int LDB_S = BN;                   // B row (floats)
    constexpr int WM = BM / 2, WN = BN / 2;     // wave tile
    constexpr int TM = WM / 32, TN = WN / 32;   // 32x32 MFMA blocks per wave
    constexpr int A_V4 = BM * BK / 4 / 256;     // float4 per thread per A tile
    constexpr int B_V4 = BK * BN / 4 / 256;     // float4 per thread per B tile
    static_assert(A_V4 >= 1 && B_V4 >= 1, "tile too small for 256 threads");
    static_assert(BK % 8 == 0, "BK must be a multiple of 8");

    __shared__ __attribute__((aligned(16))) float As[2][BM * LDA_S];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB_S];

    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63, wave = tid >> 6;
    const unsigned li = lane & 31, lh = lane >> 5;
    const unsigned wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;

    unsigned tile_m, tile_n;
    tile_coords(g, blockIdx.x, tile_m, tile_n);
    const unsigned m0 = tile_m * BM, n0 = tile_n * BN;
    unsigned long long probe_c0 = 0, probe_w0 = 0;
    probe_begin(g, probe_c0, probe_w0);

    const float *A = g.A + (size_t)blockIdx.z * g.stride_a;
    const float *B = g.B + (size_t)blockIdx.z * g.stride_b;
    float *C = g.C + (size_t)blockIdx.z * g.stride_c;

    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    v4f ra[A_V4], rb[B_V4];

    auto load_global = [&](unsigned kt) {
        const unsigned k0 = kt * BK;
#pragma unroll
        for (int r = 0; r < A_V4; ++r) {
            const unsigned f = tid + 256 * r;
            const unsigned row = f / (BK / 4), c4 = f % (BK / 4);
            const unsigned gm = m0 + row, gk = k0 + c4 * 4;
            if constexpr (VEC) {
                if (!EDGE || (gm < g.M && gk < g.K))
                    ra[r] = *(const v4f *)(A + (size_t)gm * g.lda + gk);
                else
                    ra[r] = v4f{0, 0, 0, 0};
            } else {
                if (gm < g.M && gk + 3 < g.K) {
                    ra[r] = ((const U4 *)(A + (size_t)gm * g.lda + gk))->v;   // dword-aligned dwordx4
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        ra[r][e] = (gm < g.M && gk + e < g.K) ? A[(size_t)gm * g.lda + gk + e] : 0.0f;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < B_V4; ++r) {
            const unsigned f = tid + 256 * r;
            const unsigned row = f / (BN / 4), c4 = f % (BN / 4);
            const unsigned gk = k0 + row, gn = n0 + c4 * 4;
            if constexpr (VEC) {
                if (!EDGE || (gk < g.K && gn < g.N))
                    rb[r] = *(const v4f *)(B + (size_t)gk * g.ldb + gn);
                else
                    rb[r] = v4f{0, 0, 0, 0};
            } else {
                if (gk < g.K && gn + 3 < g.N) {
                    rb[r] = ((const U4 *)(B + (size_t)gk * g.ldb + gn))->v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        rb[r][e] = (gk < g.K && gn + e < g.N) ? B[(size_t)gk * g.ldb + gn + e] : 0.0f;
                }
            }
        }
    };

    auto store_lds = [&](int buf) {
#pragma unroll
        for (int r = 0; r < A_V4; ++r) {
            const unsigned f = tid + 256 * r;
            const unsigned row = f / (BK / 4), c4 = f % (BK / 4);
            *(v4f *)&As[buf][row * LDA_S + c4 * 4] = ra[r];
        }
#pragma unroll
        for (int r = 0; r < B_V4; ++r) {
            const unsigned f = tid + 256 * r;
            const unsigned row = f / (BN / 4), c4 = f % (BN / 4);
            *(v4f *)&Bs[buf][row * LDB_S + c4 * 4] = rb[r];
        }
    };

    auto compute = [&](int buf) {
        const float *as = &As[buf][(wm0 + li) * LDA_S + 4 * lh];
        const float *bs = &Bs[buf][(4 * lh) * LDB_S + wn0 + li];
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            v4f a4[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) a4[i] = *(const v4f *)(as + i * 32 * LDA_S + kg * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float bv[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[j] = bs[(kg * 8 + s) * LDB_S + j * 32];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i][s], bv[j], acc[i][j],
                                                                        0, 0, 0);
            }
        }
    };

    const unsigned nk = (g.K + BK - 1) / BK;
    load_global(0);
    store_lds(0);
    __syncthreads();
    for (unsigned kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1 < nk);
        if (more) load_global(kt + 1);   // in flight under this tile's MFMAs
        compute(cur);
        if (more) store_lds(cur ^ 1);
        __syncthreads();
    }

    // epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const unsigned col = n0 + wn0 + j * 32 + li;
                if (!EDGE || (row < g.M && col < g.N)) C[(size_t)row * g.ldc + col] = acc[i][j][r];
            }
    probe_end(g, probe_c0, probe_w0);
}

// Software-pipelined variant (the default for large problems).
//
// Why: with the simple kernel above the 4 co-resident waves of a SIMD convoy — each wave's
// non-MFMA tail (vmcnt wait, ds_write, barrier, first ds_read latency ≈ 300 cycles per K-tile)
// lines up with the others', and rocprof shows the matrix pipe busy only 83 % of the kernel
// (profiles/r01/pmc_summary.txt: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM cycles)).
// Here every wave keeps its own MFMA stream fed:
//   * 3 LDS buffers, ONE barrier per K-tile placed in the MIDDLE of the tile: tile t+1 is written
//     to LDS after the first half of tile t's MFMAs have been issued and is published by that
//     barrier; the second half of tile t's MFMAs run behind it.  (Buffer (t+1)%3 was last read
//     for tile t-2, which every wave finished before arriving at the previous barrier.)
//   * A/B fragments are double-buffered in registers: the ds_reads for the next 8-deep k group
//     are issued before the current group's 16 MFMAs, so no MFMA waits on an LDS round trip.
//   * global loads for tile t+2 are issued right after tile t+1's registers were stored.
// BK is fixed at 16 = two k groups = the two halves.
// MODE bit 0: spread the staging work between the MFMAs.  Bits 1-4 are timing ablations used by
// tools/gemm_ab.py only (they produce WRONG results): 2 = no barrier, 4 = no global loads,
// 8 = no LDS stores, 16 = no fragment reads inside the K loop.
template <int BM, int BN, bool VEC, bool EDGE, int MODE>
__global__ __launch_bounds__(256, 2) void sgemm_pipe_kernel(GemmArgs g) {
    if (g.K_last && blockIdx.z + 1 == gridDim.z) g.K = g.K_last;   // uniform: split-K remainder chunk
    constexpr int BK = 16;
    constexpr bool SPREAD = MODE & 1, NO_BAR = MODE & 2, NO_GLD = MODE & 4, NO_STS = MODE & 8, NO_FRAG = MODE & 16;
    constexpr int LDA_S = BK + 4, LDB_S = BN;
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int A_V4 = BM * BK / 4 / 256, B_V4 = BK * BN / 4 / 256;
    constexpr int A_SZ = BM * LDA_S, B_SZ = BK * LDB_S;

    __shared__ __attribute__((aligned(16))) float smem[3 * (A_SZ + B_SZ)];
    float *const As = smem;
    float *const Bs = smem + 3 * A_SZ;

    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63, wave = tid >> 6;
    const unsigned li = lane & 31, lh = lane >> 5;
    const unsigned wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;

    unsigned tile_m, tile_n;
    tile_coords(g, blockIdx.x, tile_m, tile_n);
    const unsigned m0 = tile_m * BM, n0 = tile_n * BN;
    unsigned long long probe_c0 = 0, probe_w0 = 0;
    probe_begin(g, probe_c0, probe_w0);

    const float *A = g.A + (size_t)blockIdx.z * g.stride_a;
    const float *B = g.B + (size_t)blockIdx.z * g.stride_b;
    float *C = g.C + (size_t)blockIdx.z * g.stride_c;

    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    v4f ra[A_V4], rb[B_V4];

    auto load_global = [&](unsigned kt) {
        const unsigned k0 = kt * BK;
#pragma unroll
        for (int r = 0; r < A_V4; ++r) {
            const unsigned f = tid + 256 * r;
            const unsigned row = f / (BK / 4), c4 = f % (BK / 4);
            const unsigned gm = m0 + row, gk = k0 + c4 * 4;
            if constexpr (VEC) {
                if (!EDGE || (gm < g.M && gk < g.K))
                    ra[r] = *(const v4f *)(A + (size_t)gm * g.lda + gk);
                else
                    ra[r] = v4f{0, 0, 0, 0};
            } else {
                if (gm < g.M && gk + 3 < g.K) {
                    ra[r] = ((const U4 *)(A + (size_t)gm * g.lda + gk))->v;   // dword-aligned dwordx4
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        ra[r][e] = (gm < g.M && gk + e < g.K) ? A[(size_t)gm * g.lda + gk + e] : 0.0f;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < B_V4; ++r) {
            const unsigned f = tid + 256 * r;
            const unsigned row = f / (BN / 4), c4 = f % (BN / 4);
            const unsigned gk = k0 + row, gn = n0 + c4 * 4;
            if constexpr (VEC) {
                if (!EDGE || (gk < g.K && gn < g.N))
                    rb[r] = *(const v4f *)(B + (size_t)gk * g.ldb + gn);
                else
                    rb[r] = v4f{0, 0, 0, 0};
            } else {
                if (gk < g.K && gn + 3 < g.N) {
                    rb[r] = ((const U4 *)(B + (size_t)gk * g.ldb + gn))->v;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        rb[r][e] = (gk < g.K && gn + e < g.N) ? B[(size_t)gk * g.ldb + gn + e] : 0.0f;
                }
            }
        }
    };

    auto store_lds = [&](unsigned buf) {
        float *as = As + buf * A_SZ;
        float *bs = Bs + buf * B_SZ;
#pragma unroll
        for (int r = 0; r < A_V4; ++r) {
            const unsigned f = tid + 256 * r;
            const unsigned row = f / (BK / 4), c4 = f % (BK / 4);
            *(v4f *)&as[row * LDA_S + c4 * 4] = ra[r];
        }
#pragma unroll
        for (int r = 0; r < B_V4; ++r) {
            const unsigned f = tid + 256 * r;
            const unsigned row = f / (BN / 4), c4 = f % (BN / 4);
            *(v4f *)&bs[row * LDB_S + c4 * 4] = rb[r];
        }
    };

    // register fragments of one 8-deep k group: a4[i] = 4 consecutive k of this lane's row,
    // bv[s][j] = B[kb + 4*half + s][col]
    struct Frag {
        v4f a4[TM];
        float bv[4][TN];
    };
    auto read_frag = [&](Frag &f, unsigned buf, int kg) {
        const float *as = As + buf * A_SZ + (wm0 + li) * LDA_S + 4 * lh + kg * 8;
        const float *bs = Bs + buf * B_SZ + (4 * lh + kg * 8) * LDB_S + wn0 + li;
#pragma unroll
        for (int i = 0; i < TM; ++i) f.a4[i] = *(const v4f *)(as + i * 32 * LDA_S);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < TN; ++j) f.bv[s][j] = bs[s * LDB_S + j * 32];
    };
    auto mfma_steps = [&](const Frag &f, int s0, int s1) {
#pragma unroll
        for (int s = s0; s < s1; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a4[i][s], f.bv[s][j], acc[i][j], 0, 0, 0);
    };

    const unsigned nk = (g.K + BK - 1) / BK;
    Frag f0, f1;
    load_global(0);
    store_lds(0);
    if (nk > 1) load_global(1);
    __syncthreads();
    read_frag(f0, 0, 0);

    unsigned cur = 0;
    for (unsigned kt = 0; kt < nk; ++kt) {
        const unsigned nxt = (cur == 2) ? 0 : cur + 1;
        const bool more = (kt + 1 < nk);
        // first half: k group 0 (fragments already in f0); fetch group 1 underneath
        if (!NO_FRAG) read_frag(f1, cur, 1);
        if constexpr (SPREAD) {
            // staging work spread between the MFMAs so that it issues in their shadow
            mfma_steps(f0, 0, 2);
            if (more && !NO_STS) store_lds(nxt);  // tile kt+1 -> LDS (its loads were issued a tile ago)
            mfma_steps(f0, 2, 3);
            if (kt + 2 < nk && !NO_GLD) load_global(kt + 2);   // in flight during the next ~32 MFMAs
            mfma_steps(f0, 3, 4);
        } else {
            mfma_steps(f0, 0, 4);
            if (more) {
                if (!NO_STS) store_lds(nxt);
                if (kt + 2 < nk && !NO_GLD) load_global(kt + 2);
            }
        }
        if (!NO_BAR) __syncthreads();             // publishes tile kt+1
        // second half: k group 1; fetch group 0 of the next tile underneath
        if (more && !NO_FRAG) read_frag(f0, nxt, 0);
        mfma_steps(NO_FRAG ? f0 : f1, 0, 4);
        cur = nxt;
    }

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const unsigned col = n0 + wn0 + j * 32 + li;
                if (!EDGE || (row < g.M && col < g.N)) C[(size_t)row * g.ldc + col] = acc[i][j][r];
            }
    probe_end(g, probe_c0, probe_w0);
}

// LDS-DMA variant for fully aligned problems (M % 256 == 0, N % 128 == 0, K % 16 == 0): the
// large-matrix default.
//
// Why (ablation of the kernel above, tools/gemm_ab.py, profiles/r01/gemm_ablation.log): with
// everything but the MFMAs removed the loop runs 152 TFLOP/s; the register->LDS stores of the
// staged tiles alone cost 7 %, the fragment reads 4 %, global loads and the barrier ~0.  So:
//   * global -> LDS goes by global_load_lds_dwordx4 (no staging VGPRs, no ds_write at all).  The
//     DMA writes lane-linear 16-byte slots, so the A tile is [256 rows][4 slots of 4 k] unpadded
//     and the bank-conflict fix moves to the SOURCE address: LDS slot p of row r holds k-chunk
//     p ^ ((r >> 2) & 3); fragment reads apply the same XOR (conflict-free ds_read_b128: a
//     16-lane group covers 16 distinct 16-byte slots of the 256-byte bank row).
//   * the wave tile grows to 128 x 64 (4 x 2 MFMA blocks, 128 accumulator VGPRs): every B
//     fragment register now feeds 4 MFMAs, halving the ds_read_b32 traffic per MFMA.
//   * 256 x 128 workgroup tile, 4 waves, 3 LDS buffers (72 KiB) -> 2 workgroups per CU; 4096^2
//     is exactly one resident round of 512 workgroups.
// Pipeline per K-tile (one barrier, in the middle, as in sgemm_pipe_kernel): fragments of k-group
// 1 are fetched under the MFMAs of group 0; the barrier publishes tile t+1 (its DMAs were issued
// a full tile earlier; hipcc's vmcnt(0) in front of the barrier is then already satisfied);
// right after it the DMAs of tile t+2 go into the buffer tile t-1 just vacated.
//
// EDGE = the tile may stick out of C (M % 256 or N % 128 != 0).  Out-of-range A rows and B columns are
// fetched from CLAMPED addresses (row M-1, columns 0..3): whatever lands in those LDS slots only ever
// reaches C rows >= M or columns >= N, which the guarded epilogue does not store — no zero fill, no
// extra work in the loop.
// KTAIL = K % 16 != 0 or N % 4 != 0 (its own instantiation: the tail bookkeeping costs the aligned case
// 0.5-1 % when it is merely a run-time flag, profiles/r01/gemm_ktail_ab.log).
// Nothing has to be 16-byte aligned: global_load_lds_dwordx4 takes 4-byte-aligned global addresses at
// the aligned rate (tools/explore/dma_unaligned.hip, profiles/r03/dma_unaligned.log), so odd K, N, row
// strides and base addresses run here as they are (round 3; before, such operands were first copied
// into padded workspaces).  What K % 4 and N % 4 leave over is handled in the LAST K-tile only: a
// 4-float chunk cut by the end of a row is fetched so that it ENDS with the row and moved into place
// in LDS (zero_tail) — A's K tail must hold zeros, and B's last row must not be read past the matrix.
//
// PRIO = alternate the wave priority between the two workgroups that share a CU.  Each SIMD holds one wave of
// each; at equal priority the SIMD's issue arbitration favours the OLDER wave, so the first-dispatched workgroup
// runs ahead, finishes early and leaves its neighbour alone on the CU for the rest of its tile, where one wave
// per SIMD cannot keep the matrix pipe full (tools/gemm_probe.py: co-resident workgroups ending 1.63 M and
// 2.39 M cycles after their start).  With PRIO every wave reads its slot in the SIMD (HW_ID.WAVE_ID bit 0: the
// two resident waves differ in it) and raises / drops s_setprio every g.prio_period K-tiles in opposite phase to
// its neighbour: each workgroup is the favoured one half of the time, both finish together.
// The body of the LDS-DMA GEMM: one 256 x 128 tile of A . B over `K` inner elements starting at the pointers given
// (A: the tile's first k column of the matrix, B: its first k row), accumulators handed to `epilogue(acc)`.  A device
// function so that two kernels share it: sgemm_dma_kernel (one whole tile per workgroup) and sgemm_streamk_kernel
// (a workgroup walks a contiguous range of (tile, k) iterations, i.e. a few SEGMENTS of tiles).  `K` is this
// segment's inner length: a K tail (K % 16) exists only in the segment that holds the tile's last k.
constexpr int kDmaBM = 256, kDmaBN = 128, kDmaBK = 16, kDmaTM = 4, kDmaTN = 2;
typedef v16f DmaAcc[kDmaTM][kDmaTN];

template <bool EDGE, bool KTAIL, bool PRIO>
__device__ __forceinline__ void dma_gemm_segment(const GemmArgs &g, const float *A, const float *B, const unsigned m0,
                                                 const unsigned n0, const unsigned K, DmaAcc &acc, float **lds_out = nullptr) {
    constexpr int BM = kDmaBM, BN = kDmaBN, BK = kDmaBK;
    constexpr int WM = 128, WN = 64, TM = kDmaTM, TN = kDmaTN;
    constexpr int A_SZ = BM * BK, B_SZ = BK * BN;   // floats per buffer: 4096 + 2048

    __shared__ __attribute__((aligned(16))) float smem[3 * (A_SZ + B_SZ)];
    float *const As = smem;
    float *const Bs = smem + 3 * A_SZ;
    if (lds_out) *lds_out = smem;   // stream-K folds its partial tiles through the same 72 KiB once the K loop is over

    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned li = lane & 31, lh = lane >> 5;
    const unsigned wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;

    // DMA source pointers of this lane for K-tile 0 (advanced by BK / BK rows per tile).
    // A: the wave moves 4 chunks of 16 rows x 64 B; lane = (row_in_chunk, slot); slot p of row r
    //    fetches k-chunk p ^ ((r >> 2) & 3).
    // B: the wave moves 2 chunks of 2 k-rows x 512 B, straight row-major.
    // K tail: the last K-tile may hold only kr = 1..15 valid k.  Its out-of-range slots are fetched from
    // clamped addresses (k-chunk 0 / B row kr-1: valid memory) and overwritten with zeros by the lane
    // that DMA'd them, after they have landed and before the barrier that publishes the tile — the loop
    // itself is unchanged.
    const unsigned nk = (K + BK - 1) / BK;
    const unsigned kr = K - (nk - 1) * BK;          // 16 = no tail
    const float *a_src[4];
    unsigned a_q[4];                                   // k-chunk (0..3) this lane fetches for chunk c
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const unsigned r = (wave * 4 + c) * 16 + (lane >> 2);
        const unsigned q = (lane & 3) ^ ((r >> 2) & 3);
        unsigned grow = m0 + r;
        if (EDGE && grow >= g.M) grow = g.M - 1;
        a_src[c] = A + (size_t)grow * g.lda + q * 4;
        a_q[c] = q;
    }
    const float *b_src[2];
    unsigned b_k[2];                                   // k row (0..15) this lane fetches for chunk c
    unsigned b_cut = 0;                                // N % 4 != 0 and this lane's 4 columns straddle N: how many are inside (1..3)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const unsigned krow = (wave * 2 + c) * 2 + (lane >> 5);
        unsigned gcol = n0 + (lane & 31) * 4;
        if (EDGE && gcol + 4 > g.N) {
            if (gcol >= g.N) gcol = 0;                 // wholly outside: any valid address (N >= 4)
            else b_cut = g.N - gcol;                   // straddling: fetched as it is — the floats past N are the next row's
        }                                              // first ones (the very last row of B: see dma_tile), C columns >= N
        b_src[c] = B + (size_t)krow * g.ldb + gcol;
        b_k[c] = krow;
    }
    const size_t b_step = (size_t)BK * g.ldb;

    auto dma_tile = [&](unsigned buf, bool tail) {   // tail: this is the last tile and kr < 16 (uniform)
        float *as = As + buf * A_SZ + wave * 1024;   // 4 chunks x 256 floats per wave
        float *bs = Bs + buf * B_SZ + wave * 512;    // 2 chunks x 256 floats per wave
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float *src = a_src[c];
            if (KTAIL && tail) {
                const unsigned k0 = a_q[c] * 4;
                if (k0 + 4 > kr) src -= k0 + 4 - kr;           // cut by K (K % 4 != 0), or wholly past it: the 4 floats that END
            }                                                  // with the row instead (never past the end of A); zero_tail moves
                                                               // the valid ones into place / zeroes the slot
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(as + c * 256), 16, 0, 0);
            a_src[c] += BK;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float *src = b_src[c];
            if (KTAIL && tail && b_k[c] + 1 >= kr) {           // B's last row (k = K - 1) or below it
                if (b_k[c] >= kr) src -= (size_t)(b_k[c] - (kr - 1)) * g.ldb;
                src -= b_cut ? 4 - b_cut : 0;                  // a straddling chunk of the LAST row ends with the matrix instead
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(bs + c * 256), 16, 0, 0);
            b_src[c] += b_step;
        }
    };
    // after the tail tile has landed (vmcnt(0)): zero the slots this lane fetched from clamped addresses; a slot that was
    // fetched `s` floats early (cut by K, or by N in B's last row) has its floats moved down by s and zeros behind them
    auto shifted = [](v4f v, unsigned s) {   // s = 1..3
        return v4f{s == 1 ? v[1] : s == 2 ? v[2] : v[3], s == 1 ? v[2] : s == 2 ? v[3] : 0.0f, s == 1 ? v[3] : 0.0f, 0.0f};
    };
    auto zero_tail = [&](unsigned buf) {
        float *as = As + buf * A_SZ + wave * 1024 + lane * 4;
        float *bs = Bs + buf * B_SZ + wave * 512 + lane * 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned k0 = a_q[c] * 4;
            v4f *slot = (v4f *)(as + c * 256);
            if (k0 >= kr) *slot = v4f{0, 0, 0, 0};
            else if (k0 + 4 > kr) *slot = shifted(*slot, k0 + 4 - kr);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            v4f *slot = (v4f *)(bs + c * 256);
            if (b_k[c] >= kr) *slot = v4f{0, 0, 0, 0};
            else if (b_cut && b_k[c] + 1 == kr) *slot = shifted(*slot, 4 - b_cut);
        }
    };
    // the last K-tile needs the fix-up when K % 16 != 0, and when N % 4 != 0 (B's last row must not be read past its end)
    const bool has_tail = KTAIL && (kr < BK || (g.N & 3u) != 0);

#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    struct Frag {
        v4f a4[TM];
        float bv[4][TN];
    };
    // per-lane LDS offsets (floats): row (wm0 + i*32 + li), physical slot (kg*2 + lh) ^ sw
    const unsigned sw = (li >> 2) & 3;
    const unsigned a_off0 = (wm0 + li) * BK + ((lh ^ sw) * 4);          // k group 0
    const unsigned a_off1 = (wm0 + li) * BK + (((2 + lh) ^ sw) * 4);    // k group 1
    const unsigned b_off = (4 * lh) * BN + wn0 + li;
    auto read_frag = [&](Frag &f, unsigned buf, int kg) {
        const float *as = As + buf * A_SZ + (kg ? a_off1 : a_off0);
        const float *bs = Bs + buf * B_SZ + b_off + kg * 8 * BN;
#pragma unroll
        for (int i = 0; i < TM; ++i) f.a4[i] = *(const v4f *)(as + i * 32 * BK);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < TN; ++j) f.bv[s][j] = bs[s * BN + j * 32];
    };
    auto mfma_group = [&](const Frag &f) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a4[i][s], f.bv[s][j], acc[i][j], 0, 0, 0);
    };

    Frag f0, f1;
    dma_tile(0, has_tail && nk == 1);
    if (nk > 1) dma_tile(1, has_tail && nk == 2);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (has_tail && nk <= 2) zero_tail(nk - 1);
    __syncthreads();
    read_frag(f0, 0, 0);

    // One K-tile.  DMA / NEXT are compile-time so that the steady-state iteration is a single
    // basic block per half and the issue-order hints below can interleave across it.
    unsigned cur = 0;
    unsigned kt = 0;
    auto k_tile = [&](auto dma_c, auto next_c) {
        constexpr bool DMA = decltype(dma_c)::value, NEXT = decltype(next_c)::value;
        const unsigned nxt = (cur == 2) ? 0 : cur + 1;
        const unsigned nn = (nxt == 2) ? 0 : nxt + 1;
        // first half: MFMAs of k group 0; the 8 LDS reads of group 1 go out one behind each of the
        // first 8 MFMAs (in the shadow of a running MFMA instead of as one burst)
        read_frag(f1, cur, 1);
        mfma_group(f0);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
        // keep the wait + barrier BEHIND the 32 MFMAs (hipcc would hoist them to the top of the
        // tile: register-only MFMAs are not ordered by a "memory" clobber)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (NEXT) {
            // tile kt+1 (DMA issued one tile ago) must have landed for every wave before anyone
            // reads it; the same barrier tells everyone that tile kt-1's buffer is free
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (has_tail && kt + 2 == nk && nk > 2) zero_tail(nxt);   // tile kt+1 is the ragged last one (nk <= 2: the prologue fixed it — moving a slot's floats twice would be wrong)
            __syncthreads();
        }
        // second half: MFMAs of k group 1 with 6 LDS-DMA issues (tile kt+2 -> the buffer tile kt-1
        // vacated) and the 8 LDS reads of the next tile's group 0, each behind its own MFMA
        if constexpr (DMA) dma_tile(nn, has_tail && kt + 3 == nk);
        if constexpr (NEXT) read_frag(f0, nxt, 0);
        mfma_group(f1);
        if constexpr (DMA) {
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read (global_load_lds)
            }
        }
        if constexpr (NEXT) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 32, 0);
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    };
    using T = std::true_type;
    using F = std::false_type;
    // (a compile-time "tail DMA" variant of k_tile, peeled into its own iteration, pushed the kernel
    // to 256 VGPRs + 600 B/lane of scratch and 4096^3 from 144 to 130 TFLOP/s: the flag stays uniform
    // run-time state)
    if constexpr (PRIO) {
        // hwreg(HW_REG_HW_ID = 4, offset 0, size 4) = WAVE_ID: this wave's slot among the SIMD's resident waves
        unsigned phase = (unsigned)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 1u;
        while (kt + 2 < nk) {
            if (phase)
                __builtin_amdgcn_s_setprio(1);
            else
                __builtin_amdgcn_s_setprio(0);
            const unsigned left = nk - 2 - kt;
            const unsigned stop = kt + (left < g.prio_period ? left : g.prio_period);
            for (; kt < stop; ++kt) k_tile(T{}, T{});
            phase ^= 1u;
        }
        __builtin_amdgcn_s_setprio(0);
    } else {
        for (; kt + 2 < nk; ++kt) k_tile(T{}, T{});
    }
    if (kt + 1 < nk) k_tile(F{}, T{});
    k_tile(F{}, F{});

}

// element (i, j, r) of a lane's accumulators <-> row / column inside the 256 x 128 tile
__device__ __forceinline__ void dma_acc_coords(unsigned &row0, unsigned &col0) {
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    row0 = (wave >> 1) * 128 + 4 * (lane >> 5);   // + i * 32 + (r & 3) + 8 * (r >> 2)
    col0 = (wave & 1) * 64 + (lane & 31);         // + j * 32
}

template <bool EDGE, bool COHERENT>
__device__ __forceinline__ void dma_store_tile(const GemmArgs &g, float *C, unsigned m0, unsigned n0, const DmaAcc &acc) {
    unsigned row0, col0;
    dma_acc_coords(row0, col0);
#pragma unroll
    for (int i = 0; i < kDmaTM; ++i)
#pragma unroll
        for (int j = 0; j < kDmaTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned row = m0 + row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                const unsigned col = n0 + col0 + j * 32;
                if (!EDGE || (row < g.M && col < (g.n_store ? g.n_store : g.N))) {
                    if constexpr (COHERENT)   // performed at the memory side, past this XCD's L2
                        __hip_atomic_store(&C[(size_t)row * g.ldc + col], acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    else
                        __builtin_nontemporal_store(acc[i][j][r], &C[(size_t)row * g.ldc + col]);   // C is never re-read here
                }
            }
}

template <bool EDGE, bool KTAIL, bool PRIO = false>
__global__ __launch_bounds__(256, 2) void sgemm_dma_kernel(GemmArgs g) {
    if (g.K_last && blockIdx.z + 1 == gridDim.z) g.K = g.K_last;   // uniform: split-K remainder chunk
    unsigned tile_m, tile_n;
    tile_coords(g, blockIdx.x, tile_m, tile_n);
    const unsigned m0 = tile_m * kDmaBM, n0 = tile_n * kDmaBN;
    unsigned long long probe_c0 = 0, probe_w0 = 0;
    probe_begin(g, probe_c0, probe_w0);
    const float *A = g.A + (size_t)blockIdx.z * g.stride_a;
    const float *B = g.B + (size_t)blockIdx.z * g.stride_b;
    float *C = g.C + (size_t)blockIdx.z * g.stride_c;
    DmaAcc acc;
    dma_gemm_segment<EDGE, KTAIL, PRIO>(g, A, B, m0, n0, g.K, acc);
    if (g.progress)
        dma_store_tile<EDGE, true>(g, C, m0, n0, acc);
    else
        dma_store_tile<EDGE, false>(g, C, m0, n0, acc);
    probe_end(g, probe_c0, probe_w0);
    progress_signal(g);
}

// ---- stream-K ------------------------------------------------------------------------------------------------
// One tile per workgroup wastes the machine whenever the number of tiles is not a multiple of what is resident
// (512 workgroups): 3000^3 is 288 tiles, 2048^3 is 128, 4097^3 (padded) 561 — a third wave of 49 tiles behind two
// full ones.  Here the unit of work is the K-TILE: the T x nk iterations of the product, tile-major, are cut into
// gridDim.x equal contiguous ranges, one per workgroup (Osama et al., "Stream-K", arXiv 2301.03598).  A range is at
// most: the tail of one tile (its k from kb > 0 on), some whole tiles, the head of one more (k from 0 up to ke < nk).
//   whole tile            stored straight to C
//   segment with kb > 0   its accumulators go to this workgroup's slot of the workspace, then flag[w] = seq
//   head segment (kb = 0, ke < nk)   the tile's FINISHER: it waits for the workgroups holding the rest of the tile —
//                         w + 1, w + 2, ... which each met that tile FIRST in their range and so posted long ago —
//                         adds their partials in that order and stores C.
// The sum order of every element is fixed by the schedule: run to run the result is bit-identical.  Deadlock-free
// even if not all workgroups are resident at once: a workgroup only ever waits for HIGHER-numbered ones, whose
// contribution is the first thing they compute; the lower-numbered residents therefore finish and make room.
// Partials and flags travel by memory-side stores / loads (np_internal.h: the XCD L2s are not coherent).
struct StreamKArgs {
    float *workspace;       // gridDim.x slots of 256 x 128 floats: a workgroup's partial tile, row-major
    unsigned *flags;        // gridDim.x words; flag[w] == seq <=> workgroup w's partial of THIS launch is in its slot
    unsigned seq;
    unsigned nk;            // k-tiles per tile
    unsigned long long iters_total;   // tiles * nk
    unsigned *error_word;   // np::device_error_word(): a finisher whose poll budget runs out ORs kErrStreamK in (reported at np_sync)
};

template <bool EDGE, bool KTAIL, bool PRIO = false>
__global__ __launch_bounds__(256, 2) void sgemm_streamk_kernel(GemmArgs g, StreamKArgs sk) {
    const unsigned long long G = gridDim.x, w = blockIdx.x;
    const unsigned long long it0 = sk.iters_total * w / G, it1 = sk.iters_total * (w + 1) / G;
    bool first = true;
    for (unsigned long long it = it0; it < it1;) {
        const unsigned t = (unsigned)(it / sk.nk), kb = (unsigned)(it - (unsigned long long)t * sk.nk);
        const unsigned long long tile_end = (unsigned long long)(t + 1) * sk.nk;
        const unsigned ke = (unsigned)((it1 < tile_end ? it1 : tile_end) - (unsigned long long)t * sk.nk);
        unsigned tile_m, tile_n;
        tile_coords(g, t, tile_m, tile_n);
        const unsigned m0 = tile_m * kDmaBM, n0 = tile_n * kDmaBN;
        const unsigned K = ke == sk.nk ? g.K - kb * kDmaBK : (ke - kb) * kDmaBK;
        if (!first) __syncthreads();   // the previous segment's last LDS reads are done before this one's DMAs land
        first = false;
        DmaAcc acc;
        float *lds = nullptr;
        dma_gemm_segment<EDGE, KTAIL, PRIO>(g, g.A + (size_t)kb * kDmaBK, g.B + (size_t)kb * kDmaBK * g.ldb, m0, n0, K, acc, &lds);
        const unsigned long long next_it = it1 < tile_end ? it1 : tile_end;
        unsigned row0, col0;
        dma_acc_coords(row0, col0);
        // the epilogue's addresses all derive from these two: made opaque HERE so that none of that arithmetic is hoisted
        // above the K loop, where it lived in (and spilled from) registers the loop needs — 282 VGPRs spilled, and the
        // scratch allocation throttled wave dispatch (2048^3: 222 us against 139 for the tile form)
        asm volatile("" : "+v"(row0), "+v"(col0));
        // finisher (this segment is the HEAD of a tile that ends in other workgroups): the rest of the tile is in
        // workgroups w + 1 ... w_last, the one holding its last k-tile.  All their flags first — no data yet —
        unsigned long long w_last = w;
        if (kb == 0 && ke != sk.nk) {
            w_last = (tile_end - 1) * G / sk.iters_total;                         // candidate owner of iteration tile_end - 1 ...
            while (sk.iters_total * (w_last + 1) / G < tile_end) ++w_last;        // ... exact under the floor()s above
            while (sk.iters_total * w_last / G >= tile_end) --w_last;
            // one lane per flag (they were posted long ago: polled one after the other, each costs a round trip to memory)
            // (bounded: the schedule guarantees the flags — 2^26 polls, s_sleep between them, are tens of seconds: a wait
            // that long can only be a fault elsewhere, and a wrong tile is a lesser evil than a queue that never drains)
            // — and NOT a silent one: the lane that gives up raises the process's device-error word, np_sync / the read-back
            // of C then return NP_ERR_DEVICE instead of NP_OK with a corrupt tile)
            for (unsigned p = (unsigned)w + 1 + threadIdx.x; p <= (unsigned)w_last; p += 256) {
                unsigned spins = 0;
                while (np::dev::coherent_load(sk.flags + p) != sk.seq && spins < (1u << 26)) {
                    __builtin_amdgcn_s_sleep(8);
                    ++spins;
                }
                if (spins == (1u << 26) && sk.error_word)
                    __hip_atomic_fetch_or(sk.error_word, np::kErrStreamK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __syncthreads();   // (also: every wave is done with the K loop's LDS before the fold's DMAs land in it)
        }
        const unsigned n_partials = (unsigned)(w_last - w);
        // A partial tile travels LANE-MAJOR: group G = (i * 2 + j) * 4 + q holds acc[i][j][4q .. 4q + 3] of all 256 lanes,
        // lane t's four floats at slot[(G * 256 + t) * 4].  The writer stores a float4 per lane per group, coalesced; the
        // finisher fetches half a partial (16 groups = 64 KiB) with 16 LDS-DMA loads per lane — no registers, all in
        // flight at once — and every lane then reads back exactly the 16-byte slots it fetched itself (no barrier).
        // Fetching the partials into registers eight floats at a time (all the ragged instantiations have to spare) cost
        // ~5 us per partial at the very end of the launch, where nothing is left to hide it behind.
        const bool to_slot = kb != 0;
        if (to_slot) {
            // (ONE running pointer, kept opaque: 32 precomputed addresses were 64 VGPRs the K loop had to spill around)
            float *dst = sk.workspace + (size_t)w * (kDmaBM * kDmaBN) + (size_t)threadIdx.x * 4;
#pragma unroll
            for (int i = 0; i < kDmaTM; ++i)
#pragma unroll
                for (int j = 0; j < kDmaTN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const v4f v{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst), "v"(v) : "memory");   // memory-side, as coherent_store
                        dst += 256 * 4;
                        asm volatile("" : "+v"(dst));
                    }
        } else {
            const unsigned wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
            for (unsigned q = 0; q < n_partials; ++q) {   // workgroup order = k order: the same sum every run
                const float *peer = sk.workspace + (size_t)(w + 1 + q) * (kDmaBM * kDmaBN) + (size_t)threadIdx.x * 4;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous half's LDS reads are complete
#pragma unroll
                    for (int gi = 0; gi < 16; ++gi) {
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)peer,
                                                         (__attribute__((address_space(3))) void *)(lds + (gi * 256 + wave * 64) * 4), 16, 0,
                                                         16 /* sc1: performed at the memory side, as coherent_load */);
                        peer += 256 * 4;
                        asm volatile("" : "+v"(peer));
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int gi = 0; gi < 16; ++gi) {
                        const v4f p = *(const v4f *)(lds + (gi * 256 + threadIdx.x) * 4);
                        const int grp = half * 16 + gi, i = grp / (kDmaTN * 4), j = (grp / 4) % kDmaTN, qq = grp % 4;
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][4 * qq + e] += p[e];
                        if (gi % 2 == 1) __builtin_amdgcn_sched_barrier(0);   // two 16-byte reads in flight: the ragged instantiations have no more registers to spare
                    }
                }
            }
            // the tile of C
            const unsigned lim_n = g.n_store ? g.n_store : g.N;
#pragma unroll
            for (int i = 0; i < kDmaTM; ++i)
#pragma unroll
                for (int j = 0; j < kDmaTN; ++j) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned row = m0 + row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                        const unsigned col = n0 + col0 + j * 32;
                        if (!EDGE || (row < g.M && col < lim_n)) __builtin_nontemporal_store(acc[i][j][r], &g.C[(size_t)row * g.ldc + col]);   // as the tile kernel: no other workgroup reads C
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        if (n_partials) {   // consumed: the flags end the launch as they began it
            __syncthreads();
            if (threadIdx.x == 0)
                for (unsigned long long p = w + 1; p <= w_last; ++p) np::dev::coherent_store(sk.flags + p, 0u);
        }
        if (to_slot) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (threadIdx.x == 0) np::dev::coherent_store(sk.flags + w, sk.seq);
        }
        it = next_it;
    }
}

// ---- mid-size products: smaller LDS-DMA tiles, K split INSIDE the launch ---------------------------------------
// 1024^3 is 32 tiles of 256 x 128: an eighth of the machine.  The forms it had (profiles/r03/gemm_sweep_final.log: 30 us,
// 72 TFLOP/s, where 13.7 us is the matrix cores' time) were 64 x 64 register-staged tiles (LDS stores, one accumulator per
// wave), or 256 x 128 tiles with K cut into batch entries and a second launch folding 8 partial matrices.  What a PHP
// caller multiplies is this size, not 4096^3.  So the LDS-DMA pipeline of sgemm_dma_kernel exists
// for smaller tiles as well — 128 x 128, 128 x 64, 64 x 64, the same staging (global_load_lds, XOR-swizzled A slots,
// one barrier per K-tile), with as many LDS buffers as it takes to keep a DMA ~1.5 us ahead of its consumer when a K-tile
// is only 512-2048 cycles of MFMA — and K can be split S ways inside ONE launch:
//   * workgroup (tile, s) walks K-chunk s of its tile (chunks are multiples of 16: only the last may be ragged);
//   * S == 1: it stores its tile.  S > 1: it writes its partial tile to the workspace lane-major with memory-side
//     stores, reports to the tile's counter, waits until all S partials are there, and then folds ITS SHARE of the tile —
//     register groups g with g % S == s (S a power of two that divides the group count) — over the S partials in chunk order (its own from registers: the same bits) and
//     stores that share of C.  Every workgroup folds 1 / S of a tile: the fold is as parallel as the product, its traffic
//     2 x S x |C| in all, nothing is serial at the end (stream-K's finisher folds its tile's partials alone, one after
//     the other: 3 us each at the very end of the launch), no second launch.  The sum order is fixed: deterministic.
//   * the counter is a ticket (np::next_tickets): arrivals count to S, departures to 2 S, the last one out zeroes it.
// All S workgroups of a tile must be resident at once (they wait for each other): the launcher keeps tiles x S within
// what the device holds (2 workgroups per CU by registers and LDS), like stream-K.
// Where a small tile loses (rocprofv3, profiles/r04/gemm_mid_pmc.txt): the matrix pipe is busy 0.69 of the CU-busy cycles on
// 64 x 64 tiles (one accumulator per wave), 0.76 on 128 x 64 (two), 0.87 on 128 x 128 (four), 0.95 on 256 x 128 (eight).  That
// reads like dependent-MFMA stalls, but independent accumulator copies (step s of a k-group into copy s % 4, summed after the
// loop) changed nothing — 1024^3 21.1-21.7 us before, 22.2-22.9 after: a dependent v_mfma_f32_32x32x2 DOES issue back to
// back.  What the ratio tracks is ~200-300 cycles per K-tile that are not MFMA (the barrier, the waits around it) against 512 /
// 1024 / 2048 / 4096 cycles of MFMA: hence the deeper K-tile (BK = 32) of the small shapes.
template <int BM_, int BN_, int NBUF_, int BK_ = 16, int KW_ = 1>
struct DmasShape {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, NBUF = NBUF_;
    static constexpr int KW = KW_;                       // waves per tile position (1 or 2): KW == 2 = 8 waves, see the kernel
    static constexpr int WAVES = 4 * KW, THREADS = 256 * KW;
    static constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN = WN / 32;
    static constexpr int SLOTS = BK / 4;                 // 16-byte slots per A row in LDS (4 or 8)
    static constexpr int CHUNK_ROWS = 64 / SLOTS;        // A rows one DMA instruction of a wave covers (16 or 8)
    static constexpr int AC = BM / CHUNK_ROWS / WAVES;   // A: DMA instructions per wave per K-tile
    static constexpr int BC = BK * BN / 256 / WAVES;     // B: 256-float chunks per wave per K-tile
    static constexpr int NG = BK / 8;                    // k-groups of 8 per K-tile (the MFMA steps of one group: 4)
    static constexpr int NGW = NG / KW;                  // ... of which one wave runs this many
    static_assert(AC >= 1 && BC >= 1 && NGW >= 2 && NGW * KW == NG, "tile too small for this many waves");
    static constexpr int A_SZ = BM * BK, B_SZ = BK * BN;
    static constexpr int GROUPS = TM * TN * 4;  // float4 register groups per lane (the unit of the fold)
};

struct DmasArgs {
    unsigned S;             // K chunks per tile
    unsigned Kc;            // chunk length (multiple of the K-tile depth)
    float *workspace;       // S > 1: tiles x S partial tiles of BM x BN floats, lane-major
    unsigned *counters;     // S > 1: one zeroed ticket per tile
    unsigned *error_word;   // np::device_error_word(): a wait for the sibling chunks that runs out of polls is reported (np_sync)
};

// K-tile depth BK = 16 (two k-groups, as sgemm_dma_kernel) or 32 (four): the barrier and the waits around it cost ~200-300
// cycles per K-tile whatever its depth, against 512 cycles of MFMA per 16-deep K-tile of a 64 x 64 tile.  A rows are BK
// floats = BK / 4 slots; slot p of row r holds k-chunk p ^ f(r) with f(r) = (r >> 2) & 3 for 4 slots, (r >> 1) & 7 for 8 —
// either way the 16 lanes of a ds_read_b128 pass (16 consecutive rows, one k-chunk) land on 16 different 16-byte bank groups.
// KW = 2: eight waves; waves w and w + 4 compute the same 32 x 32 (x TM x TN) block, w over the even k-groups of every K-tile,
// w + 4 over the odd ones, and the DMAs are dealt over all eight (DmasShape5 says where that pays).
// The fold of one register group: the S partials of the sibling chunks, ALL in flight at once (one memory round trip, not
// S - 1 of them one behind the other; worth little, as it turned out — 512^3 on 128 x 128 tiles with S = 8 19.4 -> 16.6 us, on
// 64 x 64 with S = 4 12.6 -> 12.3: profiles/r04/gemm_mid_sweep_forced.log / _forced2.log — the ~5 us a fold costs are the chain
// store -> acknowledge -> ticket -> poll -> load, one memory round trip each), summed in chunk order — the same sum every run —
// with this workgroup's own partial taken from its registers (the same bits; its slot in the workspace is loaded like the
// others and ignored: no branch around the loads, so the compiler has no copies to place between a load and its wait).
template <int SV>
__device__ inline v4f fold_partials(const float *base, unsigned stride, unsigned chunk, v4f own) {
    v4f part[SV];
#pragma unroll
    for (int c = 0; c < SV; ++c) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(part[c]) : "v"(base + (size_t)c * stride) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int c = 0; c < SV; ++c) asm volatile("" : "+v"(part[c]));   // (nothing below may be scheduled above the wait)
    v4f sum = chunk == 0 ? own : part[0];
#pragma unroll
    for (int c = 1; c < SV; ++c) {
        const v4f p = (unsigned)c == chunk ? own : part[c];
#pragma unroll
        for (int e = 0; e < 4; ++e) sum[e] += p[e];
    }
    return sum;
}

template <class SH, bool EDGE, bool KTAIL>
__global__ __launch_bounds__(SH::THREADS, 2) void sgemm_dmas_kernel(GemmArgs g, DmasArgs d) {
    constexpr int BM = SH::BM, BN = SH::BN, BK = SH::BK, NBUF = SH::NBUF, WM = SH::WM, WN = SH::WN, TM = SH::TM, TN = SH::TN;
    constexpr int AC = SH::AC, BC = SH::BC, A_SZ = SH::A_SZ, B_SZ = SH::B_SZ, SLOTS = SH::SLOTS, CHUNK_ROWS = SH::CHUNK_ROWS;
    constexpr int NG = SH::NGW, KW = SH::KW;   // NG: the k-groups of a K-tile THIS wave runs
    __shared__ __attribute__((aligned(16))) float smem[NBUF * (A_SZ + B_SZ)];
    float *const As = smem;
    float *const Bs = smem + NBUF * A_SZ;

    const unsigned tile = blockIdx.x / d.S, chunk = blockIdx.x - tile * d.S;
    unsigned tile_m, tile_n;
    tile_coords(g, tile, tile_m, tile_n);
    const unsigned m0 = tile_m * BM, n0 = tile_n * BN;
    const unsigned k_begin = chunk * d.Kc;
    const unsigned K = (g.K - k_begin < d.Kc) ? g.K - k_begin : d.Kc;   // this chunk's inner length (>= 1 by construction)
    const float *A = g.A + (size_t)blockIdx.z * g.stride_a + k_begin;
    const float *B = g.B + (size_t)blockIdx.z * g.stride_b + (size_t)k_begin * g.ldb;
    float *C = g.C + (size_t)blockIdx.z * g.stride_c;

    const unsigned tid = threadIdx.x;
    const unsigned lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned li = lane & 31, lh = lane >> 5;
    // KW == 2: waves w and w + 4 share a tile position; w + 4 runs the odd k-groups of every K-tile (kh = 1), w the even ones
    const unsigned wpos = wave & 3u, kh = wave >> 2;
    const unsigned wm0 = (wpos >> 1) * WM, wn0 = (wpos & 1) * WN;
    auto swz = [](unsigned r) { return SLOTS == 4 ? (r >> 2) & 3u : (r >> 1) & 7u; };

    // DMA sources (see dma_gemm_segment: the same scheme with AC / BC chunks per wave)
    const unsigned nk = (K + BK - 1) / BK;
    const unsigned kr = K - (nk - 1) * BK;          // BK = no tail
    const float *a_src[AC];
    unsigned a_q[AC];
#pragma unroll
    for (int c = 0; c < AC; ++c) {
        const unsigned r = (wave * AC + c) * CHUNK_ROWS + lane / SLOTS;
        const unsigned q = (lane % SLOTS) ^ swz(r);
        unsigned grow = m0 + r;
        if (EDGE && grow >= g.M) grow = g.M - 1;
        a_src[c] = A + (size_t)grow * g.lda + q * 4;
        a_q[c] = q;
    }
    const float *b_src[BC];
    unsigned b_k[BC];
    unsigned b_cut = 0;
#pragma unroll
    for (int c = 0; c < BC; ++c) {
        const unsigned off = (wave * BC + c) * 256 + lane * 4;   // float offset inside the [BK][BN] tile
        const unsigned krow = off / BN;
        unsigned gcol = n0 + off % BN;
        if (EDGE && gcol + 4 > g.N) {
            if (gcol >= g.N) gcol = 0;
            else b_cut = g.N - gcol;
        }
        b_src[c] = B + (size_t)krow * g.ldb + gcol;
        b_k[c] = krow;
    }
    const size_t b_step = (size_t)BK * g.ldb;

    auto dma_tile = [&](unsigned buf, bool tail) {
        float *as = As + buf * A_SZ + wave * (AC * 256);
        float *bs = Bs + buf * B_SZ + wave * (BC * 256);
#pragma unroll
        for (int c = 0; c < AC; ++c) {
            const float *src = a_src[c];
            if (KTAIL && tail) {
                const unsigned k0 = a_q[c] * 4;
                if (k0 + 4 > kr) src -= k0 + 4 - kr;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(as + c * 256), 16, 0, 0);
            a_src[c] += BK;
        }
#pragma unroll
        for (int c = 0; c < BC; ++c) {
            const float *src = b_src[c];
            if (KTAIL && tail && b_k[c] + 1 >= kr) {
                if (b_k[c] >= kr) src -= (size_t)(b_k[c] - (kr - 1)) * g.ldb;
                src -= b_cut ? 4 - b_cut : 0;
            }
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(bs + c * 256), 16, 0, 0);
            b_src[c] += b_step;
        }
    };
    auto shifted = [](v4f v, unsigned s) {
        return v4f{s == 1 ? v[1] : s == 2 ? v[2] : v[3], s == 1 ? v[2] : s == 2 ? v[3] : 0.0f, s == 1 ? v[3] : 0.0f, 0.0f};
    };
    auto zero_tail = [&](unsigned buf) {
        float *as = As + buf * A_SZ + wave * (AC * 256) + lane * 4;
        float *bs = Bs + buf * B_SZ + wave * (BC * 256) + lane * 4;
#pragma unroll
        for (int c = 0; c < AC; ++c) {
            const unsigned k0 = a_q[c] * 4;
            v4f *slot = (v4f *)(as + c * 256);
            if (k0 >= kr) *slot = v4f{0, 0, 0, 0};
            else if (k0 + 4 > kr) *slot = shifted(*slot, k0 + 4 - kr);
        }
#pragma unroll
        for (int c = 0; c < BC; ++c) {
            v4f *slot = (v4f *)(bs + c * 256);
            if (b_k[c] >= kr) *slot = v4f{0, 0, 0, 0};
            else if (b_cut && b_k[c] + 1 == kr) *slot = shifted(*slot, 4 - b_cut);
        }
    };
    // only the chunk that ends with K can be ragged; N % 4 != 0 concerns B's very last row, which that chunk holds
    const bool last_chunk = k_begin + K == g.K;
    const bool has_tail = KTAIL && last_chunk && (kr < BK || (g.N & 3u) != 0);

    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    struct Frag {
        v4f a4[TM];
        float bv[4][TN];
    };
    const unsigned a_row = (wm0 + li) * BK, fsw = swz(li);   // (wm0 and i * 32 are multiples of 32: the swizzle of a row depends on li only)
    const unsigned b_off = (4 * lh) * BN + wn0 + li;
    auto read_frag = [&](Frag &f, unsigned buf, int kg) {
        const float *as = As + buf * A_SZ + a_row + (((unsigned)(kg * 2) + lh) ^ fsw) * 4;
        const float *bs = Bs + buf * B_SZ + b_off + kg * 8 * BN;
#pragma unroll
        for (int i = 0; i < TM; ++i) f.a4[i] = *(const v4f *)(as + i * 32 * BK);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < TN; ++j) f.bv[s][j] = bs[s * BN + j * 32];
    };
    auto mfma_group = [&](const Frag &f) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a4[i][s], f.bv[s][j], acc[i][j], 0, 0, 0);
    };
    constexpr int kMfmaPerGroup = 4 * TM * TN, kReadsPerGroup = TM + 4 * TN, kDmaPerTile = AC + BC;
    constexpr int kPaired = kReadsPerGroup < kMfmaPerGroup ? kReadsPerGroup : kMfmaPerGroup;

    // prologue: tiles 0 .. NBUF - 2 in flight
    const bool tail_in_prologue = has_tail && nk <= (unsigned)(NBUF - 1);
#pragma unroll
    for (int t = 0; t < NBUF - 1; ++t)
        if ((unsigned)t < nk) dma_tile((unsigned)t, has_tail && (unsigned)t + 1 == nk);
    if (nk >= (unsigned)(NBUF - 1) && !tail_in_prologue) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * kDmaPerTile) : "memory");   // tile 0 has landed
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tail_in_prologue) zero_tail(nk - 1);
    }
    __syncthreads();
    Frag fr[2];
    read_frag(fr[0], 0, (int)kh);

    // One K-tile = NG k-groups.  While group G's MFMAs run, the fragments of group G + 1 (of the NEXT tile behind the last
    // group) are fetched, one LDS read behind each MFMA; the ONE barrier sits behind group NG / 2 - 1: it publishes tile
    // kt + 1 (its DMAs went out NBUF - 1 tiles ago) and frees tile kt - 1's buffer, into which tile kt + NBUF - 1's DMAs go.
    unsigned cur = 0, kt = 0;
    auto k_tile = [&](auto dma_c, auto next_c) {
        constexpr bool DMA = decltype(dma_c)::value, NEXT = decltype(next_c)::value;
        const unsigned nxt = cur + 1 == (unsigned)NBUF ? 0 : cur + 1;
        const unsigned into = cur == 0 ? (unsigned)(NBUF - 1) : cur - 1;   // the buffer tile kt - 1 vacated = (kt + NBUF - 1) % NBUF
#pragma unroll
        for (int G = 0; G < NG; ++G) {
            const bool reads = G + 1 < NG || NEXT;
            if (G + 1 < NG) read_frag(fr[(G + 1) & 1], cur, (G + 1) * KW + (int)kh);
            else if (NEXT) read_frag(fr[0], nxt, (int)kh);
            if (DMA && G == NG / 2) dma_tile(into, has_tail && kt + (unsigned)NBUF == nk);
            mfma_group(fr[G & 1]);
            if (DMA && G == NG / 2) {
#pragma unroll
                for (int q = 0; q < (kDmaPerTile < kMfmaPerGroup ? kDmaPerTile : kMfmaPerGroup); ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read (global_load_lds)
                }
            }
            if (reads) {
#pragma unroll
                for (int q = 0; q < kPaired; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x008, kMfmaPerGroup, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (G == NG / 2 - 1 && NEXT) {
                // tile kt + 1 must have landed for every wave; in the steady state the tiles behind it (kt + 2 .. kt + NBUF - 2)
                // may still be in flight, towards the end nothing else is
                if constexpr (DMA)
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 3) * kDmaPerTile) : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (has_tail && kt + 2 == nk && !tail_in_prologue) zero_tail(nxt);
                __syncthreads();
            }
        }
        cur = nxt;
    };
    using T = std::true_type;
    using F = std::false_type;
    for (; kt + (unsigned)(NBUF - 1) < nk; ++kt) k_tile(T{}, T{});
    for (; kt + 1 < nk; ++kt) k_tile(F{}, T{});
    k_tile(F{}, F{});

    if constexpr (KW == 2) {
        // the two halves of every tile position meet in LDS (lane-major: no bank conflicts); waves 0 .. 3 carry the sum on
        // (kh-0 half + kh-1 half: a fixed order) and do all the stores below, waves 4 .. 7 only keep the barriers company
        __syncthreads();   // (the last K-tile's fragments have been read by everybody)
        float *ex = smem + (size_t)wpos * (TM * TN * 16 * 64) + lane;
        if (kh == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ex[((i * TN + j) * 16 + r) * 64] = acc[i][j][r];
        }
        __syncthreads();
        if (kh == 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += ex[((i * TN + j) * 16 + r) * 64];
        }
    }
    const bool writer = kh == 0;
    const unsigned row0 = (wpos >> 1) * WM + 4 * lh, col0 = (wpos & 1) * WN + li;
    const unsigned lim_n = g.n_store ? g.n_store : g.N;
    if (d.S == 1) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned row = m0 + row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                    const unsigned col = n0 + col0 + j * 32;
                    if (writer && (!EDGE || (row < g.M && col < lim_n))) __builtin_nontemporal_store(acc[i][j][r], &C[(size_t)row * g.ldc + col]);
                }
        return;
    }
    // ---- the fold ----
    // group G = (i * TN + j) * 4 + q holds acc[i][j][4q .. 4q + 3] of all 256 lanes, lane t's float4 at slot[(G * 256 + t) * 4]
    const size_t batch_tiles = (size_t)g.tiles_m * g.tiles_n;
    float *const tile_ws = d.workspace + (((size_t)blockIdx.z * batch_tiles + tile) * d.S) * (size_t)(BM * BN);
    {
        float *dst = tile_ws + (size_t)chunk * (BM * BN) + (size_t)tid * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (writer && ((unsigned)((i * TN + j) * 4 + q) & (d.S - 1)) != chunk) {   // S is a power of two; its own share never leaves the registers
                        const v4f v{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst + ((i * TN + j) * 4 + q) * 1024), "v"(v) : "memory");
                    }
                }
    }
    unsigned *const counter = d.counters + (size_t)blockIdx.z * batch_tiles + tile;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // all S partials of this tile (bounded like stream-K's wait: a fault elsewhere must not hang the queue; reported)
        unsigned spins = 0;
        while (np::dev::coherent_load(counter) < d.S && spins < (1u << 26)) {
            __builtin_amdgcn_s_sleep(2);
            ++spins;
        }
        if (spins == (1u << 26) && d.error_word)
            __hip_atomic_fetch_or(d.error_word, np::kErrStreamK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned G = (unsigned)((i * TN + j) * 4 + q);
                if ((G & (d.S - 1)) != chunk || !writer) continue;   // uniform per wave
                const float *base = tile_ws + (size_t)G * 1024 + (size_t)tid * 4;
                const v4f own{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                v4f sum;
                if (d.S == 2) sum = fold_partials<2>(base, BM * BN, chunk, own);
                else if (d.S == 4) sum = fold_partials<4>(base, BM * BN, chunk, own);
                else if (SH::GROUPS >= 8 && d.S == 8) sum = fold_partials<(SH::GROUPS >= 8 ? 8 : 2)>(base, BM * BN, chunk, own);
                else sum = fold_partials<(SH::GROUPS >= 16 ? 16 : 2)>(base, BM * BN, chunk, own);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned r = 4 * q + e;
                    const unsigned row = m0 + row0 + i * 32 + (r & 3) + 8 * (r >> 2);
                    const unsigned col = n0 + col0 + j * 32;
                    if (!EDGE || (row < g.M && col < lim_n)) __builtin_nontemporal_store(sum[e], &C[(size_t)row * g.ldc + col]);
                }
            }
    // departures: the last of the S workgroups to leave puts the ticket back to zero
    __syncthreads();
    if (tid == 0 && __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2 * d.S - 1)
        np::dev::coherent_store(counter, 0u);
}

// ---- small products: one tile per workgroup, the four waves split K ("k-quartered") -------------------------------
// 768^3 on 64 x 64 tiles is 144 workgroups — 56 % of the CUs — and any finer cut of those tiles (64 x 32, K in two) puts two
// units on some CU, which is the same critical path.  What a product of that size needs is a tile whose COUNT is ~256:
// 48 x 48 for 768^3, 32 x 32 for 512^3 — and v_mfma_f32_32x32x2 cannot cut a 48 x 48 tile into four waves.  So here every
// wave computes the WHOLE tile, (16 TM) x (16 TN) as TM x TN blocks of v_mfma_f32_16x16x4 (the same flop rate), over a
// quarter of each 64-deep K-tile, and the four partial tiles meet in LDS at the end (summed in wave order: deterministic).
// NO LDS in the K loop: a wave's k-quarter is read by nobody else, and the v_mfma_f32_16x16x4 operand layouts ARE loadable
// straight from row-major memory — lane (r, kk) takes A[16 i + r][k0 + 4 kk .. + 3] as one global_load_dwordx4 (its A operand
// for four MFMA steps; which k a lane group holds is free: the MFMA sums over it) and B[k0 + 4 kk + t][16 j + r] as four
// global_load_dword (16 lanes = 64 contiguous bytes).  Operands for the K-tile two ahead are in flight in registers (three
// stages, the loop unrolled by three so that they are addressed statically; the compiler places the s_waitcnt).
// How it got here (profiles/r04/gemm_kq_forms_ab.log, gemm_kq_ablation.log, mfma_filler_cost.log, gemm_kq_direct_ab.log):
//   * first form: the slices staged through LDS by global_load_lds, shared [BM][64] / [64][BN] buffers, one barrier per K-tile:
//     768^3 12.3 us, a K-tile of a 48 x 48 tile 0.74 us where its 36 MFMAs take 0.51.  Wave-private LDS rings without any
//     barrier: -2 %.  The DMAs on two / four extra producer waves: +15 % / +-0.  Timing ablations: MFMAs + LDS reads 0.56 us,
//     + the DMAs 0.74, the DMAs alone 0.35.  The L2 was not it (83 % hits, per-K-tile time independent of the row pitch, one CU
//     pulls 105 GB/s through LDS-DMA when it does nothing else, against the 32 GB/s used).
//   * what it was: beside back-to-back MFMAs an LDS-DMA piece costs the issuing SIMD ~45 cycles, a ds_write_b128 ~20, a
//     ds_read_b128 ~11, a plain global load ~7 and a VALU instruction ~6 (tools/explore/mfma_filler_cost.hip) — so 6 DMAs + 9 LDS
//     reads per K-tile became 15 plain loads, and the per-lane 64-bit pointer arithmetic of a first direct version (~50 VALU
//     instructions per K-tile) became a uniform base advanced by scalar instructions plus constant 32-bit byte offsets:
//     0.74 -> 0.69 -> 0.61 us per K-tile; 768^3 12.2 -> 10.3 us (74 -> 88 TFLOP/s), 768 x 768 x 3072 39.6 -> 32.3 (112),
//     512^3 5.7 -> 4.9 (55), 704^3 10.7 -> 9.3, 832^3 17.0 (64 x 64 LDS-DMA tiles) -> 14.7.
// Operands below 4 GiB each, K >= 4.  VEC = float4-loadable rows (K % 4 == 0, N % 4 == 0, 16-byte aligned): the A operand is one
// dwordx4; otherwise (odd lengths, unaligned bases) four dwords — 9 more loads per K-tile of a 48 x 48 tile, ~5 %.
// Last K-tile of a K that is not a multiple of 64: the waves whose quarter lies beyond K sit it out, the one whose quarter
// ends inside zeroes the operands beyond K (loads beyond K re-read something in bounds); K-tiles beyond the last (the prefetch
// runs two ahead) re-fetch the last one.
// Tile (16 TM) x (16 TN), TM, TN = 2 .. 6; STAGES = operand stages in registers (3: the K-tile two ahead is in flight; 2 for the
// large tiles, whose K-tile lasts longer than a load takes, so that accumulators + stages stay within 256 registers and two
// workgroups fit a CU).  The final sum goes through LDS in chunks of CH blocks (32 KiB at a time for the large tiles).
constexpr int kq_regs(int tm, int tn, int stages) { return 4 * tm * tn + stages * 4 * (tm + tn) + 28; }
template <int TM, int TN, int STAGES, bool EDGE, bool VEC>
__global__ __launch_bounds__(256, kq_regs(TM, TN, STAGES) <= 256 ? 2 : 1) void sgemm_kq_kernel(GemmArgs g) {
    constexpr int BM = 16 * TM, BN = 16 * TN, BK = 64, NB = TM * TN;
    constexpr int CH = NB <= 16 ? NB : 8;   // blocks per pass of the final sum
    static_assert(STAGES == 2 || STAGES == 3, "two or three operand stages");
    __shared__ float red[4 * CH * 4 * 64];

    // K-chunks of one product (k_chunks != 0; a one-dimensional grid of tiles x chunks): the tiles of a chunk read the same slices
    // of A and B — once each from memory if they share an L2.  Workgroup b runs on XCD b % 8 (observed dispatch order), so the
    // units, numbered chunk-major (chunk * tiles + tile), are dealt to the XCDs in eight contiguous runs (the bijection of
    // tile_coords): an XCD works on whole chunks, every XCD on the same number of units +-1.  Deep-K products of a few tiles
    // stream their operands (100 x 100 x 100000: 80 MB under 2 GFLOP); tile-major, every slice crossed the fabric once per tile
    // that reads it: 64 x 32 tiles in 32 chunks 55.7 us, chunk-major 35.1 (profiles/r05/gemm_deep_k_sweep.log).
    unsigned bx = blockIdx.x, bz = blockIdx.z, nz = gridDim.z;
    if (g.k_chunks) {
        const unsigned tiles = g.tiles_m * g.tiles_n, units = tiles * g.k_chunks;
        const unsigned q = units / 8, r = units % 8, xcd = bx % 8, idx = bx / 8;
        const unsigned id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        bz = id / tiles;
        bx = id % tiles;
        nz = g.k_chunks;
    }
    if (g.K_last && bz + 1 == nz) g.K = g.K_last;   // uniform: split-K remainder chunk (>= 4: launch_kq)
    unsigned tile_m, tile_n;
    tile_coords(g, bx, tile_m, tile_n);
    const unsigned m0 = tile_m * BM, n0 = tile_n * BN;
    const float *A = g.A + (size_t)bz * g.stride_a;
    const float *B = g.B + (size_t)bz * g.stride_b;
    float *C = g.C + (size_t)bz * g.stride_c;
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lr = lane & 15, kk = lane >> 4;
    const unsigned nk = (g.K + BK - 1) / BK, kr = g.K - (nk - 1) * BK;   // kr: inner length of the last K-tile (1 .. 64; K >= 4)
    const unsigned kq = wave * 16 + kk * 4;                                // this lane's first k inside a K-tile
    // Last K-tile: a lane whose four k do not all lie inside K is pulled back by `pull` = kq + 4 - kr, so that what it loads ends
    // at K - 1 exactly (K >= 4: in bounds whatever the row, the last one included — reading a row's first chunk instead ran up to
    // 12 bytes past the end of A when kr < 4).  It then holds k = kr - 4 + t: nothing of its own if kq >= kr, its own k for
    // t >= pull if it STRADDLES K (only when K % 4 != 0); the rest belongs to its neighbours and is zeroed below, A and B alike.
    const unsigned pull = kq + 4 > kr ? kq + 4 - kr : 0u;

    // Sources: a UNIFORM base per operand, advanced one K-tile per load (scalar arithmetic), plus per-lane BYTE offsets that never
    // change (32-bit: the launcher checks that both operands are below 4 GiB) — the loads take the base-in-SGPRs form and cost
    // no vector arithmetic per K-tile (a first version with per-lane 64-bit pointers spent ~50 VALU instructions per K-tile on
    // them, ~6 cycles each beside MFMAs).  Last K-tile: a second offset set, with what lies beyond K pulled back to something
    // in bounds (zeroed / skipped below).
    const char *a_base = (const char *)A, *b_base = (const char *)B;
    unsigned a_off[TM], a_off_last[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        unsigned grow = m0 + 16 * i + lr;
        if (EDGE && grow >= g.M) grow = g.M - 1;
        a_off[i] = (grow * g.lda + kq) * 4u;
        a_off_last[i] = a_off[i] + 16u - pull * 4u;   // relative to a_base - 16: never negative (kr - 4 may be)
    }
    unsigned b_off[4], b_off_last[4], col_b[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        unsigned gcol = n0 + 16 * j + lr;
        if (EDGE && gcol >= g.N) gcol = g.N - 1;
        col_b[j] = EDGE ? gcol * 4u : (unsigned)(16 * j * 4);   // (not EDGE: n0 + lr sits in b_off, 64 j becomes the instruction's offset)
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        b_off[t] = ((kq + t) * g.ldb + (EDGE ? 0u : n0 + lr)) * 4u;
        b_off_last[t] = b_off[t] + (4u * g.ldb - pull * g.ldb) * 4u;   // rows kr - 4 + t, like A; relative to b_base - 4 rows
    }
    const size_t b_step = (size_t)BK * g.ldb * 4, b_back4 = (size_t)4 * g.ldb * 4;
    struct Frag {
        v4f a4[TM];
        float bv[TN][4];
    };
    unsigned issued = 0;
    auto load_tile = [&](Frag &f) {
        const bool last = issued + 1 >= nk;   // uniform
        const char *ab = last ? a_base - 16 : a_base, *bb = last ? b_base - b_back4 : b_base;   // (see a_off_last / b_off_last)
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const unsigned off = last ? a_off_last[i] : a_off[i];
            if constexpr (VEC) f.a4[i] = *(const v4f *)(ab + off);
            else   // rows of any alignment and length: four dword loads (the instruction's offset field carries 4 t)
#pragma unroll
                for (int t = 0; t < 4; ++t) f.a4[i][t] = *(const float *)(ab + off + 4 * t);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const unsigned off = last ? b_off_last[t] : b_off[t];
#pragma unroll
            for (int j = 0; j < TN; ++j) f.bv[j][t] = *(const float *)(bb + (off + col_b[j]));
        }
        if (!last) {
            a_base += BK * 4;
            b_base += b_step;
        }
        ++issued;
    };
    v4f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
    auto mfma_tile = [&](const Frag &f) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a4[i][t], f.bv[j][t], acc[i][j], 0, 0, 0);
    };
    constexpr int kMfma = 4 * NB, kLoads = TM + 4 * TN;
    auto step = [&](const Frag &now, Frag &into) {   // the loads of the K-tile two ahead behind the MFMAs of this one
        load_tile(into);
        mfma_tile(now);
#pragma unroll
        for (int q = 0; q < kLoads; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, kMfma / kLoads, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, kMfma, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto final_tile = [&](Frag &f) {
        if (wave * 16 >= kr) return;   // this wave's quarter lies beyond K
        if (wave * 16 + 16 > kr) {     // ... or ends inside: lanes beyond K hold re-read values, a straddling lane its neighbours' in t < pull
            const unsigned first_ok = kq >= kr ? 4u : pull;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool ok = (unsigned)t >= first_ok;
#pragma unroll
                for (int i = 0; i < TM; ++i) f.a4[i][t] = ok ? f.a4[i][t] : 0.0f;
#pragma unroll
                for (int j = 0; j < TN; ++j) f.bv[j][t] = ok ? f.bv[j][t] : 0.0f;
            }
        }
        mfma_tile(f);
    };

    if constexpr (STAGES == 3) {
        Frag s0, s1, s2;
        load_tile(s0);
        load_tile(s1);
        unsigned kt = 0;
        for (; kt + 3 < nk; kt += 3) {   // K-tiles kt, kt + 1, kt + 2 all have a successor
            step(s0, s2);
            step(s1, s0);
            step(s2, s1);
        }
        const unsigned left = nk - 1 - kt;   // 0 .. 2 tiles before the last one
        if (left == 0) final_tile(s0);
        else if (left == 1) {
            step(s0, s2);
            final_tile(s1);
        } else {
            step(s0, s2);
            step(s1, s0);
            final_tile(s2);
        }
    } else {
        Frag s0, s1;
        load_tile(s0);
        unsigned kt = 0;
        for (; kt + 2 < nk; kt += 2) {
            step(s0, s1);
            step(s1, s0);
        }
        if (nk - 1 - kt == 0) final_tile(s0);
        else {
            step(s0, s1);
            final_tile(s1);
        }
    }

    // the four partial tiles, CH blocks at a time: red[w][block][q][lane], lane-major; thread (q = wave, lane) sums a block in
    // wave order
#pragma unroll
    for (int c0 = 0; c0 < NB; c0 += CH) {
        if (c0) __syncthreads();   // (the previous pass has been read)
#pragma unroll
        for (int b = c0; b < (c0 + CH < NB ? c0 + CH : NB); ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q) red[((wave * CH + (b - c0)) * 4 + q) * 64 + lane] = acc[b / TN][b % TN][q];
        __syncthreads();
#pragma unroll
        for (int b = c0; b < (c0 + CH < NB ? c0 + CH : NB); ++b) {
            float sum = red[((0 * CH + (b - c0)) * 4 + wave) * 64 + lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) sum += red[((w * CH + (b - c0)) * 4 + wave) * 64 + lane];
            const unsigned row = m0 + (b / TN) * 16 + 4 * kk + wave, col = n0 + (b % TN) * 16 + lr;
            if (!EDGE || (row < g.M && col < g.N)) __builtin_nontemporal_store(sum, &C[(size_t)row * g.ldc + col]);
        }
    }
}

// Zero-padded copy of a row-major matrix: out (rows_out x ld_out, ld_out % 4 == 0, 16-byte aligned)
// = in (rows_in x cols_in, row stride ld_in) in the top-left corner, zeros elsewhere.
__global__ __launch_bounds__(256) void pad_copy_kernel(const float *__restrict__ in, unsigned ld_in,
                                                       unsigned rows_in, unsigned cols_in,
                                                       float *__restrict__ out, unsigned ld_out,
                                                       unsigned rows_out) {
    const unsigned c4n = ld_out / 4;
    const size_t total = (size_t)rows_out * c4n;
    for (size_t v = (size_t)blockIdx.x * blockDim.x + threadIdx.x; v < total; v += (size_t)gridDim.x * blockDim.x) {
        const unsigned r = (unsigned)(v / c4n), c = (unsigned)(v - (size_t)r * c4n) * 4;
        v4f x{0.0f, 0.0f, 0.0f, 0.0f};
        if (r < rows_in) {
            const float *src = in + (size_t)r * ld_in + c;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c + e < cols_in) x[e] = src[e];
        }
        *(v4f *)(out + (size_t)r * ld_out + c) = x;
    }
}

// y = A x, one wave per row (rows are contiguous: float4 loads, wave64 shuffle reduce).
__global__ __launch_bounds__(256) void sgemv_kernel(const float *__restrict__ A,
                                                    const float *__restrict__ x,
                                                    float *__restrict__ y, unsigned M, unsigned N,
                                                    int vec) {
    const unsigned lane = threadIdx.x & 63;
    const unsigned row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float *a = A + (size_t)row * N;
    float acc = 0.0f;
    if (vec) {
        const unsigned n4 = N / 4;
        for (unsigned v = lane; v < n4; v += 64) {
            const v4f av = *(const v4f *)(a + (size_t)v * 4);
            const v4f xv = *(const v4f *)(x + (size_t)v * 4);
            acc = fmaf(av[0], xv[0], acc);
            acc = fmaf(av[1], xv[1], acc);
            acc = fmaf(av[2], xv[2], acc);
            acc = fmaf(av[3], xv[3], acc);
        }
    } else {
        // rows that are not 16-byte aligned (odd N): dword-aligned float4 loads, two in flight per lane, instead of one
        // float per lane per load (4096 x 4097: 25 us -> see profiles/r03/gemm_fringe_probe.log), then the N % 4 tail
        typedef v4f v4f_u __attribute__((aligned(4)));
        const unsigned n4 = N / 4;
        float acc1 = 0.0f;
        unsigned v = lane;
        for (; v + 64 < n4; v += 128) {
            const v4f a0 = *(const v4f_u *)(a + (size_t)v * 4), x0 = *(const v4f_u *)(x + (size_t)v * 4);
            const v4f a1 = *(const v4f_u *)(a + (size_t)(v + 64) * 4), x1 = *(const v4f_u *)(x + (size_t)(v + 64) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc = fmaf(a0[e], x0[e], acc);
                acc1 = fmaf(a1[e], x1[e], acc1);
            }
        }
        for (; v < n4; v += 64) {
            const v4f a0 = *(const v4f_u *)(a + (size_t)v * 4), x0 = *(const v4f_u *)(x + (size_t)v * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = fmaf(a0[e], x0[e], acc);
        }
        for (unsigned k = n4 * 4 + lane; k < N; k += 64) acc = fmaf(a[k], x[k], acc);
        acc += acc1;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) y[row] = acc;
}

// Few long rows (an inner product is 1 x N): one wave per row leaves the chip idle (1 x 10^8 took
// 179 ms).  Each row is cut into chunks, one workgroup per (chunk, row) -> partial[row][chunk]; the
// caller folds the chunks with np_reduce_axis.  Dword-aligned float4 loads: rows start anywhere.
__global__ __launch_bounds__(256) void sgemv_chunks_kernel(const float *__restrict__ A,
                                                           const float *__restrict__ x,
                                                           float *__restrict__ partial, unsigned N,
                                                           unsigned chunk_len) {
    struct __attribute__((packed, aligned(4))) U4 { v4f v; };
    __shared__ float lds4[4];
    const unsigned row = blockIdx.y, chunk = blockIdx.x;
    const unsigned k0 = chunk * chunk_len;
    const unsigned len = (N - k0 < chunk_len) ? N - k0 : chunk_len;
    const float *a = A + (size_t)row * N + k0;
    const float *xx = x + k0;
    float acc0 = 0.0f, acc1 = 0.0f;
    const unsigned n4 = len / 4;
    unsigned v = threadIdx.x;
    for (; v + 256 < n4; v += 512) {
        const v4f a0 = ((const U4 *)(a + (size_t)v * 4))->v, x0 = ((const U4 *)(xx + (size_t)v * 4))->v;
        const v4f a1 = ((const U4 *)(a + (size_t)(v + 256) * 4))->v, x1 = ((const U4 *)(xx + (size_t)(v + 256) * 4))->v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            acc0 = fmaf(a0[k], x0[k], acc0);
            acc1 = fmaf(a1[k], x1[k], acc1);
        }
    }
    for (; v < n4; v += 256) {
        const v4f a0 = ((const U4 *)(a + (size_t)v * 4))->v, x0 = ((const U4 *)(xx + (size_t)v * 4))->v;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc0 = fmaf(a0[k], x0[k], acc0);
    }
    for (unsigned k = n4 * 4 + threadIdx.x; k < len; k += 256) acc0 = fmaf(a[k], xx[k], acc0);
    float acc = acc0 + acc1;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)row * gridDim.x + chunk] = (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
}

// The same for 2 .. 16 long rows (dot(matrix, vector) with a 10 x 10^7 matrix): a workgroup per CHUNK takes every row of its
// chunk, so the chunk of x is read once and stays in registers while the rows stream past it.  With a workgroup per (chunk,
// row) each row's workgroups re-read x from memory — 10 x 10^7 moved 800 MB for 440 MB of operands (0.48 of the roofline,
// BENCH r05 lease 4).  partial[row][chunk], folded by np_reduce_axis like the kernel above.
__global__ __launch_bounds__(256) void sgemv_fewrows_chunks_kernel(const float *__restrict__ A, const float *__restrict__ x,
                                                                   float *__restrict__ partial, unsigned M, unsigned N,
                                                                   unsigned chunk_len) {
    typedef v4f v4f_u __attribute__((aligned(4)));
    __shared__ float lds[16][4];
    const unsigned chunk = blockIdx.x;
    const unsigned k0 = chunk * chunk_len;
    const unsigned len = (N - k0 < chunk_len) ? N - k0 : chunk_len;
    const float *xx = x + k0;
    float acc[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) acc[m] = 0.0f;
    const unsigned n4 = len / 4;
    for (unsigned v = threadIdx.x; v < n4; v += 256) {
        const v4f x0 = *(const v4f_u *)(xx + (size_t)v * 4);
        v4f a[16];
#pragma unroll
        for (int m = 0; m < 16; ++m)
            if ((unsigned)m < M) a[m] = __builtin_nontemporal_load((const v4f_u *)(A + (size_t)m * N + k0 + (size_t)v * 4));
#pragma unroll
        for (int m = 0; m < 16; ++m)
            if ((unsigned)m < M) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[m] = fmaf(a[m][k], x0[k], acc[m]);
            }
    }
    for (unsigned k = n4 * 4 + threadIdx.x; k < len; k += 256) {
        const float xk = xx[k];
#pragma unroll
        for (int m = 0; m < 16; ++m)
            if ((unsigned)m < M) acc[m] = fmaf(A[(size_t)m * N + k0 + k], xk, acc[m]);
    }
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        if ((unsigned)m < M) {
            float r = acc[m];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) r += __shfl_down(r, off, 64);
            if ((threadIdx.x & 63) == 0) lds[m][threadIdx.x >> 6] = r;
        }
    }
    __syncthreads();
    if (threadIdx.x < M) partial[(size_t)threadIdx.x * gridDim.x + chunk] = (lds[threadIdx.x][0] + lds[threadIdx.x][1]) + (lds[threadIdx.x][2] + lds[threadIdx.x][3]);
}

// Many short rows (N <= 256, a 10^7 x 10 matrix): L lanes per row (L = 1 .. 32, a power of two) instead
// of a whole wave, which would use 10 of its 64 lanes.  Lanes of a group read the row interleaved and
// fold with xor-shuffles; neighbouring groups read neighbouring (contiguous) rows.
template <int L>
__global__ __launch_bounds__(256) void sgemv_short_rows_kernel(const float *__restrict__ A,
                                                               const float *__restrict__ x,
                                                               float *__restrict__ y, size_t M, unsigned N) {
    const size_t row = ((size_t)blockIdx.x * 256 + threadIdx.x) / L;
    const unsigned l = threadIdx.x % L;
    float acc = 0.0f;
    if (row < M) {
        const float *a = A + row * N;
        for (unsigned k = l; k < N; k += L) acc = fmaf(a[k], x[k], acc);
    }
#pragma unroll
    for (int off = L / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (l == 0 && row < M) y[row] = acc;
}


// Many short rows, staged: lanes of sgemv_short_rows_kernel read 4 bytes each at a row's stride, so a wave needs N
// load instructions for what is one contiguous span of memory (10^7 x 10: 3.5 TB/s).  Here a workgroup copies a
// contiguous slab of R rows into LDS with coalesced float4 loads — the matrix is read as the flat stream it is — and
// then every thread forms the inner products of its rows out of LDS (row pitch N | 1 words: an even N would put all
// lanes of a wave on a few banks).  x sits in LDS too.  Needs a 16-byte aligned A and R * N % 4 == 0.
__global__ __launch_bounds__(256) void sgemv_staged_rows_kernel(const float *__restrict__ A,
                                                                const float *__restrict__ x,
                                                                float *__restrict__ y, size_t M, unsigned N,
                                                                unsigned R, unsigned magic) {
    extern __shared__ __attribute__((aligned(16))) float slab[];   // R * pitch floats, then N floats of x
    const unsigned pitch = N | 1u;
    float *xs = slab + (size_t)R * pitch;
    const size_t row0 = (size_t)blockIdx.x * R;
    const unsigned rows = (unsigned)(M - row0 < R ? M - row0 : R);
    const unsigned total = rows * N;                 // floats of this slab
    const float *src = A + row0 * N;                 // 16-byte aligned: row0 * N is a multiple of 4
    for (unsigned c = threadIdx.x; c < N; c += 256) xs[c] = x[c];
    const unsigned nvec = total / 4;
    for (unsigned v = threadIdx.x; v < nvec; v += 256) {
        const v4f a = __builtin_nontemporal_load((const v4f *)(src + (size_t)v * 4));
        const unsigned e = v * 4;
        unsigned r = __umulhi(e, magic);             // e / N  (magic = ceil(2^32 / N), exact for e < 2^16)
        unsigned c = e - r * N;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            slab[r * pitch + c] = a[k];
            if (++c == N) { c = 0; ++r; }
        }
    }
    for (unsigned e = nvec * 4 + threadIdx.x; e < total; e += 256) {   // last slab of a matrix with M * N % 4 != 0
        const unsigned r = __umulhi(e, magic);
        slab[r * pitch + (e - r * N)] = src[e];
    }
    __syncthreads();
    for (unsigned r = threadIdx.x; r < rows; r += 256) {
        const float *a = slab + r * pitch;
        float acc0 = 0.0f, acc1 = 0.0f;
        unsigned c = 0;
        for (; c + 1 < N; c += 2) {
            acc0 = fmaf(a[c], xs[c], acc0);
            acc1 = fmaf(a[c + 1], xs[c + 1], acc1);
        }
        if (c < N) acc0 = fmaf(a[c], xs[c], acc0);
        y[row0 + r] = acc0 + acc1;
    }
}

// ---- thin products: C = A (M x K) . B (K x N) with N <= 32 --------------------------------------
// (points x 3) . (3 x 3), (samples x 784) . (784 x 10): these are HBM-bound reads of A — a 64 x 64
// MFMA tile would be 90 % padding (10^7 x 3 x 3 ran at 0.46 TB/s).  They are GEMVs with NV right-hand
// sides: L lanes per row of A (L = 1 .. 64, power of two; lane l takes k = l, l + L, ...), NV
// accumulators per lane, xor-shuffle fold, lane 0 of the group writes the N results of its row.  B is
// at most K x 32 floats and is read through the caches.  A is read exactly once, coalesced (a group
// reads its row contiguously, neighbouring groups read neighbouring rows).
template <int NV, int L>
__global__ __launch_bounds__(256) void sgemm_thin_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                         float *__restrict__ C, size_t M, unsigned N, unsigned K) {
    const size_t row = ((size_t)blockIdx.x * 256 + threadIdx.x) / L;
    const unsigned l = threadIdx.x % L;
    float acc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] = 0.0f;
    if (row < M) {
        const float *a = A + row * K;
        for (unsigned k = l; k < K; k += L) {
            const float x = a[k];
            const float *b = B + (size_t)k * N;
#pragma unroll
            for (int j = 0; j < NV; ++j)
                if (j < (int)N) acc[j] = fmaf(x, b[j], acc[j]);
        }
    }
#pragma unroll
    for (int off = L / 2; off > 0; off >>= 1)
#pragma unroll
        for (int j = 0; j < NV; ++j) acc[j] += __shfl_xor(acc[j], off, 64);
    if (l == 0 && row < M) {
        float *c = C + row * N;
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (j < (int)N) c[j] = acc[j];
    }
}

// N = 5 .. 32 columns ((samples x 784) . (784 x 10), a projection onto a few components): too many
// accumulators for the lane-group kernel above (it turns load-issue-bound), too few columns for a
// 64-wide tile.  One wave = ROWS rows of A x ROWS (padded) columns on the fp32 MFMA (32x32x2 for
// N > 16, 16x16x4 — half the matrix-core time per element of A — for N <= 16), A straight from
// global memory into registers: lane (i, q) owns the q-th 128 / KK bytes of row i's next 128-byte stretch and
// loads them with back-to-back float4 loads (so a line is consumed in one go — with one float4 per
// step the 32 lines per wave x 32 waves per CU fell out of the vector L1 before their fourth use),
// then feeds one component to each of W MFMAs; the MFMA sums over q, so every k of the 32-wide step is
// covered exactly once as long as B is indexed the same way.  B reaches the MFMA through LDS: the four
// waves of a block walk K in lockstep, 64 rows of B (zero-padded to ROWS columns) per chunk, double
// buffered so one barrier per chunk is enough (every wave reading B out of the caches itself moved
// as many bytes of B as of A).
template <int ROWS, int NCB = 1>   // NCB column blocks of ROWS columns (2 with ROWS = 32: N up to 64)
__global__ __launch_bounds__(256) void sgemm_thin_mfma_kernel(const float *__restrict__ A0, const float *__restrict__ B0,
                                                              float *__restrict__ C0, size_t M, unsigned N, unsigned Ktotal,
                                                              unsigned kc) {
    typedef v4f v4f_u __attribute__((aligned(4)));
    // split-K (gridDim.y > 1; a few hundred rows against a very long K — X^T . G of a dense layer's backward
    // pass): chunk blockIdx.y covers k in [k_begin, k_begin + K) and writes its own M x N partial
    const unsigned k_begin = blockIdx.y * kc;
    const unsigned K = Ktotal - k_begin < kc ? Ktotal - k_begin : kc;
    const float *A = A0 + k_begin;
    const float *B = B0 + (size_t)k_begin * N;
    float *C = C0 + (size_t)blockIdx.y * M * N;
    constexpr int KK = 64 / ROWS;          // k-slices the MFMA sums over (2 or 4)
    constexpr int W = 32 / KK;             // k values per lane per step (16 or 8)
    constexpr int NACC = ROWS * ROWS / 64; // 16 or 4
    constexpr int CH = 64;                 // k rows of B per LDS chunk = two steps
    constexpr int COLS = ROWS * NCB;
    constexpr int STR = ROWS == 32 ? COLS : 18;   // 16-wide rows: q and q + 1 land 16 banks apart
    constexpr int NB = CH * COLS / 256;    // staged elements per thread
    typedef float acc_t __attribute__((ext_vector_type(NACC)));
    __shared__ float Bs[2][CH * STR];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned i = lane % ROWS, q = lane / ROWS;
    const size_t r0 = ((size_t)blockIdx.x * 4 + wave) * ROWS;
    const size_t row = r0 + i;
    const bool row_ok = row < M;
    const float *a = A + (row_ok ? row : 0) * (size_t)Ktotal + W * q;
    acc_t acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < NACC; ++r) acc[cb][r] = 0.0f;
    auto fetch = [&](unsigned k0, float (&av)[W]) {
        const unsigned k = k0 + W * q;
#pragma unroll
        for (int v = 0; v < W / 4; ++v) {
            v4f x = {0, 0, 0, 0};
            const unsigned kv = k + 4 * v;
            if (row_ok && kv + 3 < K) x = *(const v4f_u *)(a + k0 + 4 * v);   // dword-aligned is enough
            else if (row_ok && kv < K) {                                       // the float4 that straddles K
                x[0] = a[k0 + 4 * v];
                if (kv + 1 < K) x[1] = a[k0 + 4 * v + 1];
                if (kv + 2 < K) x[2] = a[k0 + 4 * v + 2];
            }
            av[4 * v] = x[0]; av[4 * v + 1] = x[1]; av[4 * v + 2] = x[2]; av[4 * v + 3] = x[3];
        }
    };
    auto stage_load = [&](unsigned kc, float (&reg)[NB]) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const unsigned e = threadIdx.x + 256 * u, kk = e / COLS, j = e % COLS;
            reg[u] = (j < N && kc + kk < K) ? B[(size_t)(kc + kk) * N + j] : 0.0f;
        }
    };
    auto stage_store = [&](int buf, const float (&reg)[NB]) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const unsigned e = threadIdx.x + 256 * u, kk = e / COLS, j = e % COLS;
            Bs[buf][kk * STR + j] = reg[u];
        }
    };
    auto mma = [&](const float (&av)[W], const float *bs) {   // bs: this step's 32 rows of the chunk
#pragma unroll
        for (int t = 0; t < W; ++t) {
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const float bv = bs[(W * q + t) * STR + cb * ROWS + i];
                if constexpr (ROWS == 32) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv, acc[cb], 0, 0, 0);
                else acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t], bv, acc[cb], 0, 0, 0);
            }
        }
    };
    float a0[W], a1[W], breg[NB];
    stage_load(0, breg);
    fetch(0, a0);
    stage_store(0, breg);
    __syncthreads();
    int buf = 0;
    for (unsigned kc = 0; kc < K; kc += CH, buf ^= 1) {
        const bool more = kc + CH < K;                // uniform
        if (more) stage_load(kc + CH, breg);
        fetch(kc + 32, a1);                           // next step in flight under the MFMAs
        mma(a0, &Bs[buf][0]);
        fetch(kc + 64, a0);
        mma(a1, &Bs[buf][32 * STR]);
        if (more) stage_store(buf ^ 1, breg);
        __syncthreads();
    }
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const unsigned col = cb * ROWS + i;
        if (col >= N) continue;
#pragma unroll
        for (int r = 0; r < NACC; ++r) {
            const size_t orow = ROWS == 32 ? r0 + (r & 3) + 8 * (r >> 2) + 4 * q : r0 + 4 * q + r;
            if (orow < M) C[orow * N + col] = acc[cb][r];
        }
    }
}

// One or two row tiles (M <= 64) against a very long K — cluster sums H^T . X, Gram matrices of <= 64 features:
// B (K x N, one contiguous stream) is the bigger operand and every row tile reads it once anyway, so nothing is
// gained by sharing it through LDS; and with one or two row tiles a workgroup's waves should not be row tiles at
// all.  Here every WAVE owns a K-chunk of its own: A as in the kernel above (lane (i, q) takes 64 contiguous bytes
// of row i per 32-k step), B as plain dword loads — lanes j = 0..31 of a half-wave read 128 contiguous bytes of
// one row of B per MFMA — no LDS, no barrier.  Partials [chunk][M][N], folded by np_reduce_axis.
template <int NCB>
__global__ __launch_bounds__(256) void sgemm_fewrows_splitk_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                                   float *__restrict__ P, unsigned M, unsigned N, unsigned K,
                                                                   unsigned kc) {
    typedef v4f v4f_u __attribute__((aligned(4)));
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned i = lane & 31, q = lane >> 5;
    const unsigned chunk = blockIdx.y * 4 + wave;
    const unsigned kb = chunk * kc;                 // kc is a multiple of 32
    if (kb >= K) return;
    const unsigned ke = kb + kc < K ? kb + kc : K;
    const unsigned row = blockIdx.x * 32 + i;
    const bool row_ok = row < M;
    const float *a = A + (size_t)(row_ok ? row : 0) * K + 16 * q;
    bool col_ok[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) col_ok[cb] = cb * 32 + i < N;
    v16f acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.0f;
    auto fetch = [&](unsigned k0, float (&av)[16], float (&bv)[NCB][16]) {
        const unsigned k = k0 + 16 * q;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            v4f x = {0, 0, 0, 0};
            const unsigned kv = k + 4 * v;
            if (row_ok && kv + 3 < ke) x = *(const v4f_u *)(a + k0 + 4 * v);
            else if (row_ok && kv < ke) {
                x[0] = a[k0 + 4 * v];
                if (kv + 1 < ke) x[1] = a[k0 + 4 * v + 1];
                if (kv + 2 < ke) x[2] = a[k0 + 4 * v + 2];
            }
            av[4 * v] = x[0]; av[4 * v + 1] = x[1]; av[4 * v + 2] = x[2]; av[4 * v + 3] = x[3];
        }
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb)
                bv[cb][t] = (col_ok[cb] && k + t < ke) ? B[(size_t)(k + t) * N + cb * 32 + i] : 0.0f;
    };
    auto mma = [&](const float (&av)[16], const float (&bv)[NCB][16]) {
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv[cb][t], acc[cb], 0, 0, 0);
    };
    float a0[16], a1[16], b0[NCB][16], b1[NCB][16];
    fetch(kb, a0, b0);
    for (unsigned k0 = kb; k0 < ke; k0 += 64) {
        fetch(k0 + 32, a1, b1);                      // (past ke: every guard fails, zeros)
        mma(a0, b0);
        fetch(k0 + 64, a0, b0);
        mma(a1, b1);
    }
    float *out = P + (size_t)chunk * M * N;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        if (!col_ok[cb]) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned orow = blockIdx.x * 32 + (r & 3) + 8 * (r >> 2) + 4 * q;
            if (orow < M) out[(size_t)orow * N + cb * 32 + i] = acc[cb][r];
        }
    }
}

// The mirror image: M <= 16 rows of A, K <= 64, a very wide B ((3 x 3) . (3 x 10^7)): every thread owns four
// columns, reads the K float4s of B above them (coalesced rows) and keeps MV float4 accumulators; A is
// a handful of uniform scalars.  B is read once, C written once.
template <int MV>
__global__ __launch_bounds__(256) void sgemm_thin_left_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                              float *__restrict__ C, unsigned M, size_t N, unsigned K) {
    typedef v4f v4f_u __attribute__((aligned(4)));
    const size_t nvec = N / 4;
    for (size_t v = (size_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (size_t)gridDim.x * 256) {
        v4f acc[MV];
#pragma unroll
        for (int i = 0; i < MV; ++i) acc[i] = v4f{0, 0, 0, 0};
        for (unsigned k = 0; k < K; ++k) {
            const v4f b = __builtin_nontemporal_load((const v4f_u *)(B + (size_t)k * N + v * 4));
#pragma unroll
            for (int i = 0; i < MV; ++i) {
                if (i < (int)M) {
                    const float a = A[(size_t)i * K + k];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][e] = fmaf(a, b[e], acc[i][e]);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < MV; ++i)
            if (i < (int)M) __builtin_nontemporal_store(acc[i], (v4f_u *)(C + (size_t)i * N + v * 4));
    }
    // ragged tail: N % 4 columns
    if (blockIdx.x == 0) {
        const size_t j = nvec * 4 + threadIdx.x;
        if (j < N) {
            for (unsigned i = 0; i < M; ++i) {
                float acc = 0.0f;
                for (unsigned k = 0; k < K; ++k) acc = fmaf(A[(size_t)i * K + k], B[(size_t)k * N + j], acc);
                C[(size_t)i * N + j] = acc;
            }
        }
    }
}

// Few rows of A (M <= 8) against a large B (a vector . matrix product, the row edge of a peeled product): an
// HBM-bound read of B — a 64-row MFMA tile would be 90 % padding and the tiled kernels reach 1.8 TB/s (1 x 4097 x 4097:
// 37 us).  A wave owns 256 columns (one float4 per lane, dword-aligned: rows of B start anywhere) and every fourth row
// of a chunk of K; a lane keeps MV float4 accumulators; A's elements are wave-uniform (scalar loads).  The four waves
// of a workgroup take interleaved rows of the same chunk and are summed through LDS; (column block, chunk) workgroups
// write partial[chunk][M][N], folded by np_reduce_axis in chunk order (deterministic) — or C directly when one chunk
// covers K.  B is read once, 1 KiB per wave per row.
template <int MV>
__global__ __launch_bounds__(256) void sgemm_fewrows_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                            float *__restrict__ out, unsigned M, unsigned N, unsigned K,
                                                            unsigned chunk_len) {
    typedef v4f v4f_u __attribute__((aligned(4)));
    __shared__ v4f part[3][MV][64];
    const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned col = blockIdx.x * 256 + lane * 4;
    const unsigned k0 = blockIdx.y * chunk_len;
    const unsigned k1 = (K - k0 < chunk_len) ? K : k0 + chunk_len;
    const bool whole = col + 4 <= N;   // this lane's four columns are all inside
    v4f acc[MV];
#pragma unroll
    for (int i = 0; i < MV; ++i) acc[i] = v4f{0, 0, 0, 0};
    auto load_b = [&](unsigned k) {
        v4f b{0, 0, 0, 0};
        const float *src = B + (size_t)k * N + col;
        if (whole) b = __builtin_nontemporal_load((const v4f_u *)src);
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (col + e < N) b[e] = src[e];
        }
        return b;
    };
    auto rank1 = [&](unsigned k, const v4f &b) {
#pragma unroll
        for (int i = 0; i < MV; ++i)
            if (i < (int)M) {
                const float a = A[(size_t)i * K + k];   // k is wave-uniform
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][e] = fmaf(a, b[e], acc[i][e]);
            }
    };
    unsigned k = k0 + wave;
    for (; k + 12 < k1; k += 16) {   // four rows in flight
        const v4f b0 = load_b(k), b1 = load_b(k + 4), b2 = load_b(k + 8), b3 = load_b(k + 12);
        rank1(k, b0); rank1(k + 4, b1); rank1(k + 8, b2); rank1(k + 12, b3);
    }
    for (; k < k1; k += 4) rank1(k, load_b(k));
    if (wave) {
#pragma unroll
        for (int i = 0; i < MV; ++i) part[wave - 1][i][lane] = acc[i];
    }
    __syncthreads();
    if (wave == 0) {
        float *dst = out + (size_t)blockIdx.y * M * N;
#pragma unroll
        for (int i = 0; i < MV; ++i)
            if (i < (int)M) {
                const v4f sum = (acc[i] + part[0][i][lane]) + (part[1][i][lane] + part[2][i][lane]);
                if (whole) *(v4f_u *)(dst + (size_t)i * N + col) = sum;
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < N) dst[(size_t)i * N + col + e] = sum[e];
                }
            }
    }
}

// 9 ... 64 rows of A against a large B (a small batch against a weight matrix): still an HBM-bound read of B — the
// 256 x 128 tiles are three quarters padding there and the 64 x 64 ones reach 2.8 TB/s (16 ... 64 x 8192 x 8192: 93-96 us
// for 268 MB) — but too many rows for a lane to keep in registers (sgemm_fewrows_kernel).  Here the rows go to the
// matrix cores.  A workgroup owns 128 columns and a chunk of K; its four waves take every fourth 16-row tile of the
// chunk.  Per tile a lane fetches eight float4s of B — lane (jl = lane & 31, kh = lane >> 5) reads rows kt + 8 kh + s,
// s = 0..7, at columns n0 + 4 jl .. + 3: a wave covers two rows x 512 bytes per load — and component c of those float4s
// is the B operand of MFMA (s, c): v_mfma_f32_32x32x2_f32 pairs k = kt + s (kh = 0) with k = kt + 8 + s (kh = 1), and
// MFMA c produces the output columns n0 + 4 j + c — the four MFMAs of a step interleave to 128 consecutive columns, so a
// lane stores float4s.  A's 16-column tile (<= 64 rows: 4 KiB) is staged by the wave itself in its own LDS slice (rows
// padded to 20 floats) and read back as the A operands: lane (i, kh) takes A[rb * 32 + i][kt + 8 kh .. + 7].  Rows >= M
// and k >= the chunk's end are zeros.  The four waves' accumulators are summed through LDS block by block; (column
// block, chunk) partials [chunk][M][N] are folded by np_reduce_axis in chunk order (deterministic).
template <int RB>   // 32-row blocks: 1 (M <= 32) or 2 (M <= 64)
__global__ __launch_bounds__(256, 2) void sgemm_skinny_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                              float *__restrict__ out, unsigned M, unsigned N, unsigned K,
                                                              unsigned chunk_len) {
    typedef v4f v4f_u __attribute__((aligned(4)));
    constexpr int PITCH = 20;                                   // floats per staged row of A (16 + 4: 16-byte aligned, off the bank period)
    __shared__ __attribute__((aligned(16))) float lds[4 * RB * 32 * PITCH > 3 * 64 * 16 ? 4 * RB * 32 * PITCH : 3 * 64 * 16];
    const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned jl = lane & 31, kh = lane >> 5;
    const unsigned n0 = blockIdx.x * 128, col = n0 + 4 * jl;
    const unsigned k0 = blockIdx.y * chunk_len;
    const unsigned k1 = (K - k0 < chunk_len) ? K : k0 + chunk_len;
    const bool whole = col + 4 <= N;
    float *as = lds + wave * (RB * 32 * PITCH);
    v16f acc[RB][4];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][c][r] = 0.0f;

    // (issuing the next tile's loads before this tile's MFMAs costs 70 more registers and half the resident waves:
    // 16 x 8192 x 8192 64 -> 83 us; the other waves of the SIMD are the prefetch)
    for (unsigned kt = k0 + wave * 16; kt < k1; kt += 64) {
        // B: eight rows per lane group
        v4f b[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const unsigned k = kt + 8 * kh + s;
            b[s] = v4f{0, 0, 0, 0};
            if (k < k1) {
                const float *src = B + (size_t)k * N + col;
                if (whole) b[s] = __builtin_nontemporal_load((const v4f_u *)src);
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < N) b[s][e] = src[e];
                }
            }
        }
        // A: the wave stages rows 0 .. RB * 32 - 1 of its 16 k-columns in its own LDS slice (lane: row lane / 4 + 16 p,
        // k-columns 4 (lane % 4) .. + 3) and reads the operands back
#pragma unroll
        for (int p = 0; p < RB * 2; ++p) {
            const unsigned row = (lane >> 2) + 16 * p, kq = kt + 4 * (lane & 3);
            v4f x{0, 0, 0, 0};
            if (row < M) {
                const float *src = A + (size_t)row * K + kq;
                if (kq + 4 <= k1) x = *(const v4f_u *)src;
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (kq + e < k1) x[e] = src[e];
                }
            }
            *(v4f *)(as + row * PITCH + 4 * (lane & 3)) = x;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // a wave's LDS operations complete in order: its reads below see its writes
        __builtin_amdgcn_wave_barrier();
        float a[RB][8];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const v4f lo = *(const v4f *)(as + (rb * 32 + jl) * PITCH + 8 * kh), hi = *(const v4f *)(as + (rb * 32 + jl) * PITCH + 8 * kh + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[rb][e] = lo[e];
                a[rb][4 + e] = hi[e];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // the operands are in registers before the next tile is staged over them
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[rb][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[rb][s], b[s][c], acc[rb][c], 0, 0, 0);
    }

    // the four waves' sums, one (row block, column phase) block of 16 floats per lane at a time: waves 1..3 -> LDS -> wave 0
    __syncthreads();
    float *red = lds;   // [3][64 lanes][16]
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (wave) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(v4f *)(red + ((wave - 1) * 64 + lane) * 16 + 4 * q) =
                        v4f{acc[rb][c][4 * q], acc[rb][c][4 * q + 1], acc[rb][c][4 * q + 2], acc[rb][c][4 * q + 3]};
            }
            __syncthreads();
            if (!wave) {
#pragma unroll
                for (int w = 0; w < 3; ++w)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const v4f t = *(const v4f *)(red + (w * 64 + lane) * 16 + 4 * q);
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[rb][c][4 * q + e] += t[e];
                    }
            }
            __syncthreads();
        }
    if (wave) return;
    float *dst = out + (size_t)blockIdx.y * M * N;
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (row < M) {
                const v4f v{acc[rb][0][r], acc[rb][1][r], acc[rb][2][r], acc[rb][3][r]};
                float *d = dst + (size_t)row * N + col;
                if (whole) *(v4f_u *)d = v;
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < N) d[e] = v[e];
                }
            }
        }
}

// Few rows of A, a long K, N <= 32 (X^T X of a 10^7 x 3 array): (chunk, row) workgroups as in
// sgemv_chunks_kernel, with NV accumulators; partial[row][chunk][N], folded by np_reduce_axis.
template <int NV>
__global__ __launch_bounds__(256) void sgemm_thin_chunks_kernel(const float *__restrict__ A, const float *__restrict__ B,
                                                                float *__restrict__ partial, unsigned N, unsigned K,
                                                                unsigned chunk_len) {
    __shared__ float lds[4][NV];
    const unsigned row = blockIdx.y, chunk = blockIdx.x;
    const unsigned k0 = chunk * chunk_len;
    const unsigned len = (K - k0 < chunk_len) ? K - k0 : chunk_len;
    const float *a = A + (size_t)row * K + k0;
    const float *b = B + (size_t)k0 * N;
    float acc[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) acc[j] = 0.0f;
    for (unsigned k = threadIdx.x; k < len; k += 256) {
        const float x = a[k];
        const float *bk = b + (size_t)k * N;
#pragma unroll
        for (int j = 0; j < NV; ++j)
            if (j < (int)N) acc[j] = fmaf(x, bk[j], acc[j]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int j = 0; j < NV; ++j) acc[j] += __shfl_xor(acc[j], off, 64);
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NV; ++j) lds[wave][j] = acc[j];
    }
    __syncthreads();
    if (threadIdx.x < N)
        partial[((size_t)row * gridDim.x + chunk) * N + threadIdx.x] =
            (lds[0][threadIdx.x] + lds[1][threadIdx.x]) + (lds[2][threadIdx.x] + lds[3][threadIdx.x]);
}

int g_variant = 0;
// sgemm_dma_kernel<.., PRIO>: K-tiles per priority phase; np_sgemm_set_variant(-(100 + p)) sets it (p = 0: off).
// 16: 145.4 -> 146.4 TFLOP/s at 4096^3 (tools/gemm_prio_ab.py, profiles/r02/gemm_prio_ab.log; 2...64 all within 0.3 %)
unsigned g_prio_period = 16;
// ... from this K up.  Below it the alternation is OFF (round 5): with few K-tiles the epilogue — 128 KiB of C per workgroup, stored
// by all 512 of them at once — is a fifth of the kernel, and two workgroups of a CU that do NOT finish together overlap one's stores
// with the other's MFMAs.  Same box, alternation on -> off (profiles/r05/gemm_thin_k_ab.log): 4096 x 4096 x 128 94.6 -> 108.9 TFLOP/s,
// x 256 119.5 -> 124.9, 8192 x 8192 x 256 124.8 -> 133.5, K = 512 equal, K = 1024 139.5 -> 141.7, 4096^3 144.9 -> 145.3 (noise).
// np_sgemm_set_variant(-(100 + p)) with p != 16 applies period p at every K (A/B); p = 16 restores this default.
unsigned g_prio_min_k = 2048;
unsigned long long *g_probe = nullptr;

inline bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

// What np::sgemm_batched_with_progress asks of the ONE launch it makes (host-side plumbing).  Per host thread: an np_sgemm
// on another thread while the communicator's thread is inside np_sgemm_strided_batched_allgather must not pick up that
// call's progress counters (it would release transfers early and switch its own C to memory-side stores).
struct ProgressRequest {
    unsigned *counters = nullptr;
    unsigned base = 0, extra = 0;
    unsigned tiles_per_matrix = 0;   // out: written by the launcher that took the request
    unsigned launches = 0;           // out: kernel launches made while the request was active
};
thread_local ProgressRequest g_progress;
inline void note_progress_launch(const GemmArgs &g) {
    if (!g.progress) return;
    g_progress.tiles_per_matrix = g.tiles_m * g.tiles_n;
    ++g_progress.launches;
}

template <int BM, int BN, int BK, int MINW>
int launch_sgemm_tile(GemmArgs g, unsigned batch, bool vec) {
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    const bool edge = (g.M % BM) || (g.N % BN) || (g.K % BK) || (g.K_last % BK);
    const dim3 grid(g.tiles_m * g.tiles_n, 1, batch);
    hipStream_t s = np::stream();
    if (vec && !edge)
        sgemm_kernel<BM, BN, BK, MINW, true, false><<<grid, 256, 0, s>>>(g);
    else if (vec)
        sgemm_kernel<BM, BN, BK, MINW, true, true><<<grid, 256, 0, s>>>(g);
    else
        sgemm_kernel<BM, BN, BK, MINW, false, true><<<grid, 256, 0, s>>>(g);
    NP_LAUNCH_CHECK("sgemm_kernel");
    return NP_OK;
}

template <int BM, int BN, int MODE>
int launch_sgemm_pipe(GemmArgs g, unsigned batch, bool vec) {
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    const bool edge = (g.M % BM) || (g.N % BN) || (g.K % 16) || (g.K_last % 16);
    const dim3 grid(g.tiles_m * g.tiles_n, 1, batch);
    hipStream_t s = np::stream();
    if (vec && !edge)
        sgemm_pipe_kernel<BM, BN, true, false, MODE><<<grid, 256, 0, s>>>(g);
    else if (vec)
        sgemm_pipe_kernel<BM, BN, true, true, MODE><<<grid, 256, 0, s>>>(g);
    else
        sgemm_pipe_kernel<BM, BN, false, true, MODE><<<grid, 256, 0, s>>>(g);
    NP_LAUNCH_CHECK("sgemm_pipe_kernel");
    return NP_OK;
}

// ---- default kernel choice -------------------------------------------------------------------
// Three tile configurations, efficiencies measured at 4096^3 on MI355X (profiles/r01/gemm_ab.log):
//   0: sgemm_dma_kernel 256x128 (K % 16 == 0, float4 rows)      0.93   (145 TFLOP/s)
//   1: sgemm_kernel     128x128                               0.85   (132-135)
//   2: sgemm_kernel      64x64                                0.71   (110)
// eff = MFMA efficiency with several workgroups resident per CU (>= 2 waves of tiles: one
// workgroup's barriers and LDS traffic hide under another's MFMAs), eff1 = with a single workgroup
// per CU (one wave of tiles or less): 2000^3 on 128x128 tiles is exactly 256 tiles and runs at 0.64,
// not 0.85 (profiles/r01/gemm_plan_sweep.log).
//   3-5: sgemm_dmas_kernel 128x128 / 128x64 / 64x64 (round 4): the LDS-DMA pipeline on smaller tiles, whole K or K split
//        in-launch.  Calibrated against the register-staged kernels ON THE SAME BOX (profiles/r04/gemm_mid_sweep_forced.log):
//        2048^3 on 256 tiles of 128x128 149 us (the model's 0.64 said 171) -> 0.83; 1024^3 on 256 tiles of 64x64 22.4 us
//        against the old kernel's 31.1 -> 0.74; several waves: 0.87 / 0.82 / 0.80.  Operands of any alignment (1001 x 1003
//        x 1002: 26.2 us against 40.6 on the padded / register-staged forms), so these take every product the planner has.
//   6 .. 8 sgemm_kq_kernel: 48 x 48, 32 x 32 and 64 x 64 tiles, the four waves of a workgroup split K (v_mfma_f32_16x16x4, operands straight
//        from memory); whole K only, float4-loadable operands.  Two 48 x 48 workgroups fit a CU, four 32 x 32 ones; co-resident
//        ones share the matrix pipe (time = rounds x K-tiles x the K-tile time) but pay what lies outside the K loop once.
//        Fitted on profiles/r04/gemm_kq_direct_ab.log: a 64-deep K-tile of a 48 x 48 tile 0.61 us (0.79 of the pipe's rate), of a
//        32 x 32 tile 0.30-0.31 us; outside the loop 3.0 / 2.5 us: 768^3 10.3 us (model 10.3), 768 x 768 x 3072 32.3 (32.3), 512^3
//        on 32 x 32 tiles 4.9 (4.9), 704^3 9.3 (9.3), 832^3 14.7 (14.6), 1024^3 on 48 x 48 tiles 22.4 (22.5).  64 x 64 (247 registers,
//        still two workgroups per CU): a K-tile 1.02-1.05 us alone, ~1.0 with a co-resident one — 1024^3 19.8 us (19.5; the LDS-DMA
//        64 x 64 tiles: 20.6), 256 x 4096 x 4096 67.9 (68.9; 73-76: 126 TFLOP/s, the 0.80 of peak VERDICT r03 asked of this shape),
//        2048^3 127.6 (131.5), 1536^3 74.8 (75; the 48 x 48 tiles 58.8).
struct TileCfg { unsigned bm, bn; double eff, eff1, extra; };
struct KqShape { int tm, tn, stages; };
// (measured and not kept — profiles/r04/gemm_kq_family.log: 80 x 64, 80 x 80, 96 x 64, 96 x 80 spill registers at two workgroups
// per CU and run 1.5-3x slower than the model; 96 x 96 on one workgroup per CU never beat 48 x 48 on three rounds)
constexpr KqShape kKqShapes[] = {{3, 3, 3}, {2, 2, 3}, {4, 4, 3}, {3, 2, 3}, {4, 2, 3}, {4, 3, 3}, {5, 3, 3}};
constexpr int kKqShapeCount = (int)(sizeof(kKqShapes) / sizeof(kKqShapes[0]));
//        The whole family (TM, TN = 2 .. 6) follows one cost model, which is what the table below is generated from: a K-tile takes
//        128 TM TN cycles of MFMA + 7 per load (TM + 4 TN of them) + ~90 (scalar bookkeeping, the waits) at the ~2.25 GHz these
//        kernels run at, i.e. eff = 0.9375 x 128 TM TN / that; outside the loop 1.0 us + 0.06 us per block (the final sum), once.
constexpr int kFirstMidCfg = 3, kFirstKqCfg = 6, kCfgCount = kFirstKqCfg + kKqShapeCount;
struct CfgTable { TileCfg c[kCfgCount]; };
constexpr CfgTable make_cfg_table() {
    CfgTable t{};
    t.c[0] = {256, 128, 0.93, 0.89, 0};
    t.c[1] = {128, 128, 0.85, 0.64, 0};
    t.c[2] = {64, 64, 0.71, 0.52, 0};
    t.c[3] = {128, 128, 0.87, 0.83, 0};
    t.c[4] = {128, 64, 0.82, 0.80, 0};
    t.c[5] = {64, 64, 0.80, 0.74, 0};
    for (int s = 0; s < kKqShapeCount; ++s) {
        const double tm = kKqShapes[s].tm, tn = kKqShapes[s].tn, mfma = 128.0 * tm * tn;
        const double eff = 0.9375 * mfma / (mfma + 7.0 * (tm + 4.0 * tn) + 90.0);
        t.c[kFirstKqCfg + s] = {(unsigned)(16 * kKqShapes[s].tm), (unsigned)(16 * kKqShapes[s].tn), eff, eff, 1.0e-6 + 0.06e-6 * tm * tn};
    }
    return t;
}
constexpr CfgTable kCfgTable = make_cfg_table();
constexpr const TileCfg (&kCfg)[kCfgCount] = kCfgTable.c;
int g_kq_tiles = 1;      // np_sgemm_set_variant(-20) = 0: plans without sgemm_kq_kernel, (-21): back
int g_force_kq_split_shape = 0, g_force_kq_split_S = 0;   // np_sgemm_set_variant(-(30000 + 1000 * shape + S)): single products of K >= 512 as S K-chunks on that k-quartered shape (A/B); -30000: off
// The two K-chunked thin forms (17 .. 2047 rows, N <= 64, K >= 16384) were sized on K ~ 2e6: chunks of >= 1024 / 2048 inner elements.
// At K ~ 1e5 that is ~50 workgroups on 256 CUs — 64 x 64 x 100000: 89 us for 51 MB — so a launch of fewer than q / 4 workgroups
// per CU is left to the planner (its K-chunked k-quartered tiles: 16.5 us).  np_sgemm_set_variant(-(40 + q)), q = 0: never.
// Measured at q = 0, 1, 2, 4, 8, 59 on 16 shapes (profiles/r05/gemm_thin_fill_ab.log): the planner is level or ahead at EVERY fill —
// 64 x 64 x 400000 (196 workgroups) 94 -> 40 us, 32 x 32 x 1000000 (244) 62 -> 46, 32 x 64 x 2000000 (489) 190 -> 149, 1500 x 32 x 20000
// 89 -> 25 — so q = 59 (the largest the variant encodes): these two forms now only run launches of ~15 workgroups per CU and more
// (K beyond ~1.5e7).
int g_thin_underfilled_to_planner = 59;
inline bool thin_to_planner(size_t workgroups) { return workgroups * 4 < (size_t)g_thin_underfilled_to_planner * (size_t)np::num_cus(); }
int g_kq_chunk_major = 1;   // np_sgemm_set_variant(-25) = 0: K-chunked k-quartered launches tile-major (blockIdx.z = chunk), (-26): chunk-major over the XCDs (default)
int g_kq_splitk = 1;     // np_sgemm_set_variant(-22) = 0: no K-chunked plans on the k-quartered tiles (deep-K products as before round 5: the register-staged tiles), (-23): back
int g_mid_tiles = 1;   // np_sgemm_set_variant(-14) = 0: plans as before round 4 (no sgemm_dmas_kernel), (-15): back
int g_mid_waves = 1;     // np_sgemm_set_variant(-18) = 0: ragged whole-K 64 x 64 products on four waves like the aligned ones, (-19): on eight (default; see DmasShape5)
int g_mid_swizzle = 1;   // np_sgemm_set_variant(-16) = 0: sgemm_dmas_kernel walks its tiles row-major, (-17): XCD-aware bands (default)
constexpr unsigned kDmasBM[6] = {128, 128, 64, 128, 128, 64}, kDmasBN[6] = {128, 64, 64, 128, 64, 64}, kDmasMaxS[6] = {16, 8, 4, 16, 8, 4},
                   kDmasBK[6] = {16, 32, 32, 32, 32, 32};   // 0 .. 2: the shapes of cfg 3 .. 5; 3: 128 x 128 with the deeper K-tile, 4 / 5: shapes 1 / 2 on 8 waves (A/B)

int launch_dmas(int shape, GemmArgs g, unsigned batch, unsigned S);
int launch_kq(int shape, GemmArgs g, unsigned batch, bool vec);

int launch_cfg(int cfg, GemmArgs g, unsigned batch, bool vec) {
    if (cfg >= kFirstKqCfg) {
        const int rc = launch_kq(cfg - kFirstKqCfg, g, batch, vec);
        if (rc != 1) return rc;
        cfg = kFirstMidCfg + 2;   // (does not apply after all: the 64 x 64 LDS-DMA tiles)
    }
    if (cfg >= kFirstMidCfg) {
        const int rc = launch_dmas(cfg - kFirstMidCfg, g, batch, 1);
        if (rc != 1) return rc;
        cfg = cfg == kFirstMidCfg ? 1 : 2;   // (does not apply after all: the register-staged kernel of that size)
    }
    if (cfg == 0) {
        g.tiles_m = (g.M + 255) / 256;
        g.tiles_n = (g.N + 127) / 128;
        const dim3 grid(g.tiles_m * g.tiles_n, 1, batch);
        // XCD-aware tile order, bands of 4 tile rows: time-neutral at 4096^3 (the matrices sit in the
        // 256 MiB Infinity Cache) but the workgroups of one XCD share panels in its L2, L2->fabric
        // reads 944 -> 691 MB per launch (tools/gemm_swizzle_pmc.py, profiles/r01/gemm_swizzle_pmc.log)
        if (g_variant == 0 && g.tiles_m >= 8) g.swizzle = 4;
        const bool edge = g.M % 256 || g.N % 128 || g.n_store;
        const bool ktail = g.K % 16 || g.K_last % 16 || g.N % 4;   // N % 4: B's last row needs the tail tile's fix-up
        if (edge && ktail)
            sgemm_dma_kernel<true, true><<<grid, 256, 0, np::stream()>>>(g);
        else if (edge)
            sgemm_dma_kernel<true, false><<<grid, 256, 0, np::stream()>>>(g);
        else if (ktail)
            sgemm_dma_kernel<false, true><<<grid, 256, 0, np::stream()>>>(g);
        else if (g_prio_period && g.K >= g_prio_min_k) {
            g.prio_period = g_prio_period;
            sgemm_dma_kernel<false, false, true><<<grid, 256, 0, np::stream()>>>(g);
        } else
            sgemm_dma_kernel<false, false><<<grid, 256, 0, np::stream()>>>(g);
        NP_LAUNCH_CHECK("sgemm_dma_kernel");
        note_progress_launch(g);
        return NP_OK;
    }
    if (cfg == 1) return launch_sgemm_tile<128, 128, 16, 4>(g, batch, vec);
    return launch_sgemm_tile<64, 64, 16, 4>(g, batch, vec);
}

bool g_force_pad = false;   // np_sgemm_set_variant(-3): always take launch_padded when it applies (tests)
int g_dma_any_alignment = 1;   // np_sgemm_set_variant(-6) = 0: the LDS-DMA kernel only for float4-loadable operands, as before round 3; (-8) = 2: for unaligned ones of any size (A/B); (-7) = 1: default

int g_streamk = 0;      // np_sgemm_set_variant(-4): stream-K wherever the kernel can run it, (-5): never, (-2): back to the model

// Shapes the LDS-DMA kernel (and its stream-K form) takes: a row holds at least one whole 4-float chunk.  `vec` =
// every row of both operands is float4-loadable — no longer a requirement (sgemm_dma_kernel's comment), but small
// products whose rows are not are left to the forms they took before (one matrix: padded copies; batches: the
// register-staged kernels): measured faster there (profiles/r03/gemm_unaligned.log: 1001^3 38 us against 42, 64 x 513^3
// 238 against 263; from 1537^3 / 16 x 1001^3 up the LDS-DMA kernel on the operands as they are wins by 3-20 %).
inline bool dma_takes(size_t M, size_t N, size_t K, size_t batch, bool vec) {
    if (N < 4 || K < 4) return false;
    if (vec) return true;
    if (!g_dma_any_alignment) return false;
    const double flop = 2.0 * (double)M * (double)N * (double)K;
    return g_dma_any_alignment == 2 || g_streamk > 0 || flop >= (batch == 1 ? 4e9 : 1e9);
}
bool g_splitk = true;   // np_sgemm_set_variant(-1) turns the K-splitting plans off (A/B in tools/)

// A plan = tile configuration + how many trailing tile-ROWS of C are computed split-K.
//   tail_rows == 0        : one launch, every workgroup walks the whole K (the classic grid)
//   0 < tail_rows < all   : the leading rows fill whole waves of the machine; the rows that would
//                           have formed a mostly-empty last wave are cut into S chunks of K so they
//                           fill the CUs too (3072^3: 288 tiles on 256 CUs = 2 waves, the second
//                           12 % full -> 240 tiles + 48 tiles x 5 chunks)
//   tail_rows == all      : the whole product is split-K (few tiles, long K: X^T X of a tall X)
// Split parts run as the BATCH dimension of the same kernels (chunk s reads A[:, s*Kc:(s+1)*Kc]
// with row stride K and B[s*Kc:(s+1)*Kc, :], writes its partial to W[s]); one deterministic
// np_reduce_axis(sum over s) pass writes C.  No new device code, no atomics: the result never
// depends on scheduling.  Times are modelled in seconds: a work unit of k inner steps costs
// 2*bm*bn*k / (eff * peak per CU) + a fixed prologue/epilogue, the reduce costs its HBM traffic.
struct Plan { int cfg; unsigned tail_rows, S; size_t Kc; double t; };

// t_alt: the modelled time of an alternative the caller holds (stream-K): the mid-size tiles must clear it as well
thread_local int g_plan_cus = 0;   // (per host thread, like g_progress) np_sgemm_debug_plan: plan for this many CUs instead of the current device's (the planner is host arithmetic: testable without a device)
inline int plan_cus() { return g_plan_cus ? g_plan_cus : np::num_cus(); }

Plan plan_sgemm(size_t M, size_t N, size_t K, size_t batch, bool dma_ok, bool only_dma = false, bool vec = true, bool splitk = true,
                double t_alt = 1e300, Plan *mid_out = nullptr, Plan *other_out = nullptr) {
    const double cus = (double)plan_cus();
    const double cu_flops = 157.3e12 / 256.0, unit_fixed = 1.5e-6, launch = 3e-6, hbm = 4e12;
    Plan best{2, 0, 1, K, 1e300};
    Plan best_other{2, 0, 1, K, 1e300};   // the best plan WITHOUT the mid-size tiles (see the end of the function)
    Plan best_one{2, 0, 1, K, 1e300};     // the best whole-K mid-size plan whose tiles fit the machine in ONE round
    Plan best_kq_split{2, 0, 1, K, 1e300};   // the best K-chunked plan on the k-quartered tiles
    const bool mid_ok = g_mid_tiles && !only_dma && N >= 4 && K >= 4 && !g_progress.counters;
    for (int c = 0; c < kCfgCount; ++c) {
        if (c == 0 && !dma_ok) continue;
        if (c != 0 && only_dma) continue;
        if (c >= kFirstMidCfg && !mid_ok) continue;
        if (c >= kFirstKqCfg && (!g_kq_tiles || K < 4 || !splitk)) continue;   // (!splitk: C is a window of a wider matrix — fine for the kernel, but keep the peeled forms on the plans they were measured with)
        if (c == kFirstMidCfg) {   // cfg 0 .. 2 are done: what follows competes with their best
            best_other = best;
            best = Plan{2, 0, 1, K, 1e300};
        }
        const TileCfg &T = kCfg[c];
        const size_t tm = (M + T.bm - 1) / T.bm, tn = (N + T.bn - 1) / T.bn;
        // time of `units` work units of k inner steps each
        auto span = [&](double units, size_t k) {
            if (units <= 0) return 0.0;
            const double waves = units / cus;
            const double blend = waves <= 1.0 ? 0.0 : waves >= 2.0 ? 1.0 : waves - 1.0;
            // the register-staged kernels lose ~1/4 when rows are not float4-loadable (4097^3: 97 vs 132), the LDS-DMA
            // kernel 5-9 % (16-byte fetches that straddle cache lines: profiles/r03/gemm_unaligned.log)
            const double eff = (T.eff1 + (T.eff - T.eff1) * blend) * (vec ? 1.0 : c == 0 ? 0.93 : c >= kFirstKqCfg ? 0.95 : c >= kFirstMidCfg ? 0.97 : 0.78);
            if (c >= kFirstKqCfg) return ceil(waves) * (2.0 * T.bm * T.bn * (double)k / (eff * cu_flops)) + unit_fixed;   // (co-resident rounds)
            return ceil(waves) * (2.0 * T.bm * T.bn * (double)k / (eff * cu_flops) + unit_fixed);
        };
        const double whole = span((double)(tm * tn * batch), K) + T.extra;
        if (whole < best.t) best = Plan{c, 0, 1, K, whole};
        if (c >= kFirstMidCfg && (double)(tm * tn * batch) <= cus && whole < best_one.t) best_one = Plan{c, 0, 1, K, whole};
        if (c >= kFirstKqCfg) {
            // ... or, for a handful of tiles under a very deep K (100 x 100 x 100000: four 64 x 64 tiles), the whole product cut
            // along K into chunks that run as the batch dimension of one launch, folded by one np_reduce_axis like the
            // register-staged forms below (tail_rows == all tile rows says so) — whose 64 x 64 tiles run at 0.52 of a CU's rate
            // where these run at 0.84.  Chunks of whole 64-deep K-tiles, at least four of them.
            if (!g_splitk || !g_kq_splitk || !splitk || batch != 1 || K < 2048 || tm * tn > (size_t)cus / 2) continue;
            for (size_t S = 2; S <= 256; S = S < 8 ? S + 1 : S * 2) {
                const size_t Kc = ((K + S - 1) / S + 63) / 64 * 64;
                if (Kc < 256) break;
                const size_t chunks = (K + Kc - 1) / Kc, rem = K - (chunks - 1) * Kc;
                if (chunks < 2 || rem < 4) continue;
                // launch_kq's own refusal (32-bit byte offsets: a chunk of A spans M rows of the FULL K, a chunk of B Kc rows): a plan
                // it would decline must not win here — launch_cfg would run the same chunking on kernels it was not sized for
                if (((size_t)M * K + Kc) * 4 >= (size_t(1) << 32) || ((size_t)Kc * N + N) * 4 >= (size_t(1) << 32)) continue;
                // the fold, timed alone (profiles/r05/gemm_deep_k_sweep.log): 3.6-4.0 us up to 32 chunks of 100 x 100 .. 300 x 300, 4.4-5.4 at 64,
                // 6.7-10 at 128, 7-18 at 256
                const double fold = 3.5e-6 + (double)((chunks + 1) * M * N * sizeof(float)) / hbm + (chunks > 32 ? 0.025e-6 * (double)(chunks - 32) : 0.0);
                const double t = span((double)(tm * tn * chunks), Kc) + T.extra + fold + 0.5e-6;
                if (t < best.t) best = Plan{c, (unsigned)tm, (unsigned)chunks, Kc, t};
                if (t < best_kq_split.t) best_kq_split = Plan{c, (unsigned)tm, (unsigned)chunks, Kc, t};
            }
            continue;
        }
        if (c >= kFirstMidCfg) {
            // K split S ways INSIDE the launch (sgemm_dmas_kernel's distributed fold; tail_rows == 0 and S > 1 says so): few
            // tiles and a long K.  All S workgroups of a tile are resident together: tiles x S within two per CU.  The fold
            // moves 2 S |C| bytes through memory-side accesses and ends with a round of waiting (512 x 512 x 4096 on 64 x
            // 64 tiles: 69 us whole, 27.6 with S = 4, where the chunks alone would be ~22).
            if (!g_splitk || !splitk || K < 256) continue;
            const size_t tiles = tm * tn * batch;
            const unsigned max_s = kDmasMaxS[c - kFirstMidCfg];
            for (unsigned S = 2; S <= max_s && tiles * S <= (size_t)cus * 2 && tiles <= 256; S *= 2) {
                const size_t Kc = ((K + S - 1) / S + 15) / 16 * 16;
                if (Kc < 64 || (size_t)(S - 1) * Kc >= K) break;
                const double t = span((double)(tiles * S), Kc) + 2.0 * S * (double)(M * N * batch * sizeof(float)) / 3e12 + 2.5e-6;
                if (t < best.t) best = Plan{c, 0, S, Kc, t};
            }
            continue;
        }
        if (!g_splitk || !splitk || batch != 1 || K < 512) continue;
        // candidate tails: up to one machine-wave worth of tile rows, or everything
        size_t max_tail = (size_t)(cus / (double)tn) + 1;
        if (max_tail > tm) max_tail = tm;
        for (size_t pass = 0; pass <= max_tail; ++pass) {
            const size_t r = pass == 0 ? tm : pass;          // pass 0 = split everything
            if (pass != 0 && r == tm) continue;
            const size_t lead_rows = tm - r;
            const size_t m2 = M - lead_rows * T.bm;         // matrix rows in the split part
            const size_t max_s = r == tm ? 128 : 8;
            for (size_t S = 2; S <= max_s; S = S < 8 ? S + 1 : S * 2) {
                const size_t Kc = ((K + S - 1) / S + 15) / 16 * 16;
                if (Kc < 128) break;
                const size_t chunks = (K + Kc - 1) / Kc;
                if (chunks < 2) continue;
                double t = span((double)(lead_rows * tn), K);
                t += span((double)(r * tn * chunks), Kc);
                t += (double)((chunks + 1) * m2 * N * sizeof(float)) / hbm + launch;
                if (lead_rows) t += launch;
                if (t < best.t) best = Plan{c, (unsigned)r, (unsigned)chunks, Kc, t};
            }
        }
    }
    // The mid-size tiles were calibrated on one box, the older forms (and stream-K, which launch_planned weighs against what
    // is returned here) on others; where the two models are within a few per cent of each other the measured order goes
    // either way (2560^3: the model's 323 us on 128 x 128 tiles was right — and stream-K's 296 was better; profiles/r04/
    // gemm_plans_first.log).  They are taken where they are CLEARLY ahead: up to ~1100^3, thin K, ragged small products.
    if (mid_out) *mid_out = mid_ok ? best : Plan{-1, 0, 1, K, 1e300};
    if (other_out) *other_out = mid_ok ? best_other : best;
    if (!mid_ok) return best;              // (the loop never reached cfg 3: `best` is the old best)
    // One exception, measured in alternation (profiles/r04/gemm_kdeep_ab.log): a whole-K mid plan whose tiles fit the machine
    // in ONE round is the most predictable form there is (no fold, no second round) and needs no margin — 256 x 4096 x 4096:
    // both models say 75.3 us, the 64 x 64 tiles run 81.1, the 8-way split with its second launch 84.7; 4096 x 256 x 4096 82.2
    // against 87.7, 1024 x 1024 x 4096 80.7 against 86.7.
    // (the two tests are made per candidate: a several-round plan that models a shade better than a one-round one must not
    // take its place and then fail the margin — 2000^3: 64 x 64 k-quartered tiles in four rounds 128.5 us, 128 x 128 tiles in one
    // 130.0, stream-K 134.2)
    if (g_kq_splitk == 2 && best_kq_split.t < 1e299) return best_kq_split;   // (A/B: np_sgemm_set_variant(-24))
    if (g_force_kq_split_S >= 2 && batch == 1 && splitk && K >= 512) {        // (A/B: np_sgemm_set_variant(-(30000 + 1000 * shape + S)))
        const TileCfg &T = kCfg[kFirstKqCfg + g_force_kq_split_shape];
        const size_t S = (size_t)g_force_kq_split_S, Kc = ((K + S - 1) / S + 63) / 64 * 64, chunks = (K + Kc - 1) / Kc;
        if (chunks >= 2 && K - (chunks - 1) * Kc >= 4) return Plan{kFirstKqCfg + g_force_kq_split_shape, (unsigned)((M + T.bm - 1) / T.bm), (unsigned)chunks, Kc, 1e-6};
    }
    const double rival = best_other.t < t_alt ? best_other.t : t_alt;
    // The K-chunked k-quartered plans need no margin against the older forms' models either: theirs is within 5 % of the clock
    // on every shape of profiles/r05/gemm_deep_k_sweep.log, while the chunked register-staged / LDS-DMA plans they replace run
    // 5-27 % BEHIND their models (100 x 100 x 100000: 36.7 us modelled, 46.7 measured; 200 x 200 x 50000: 57.5 / 68).
    // (hence the 12 % against a rival that is such a chunked plan: 100 x 100 x 10000 modelled 9.9 against 9.2, measured 8.6 against 12.9)
    const bool rival_chunked = best_other.t <= t_alt && best_other.tail_rows > 0;
    if (best_kq_split.t < 1e299 && best_kq_split.t <= best.t && best_kq_split.t < (rival_chunked ? 1.12 : 1.0) * rival) return best_kq_split;
    if (best.t < 0.93 * rival) return best;
    if (best_one.t < 1.001 * rival) return best_one;
    return best_other;
}

int launch_plan(const Plan &p, GemmArgs g, size_t batch, bool vec);
int launch_streamk(GemmArgs g, unsigned G);
double streamk_model(size_t M, size_t N, size_t K, unsigned *grid_out);

// Until round 3 the LDS-DMA kernel wanted float4-loadable rows, and operands that were not — K % 4 or
// N % 4 != 0, odd base addresses — were first copied into padded workspaces; it now takes them as they
// are (4097^3: 124 -> see profiles/r03/gemm_unaligned.log), so this path is left for rows shorter than
// one 4-float chunk and for A/B (np_sgemm_set_variant(-6), (-3)).  The operands are copied once into zero-padded, aligned
// workspaces: A' (M x K'), B' (K' x N'), K' = K rounded up to 16, N' = N rounded up to 4.  The
// zeros contribute nothing, C is written in place with its real row length (n_store), and the
// copies cost O(M K + K N) bytes against O(M N K) flops: 4097^3 pays ~70 us of copies to run at the
// DMA kernel's rate instead of the scalar-load fallback's (97 -> 130+ TFLOP/s).  Taken only when
// the model says the copies pay (large products).
int launch_padded(const GemmArgs &g, const Plan &p, size_t Kp, size_t Np) {
    np::Scratch wa, wb;
    if (int rc = wa.alloc((size_t)g.M * Kp * sizeof(float))) return rc;
    if (int rc = wb.alloc(Kp * Np * sizeof(float))) return rc;
    hipStream_t s = np::stream();
    const unsigned cap = (unsigned)np::num_cus() * 16;
    auto blocks = [&](size_t items) { return (unsigned)std::min<size_t>((items + 255) / 256, cap); };
    pad_copy_kernel<<<blocks((size_t)g.M * Kp / 4), 256, 0, s>>>(g.A, g.lda, g.M, g.K, (float *)wa.ptr, (unsigned)Kp, g.M);
    NP_LAUNCH_CHECK("pad_copy_kernel");
    pad_copy_kernel<<<blocks(Kp * Np / 4), 256, 0, s>>>(g.B, g.ldb, g.K, g.N, (float *)wb.ptr, (unsigned)Np, (unsigned)Kp);
    NP_LAUNCH_CHECK("pad_copy_kernel");
    GemmArgs q = g;
    q.A = (const float *)wa.ptr;
    q.B = (const float *)wb.ptr;
    q.K = (unsigned)Kp;
    q.lda = (unsigned)Kp;
    q.ldb = (unsigned)Np;
    q.n_store = g.N;          // C keeps its real row length (ldc = N)
    q.N = (unsigned)Np;
    if (g_streamk >= 0 && !q.progress) {   // the padded product is as ragged as they come (4097^3: 17 x 33 tiles, 49 of them in a third wave)
        unsigned G = 0;
        const double t_sk = streamk_model(q.M, Np, Kp, &G);
        if (t_sk < 1e299 && (g_streamk > 0 || t_sk < 0.99 * p.t)) return launch_streamk(q, G);
    }
    return launch_plan(p, q, 1, true);
}

// ---- stream-K launch (sgemm_streamk_kernel) ----
// Flags live for the life of the process, one array per device (np_runtime.hip allocates it in np_init, next to the
// ticket ring: a first stream-K launch inside a stream capture must not meet a hipMalloc), zero between launches: the
// finisher that consumes workgroup p's partial puts flag[p] back to 0, so a launch leaves them as it found them (and a
// captured graph can be replayed: nothing in the launch depends on a per-launch sequence number).
constexpr unsigned kStreamKMaxGrid = np::kStreamKFlagCount;

// Modelled time of the stream-K form of an M x N x K product (one matrix; rows float4-loadable), or a huge number
// where it does not apply, and the grid to run it on: two workgroups per CU (they share the matrix pipe at the rate
// the tile kernel reaches with two resident tiles, kCfg[0].eff) or one (eff1; for few tiles: half as many partials).
// Calibrated on profiles/r03/gemm_sweep_streamk.log: on top of the K-tiles themselves a launch pays ~8 us (pipeline
// fills of up to three segments, the flag round trip) and ~3 us for every partial tile its busiest finisher folds in —
// those reads come at the very end, when nothing is left to hide them behind (5 us before the fold went through LDS-DMA:
// 2048^3 on 512 workgroups, 3 partials per tile, 118 us of K-tiles, 149 us measured then, 131 now; 1280 x 1280 x 8192:
// 10 partials, level with the tile form).
// Ragged tiles (M % 256, N % 128, K % 16) run the guarded instantiations: ~2 % slower.
double streamk_model(size_t M, size_t N, size_t K, unsigned *grid_out) {
    const double cus = (double)plan_cus(), cu_flops = 157.3e12 / 256.0;
    const size_t tm = (M + 255) / 256, tn = (N + 127) / 128, nk = (K + 15) / 16;
    *grid_out = 0;
    if (tm * tn * nk >= (1ull << 40)) return 1e300;
    const double iters = (double)(tm * tn) * (double)nk;
    const double ragged = (M % 256 || N % 128 || K % 16) ? 1.02 : 1.0;
    double best = 1e300;
    for (int per_cu = 2; per_cu >= 1; --per_cu) {
        unsigned G = (unsigned)(per_cu * cus);
        if (G > kStreamKMaxGrid) G = kStreamKMaxGrid;
        const double per_wg = ceil(iters / G);
        if (per_wg < 24.0) continue;   // ranges too short to amortise the pipeline fill
        const double eff = per_cu == 2 ? kCfg[0].eff / 2.0 : kCfg[0].eff1;
        const double t_kt = 2.0 * 256.0 * 128.0 * 16.0 / (eff * cu_flops);   // one k-tile of one workgroup
        const double folds = per_wg >= (double)nk ? 1.0 : ceil((double)nk / per_wg) - 1.0;
        const double t = per_wg * t_kt * ragged + 8e-6 + 3e-6 * folds;
        if (t < best) {
            best = t;
            *grid_out = G;
        }
    }
    return best;
}

int launch_streamk(GemmArgs g, unsigned G) {
    g.tiles_m = (g.M + 255) / 256;
    g.tiles_n = (g.N + 127) / 128;
    g.swizzle = 0;   // tile-major ranges: a workgroup's consecutive tiles (and its neighbours') share an A panel
    StreamKArgs sk;
    sk.nk = (g.K + 15) / 16;
    sk.iters_total = (unsigned long long)g.tiles_m * g.tiles_n * sk.nk;
    sk.seq = 1;
    if (sk.iters_total < G) G = (unsigned)sk.iters_total;
    sk.flags = np::streamk_flags();
    if (!sk.flags) return NP_ERR_ALLOC;   // (message set by the runtime)
    sk.error_word = np::device_error_word();
    np::Scratch ws;
    if (int rc = ws.alloc((size_t)G * 256 * 128 * sizeof(float))) return rc;
    sk.workspace = (float *)ws.ptr;
    const bool edge = g.M % 256 || g.N % 128 || g.n_store;
    const bool ktail = g.K % 16 || g.N % 4;
    hipStream_t s = np::stream();
    if (edge && ktail)
        sgemm_streamk_kernel<true, true><<<G, 256, 0, s>>>(g, sk);
    else if (edge)
        sgemm_streamk_kernel<true, false><<<G, 256, 0, s>>>(g, sk);
    else if (ktail)
        sgemm_streamk_kernel<false, true><<<G, 256, 0, s>>>(g, sk);
    else if (g_prio_period) {
        g.prio_period = g_prio_period;
        sgemm_streamk_kernel<false, false, true><<<G, 256, 0, s>>>(g, sk);
    } else
        sgemm_streamk_kernel<false, false><<<G, 256, 0, s>>>(g, sk);
    NP_LAUNCH_CHECK("sgemm_streamk_kernel");
    return NP_OK;
}

// ---- launch of sgemm_dmas_kernel ----
// shape: 0 = 128 x 128 tiles (3 LDS buffers), 1 = 128 x 64 (4), 2 = 64 x 64 (6).  S = K chunks per tile: a power of two,
// at most the shape's register-group count, chunks of at least 64 inner elements.  Returns 1 when the form does not
// apply (the caller takes another plan): S workgroups wait for each other, so tiles x S x batch must be resident at once.
// K-tile depth per shape, measured (profiles/r04/gemm_mid_bk_ab.log, same box, BK 16 -> 32): 64 x 64 tiles 1024^3 23.7 -> 22.4 us,
// 768^3 16.6 -> 15.6, 640^3 14.0 -> 13.2, 256 x 4096 x 4096 86.7 -> 81.9, 1000^3 and 1280^3 +-0 / -2 %; 128 x 64 tiles 1000^3
// 44.6 -> 40.6, else +1-3 %; 128 x 128 tiles 2 % SLOWER (their K-tile is 2048 cycles of MFMA already; 96 KiB of LDS).
typedef DmasShape<128, 128, 3> DmasShape0;
typedef DmasShape<128, 64, 3, 32> DmasShape1;    // 72 KiB of LDS
typedef DmasShape<64, 64, 4, 32> DmasShape2;     // 64 KiB
typedef DmasShape<128, 128, 3, 32> DmasShape3;   // A/B partners of 0 .. 2
// 64 x 64 on EIGHT waves: waves w and w + 4 share a tile position and split the k-groups of every K-tile; they meet in LDS
// at the end.  Both waves of a SIMD wait at the same barrier, so this hides nothing of the barrier's cost — aligned products
// lose 5-10 % to it (profiles/r04/gemm_mid_waves_ab.log: 1024^3 21.9 -> 23.7 us, 768^3 15.6 -> 16.9, 128 x 64 tiles likewise:
// that instantiation is not kept).  Products whose rows are NOT 128-byte multiples gain: 1000^3 25.4 -> 23.4 us (78 -> 85
// TFLOP/s), 1001 x 1003 x 1002 25.8 -> 24.0 — twice as many waves issue the DMAs whose rows straddle cache lines.  So: the
// whole-K 64 x 64 form of ragged / unaligned products only (np_sgemm_set_variant(-18): off, -19: on).
typedef DmasShape<64, 64, 4, 32, 2> DmasShape5;
// (one to three more LDS buffers per shape — 4 / 6 / 9 — were measured: no difference anywhere, profiles/r04/gemm_mid_depth_ab.log:
// the DMAs are far enough ahead; what a small tile loses, it loses to its barrier per K-tile and its single accumulator)

template <class SH>
void launch_dmas_shape(const GemmArgs &g, const DmasArgs &d, dim3 grid, bool edge, bool ktail, hipStream_t s) {
    if (edge && ktail)
        sgemm_dmas_kernel<SH, true, true><<<grid, SH::THREADS, 0, s>>>(g, d);
    else if (edge)
        sgemm_dmas_kernel<SH, true, false><<<grid, SH::THREADS, 0, s>>>(g, d);
    else if (ktail)
        sgemm_dmas_kernel<SH, false, true><<<grid, SH::THREADS, 0, s>>>(g, d);
    else
        sgemm_dmas_kernel<SH, false, false><<<grid, SH::THREADS, 0, s>>>(g, d);
}

int launch_dmas(int shape, GemmArgs g, unsigned batch, unsigned S) {
    if (shape < 0 || shape > 5 || shape == 4 || g.N < 4 || g.K < 4 || g.K_last || g.progress) return 1;
    const unsigned bm = kDmasBM[shape], bn = kDmasBN[shape], bk = kDmasBK[shape];
    g.tiles_m = (g.M + bm - 1) / bm;
    g.tiles_n = (g.N + bn - 1) / bn;
    g.swizzle = 0;
    const size_t tiles = (size_t)g.tiles_m * g.tiles_n;
    if (S < 1 || (S & (S - 1)) || S > kDmasMaxS[shape]) return 1;
    // XCD-aware tile order (tile_coords): workgroup b runs on XCD b % 8, and with the plain row-major order the 32
    // workgroups of one XCD need ALL of A (1024^3 on 64 x 64 tiles: 4 MiB + half a MiB of B — more than its 4 MiB L2).
    // Bands of 4 tile rows give every XCD a compact patch (4 x 8 tiles: 1 MiB of A, 2 MiB of B).  Whole-K launches only:
    // with S > 1 the S chunks of a tile sit on S different XCDs by construction.  (np_sgemm_set_variant(-16): off, -17: on)
    if (g_mid_swizzle && S == 1 && g.tiles_m >= 8 && tiles >= 64) g.swizzle = 4;
    unsigned Kc = ((g.K + S - 1) / S + bk - 1) / bk * bk;
    while (S > 1 && (Kc < 64 || (size_t)(S - 1) * Kc >= g.K)) {   // every chunk holds work
        S >>= 1;
        Kc = ((g.K + S - 1) / S + bk - 1) / bk * bk;
    }
    DmasArgs d{S, S == 1 ? (g.K + bk - 1) / bk * bk : Kc, nullptr, nullptr, np::device_error_word()};
    np::Scratch ws;
    if (S > 1) {
        // (shape 3, the A/B-only 128 x 128 tile with a 32-deep K-tile, holds 96 KiB of LDS: one workgroup per CU)
        if (tiles * S * batch > (size_t)np::num_cus() * (shape == 3 ? 1 : 2) || tiles * batch > 256) return 1;
        if (int rc = ws.alloc(tiles * batch * S * (size_t)bm * bn * sizeof(float))) return rc;
        d.workspace = (float *)ws.ptr;
        d.counters = np::next_tickets((unsigned)(tiles * batch));
        if (!d.counters) return NP_ERR_ALLOC;
    }
    if (tiles * S > 0x7fffffffu) return 1;
    const dim3 grid((unsigned)(tiles * S), 1, batch);
    const bool edge = g.M % bm || g.N % bn || g.n_store;
    const bool ktail = g.K % bk || g.N % 4;
    hipStream_t s = np::stream();
    if (shape == 0) launch_dmas_shape<DmasShape0>(g, d, grid, edge, ktail, s);
    else if (shape == 1) launch_dmas_shape<DmasShape1>(g, d, grid, edge, ktail, s);
    else if (shape == 2 && !(g_mid_waves && S == 1 && (edge || ktail))) launch_dmas_shape<DmasShape2>(g, d, grid, edge, ktail, s);
    else if (shape == 2) launch_dmas_shape<DmasShape5>(g, d, grid, edge, ktail, s);
    else if (shape == 3) launch_dmas_shape<DmasShape3>(g, d, grid, edge, ktail, s);
    else launch_dmas_shape<DmasShape5>(g, d, grid, edge, ktail, s);
    NP_LAUNCH_CHECK("sgemm_dmas_kernel");
    return NP_OK;
}
// ---- launch of sgemm_kq_kernel ----
// The tile shapes (in 16-row / 16-column blocks) and their operand stages; shape s is planner cfg kFirstKqCfg + s.  0 .. 2 are the
// three the kernel was developed on (48 x 48, 32 x 32, 64 x 64).  Returns 1 where the form does not apply (operands that are
// not float4-loadable or not below 4 GiB, a padded C, a progress request).
int g_kq_swizzle = 1;

template <int S>
void launch_kq_shape(const GemmArgs &g, dim3 grid, bool edge, bool vec, hipStream_t s) {
    constexpr KqShape sh = kKqShapes[S];
    if (vec) {
        if (edge) sgemm_kq_kernel<sh.tm, sh.tn, sh.stages, true, true><<<grid, 256, 0, s>>>(g);
        else sgemm_kq_kernel<sh.tm, sh.tn, sh.stages, false, true><<<grid, 256, 0, s>>>(g);
    } else   // rows that are not float4-loadable (odd lengths, unaligned bases, K % 4): dword loads, always the guarded form
        sgemm_kq_kernel<sh.tm, sh.tn, sh.stages, true, false><<<grid, 256, 0, s>>>(g);
}
template <int... S>
void launch_kq_any(int shape, const GemmArgs &g, dim3 grid, bool edge, bool vec, hipStream_t s, std::integer_sequence<int, S...>) {
    ((shape == S ? launch_kq_shape<S>(g, grid, edge, vec, s) : (void)0), ...);
}

int launch_kq(int shape, GemmArgs g, unsigned batch, bool vec) {
    if (shape < 0 || shape >= kKqShapeCount || g.K < 4 || (g.K_last && g.K_last < 4) || g.n_store || g.progress) return 1;
    const unsigned bm = 16 * kKqShapes[shape].tm, bn = 16 * kKqShapes[shape].tn;
    g.tiles_m = (g.M + bm - 1) / bm;
    g.tiles_n = (g.N + bn - 1) / bn;
    const size_t tiles = (size_t)g.tiles_m * g.tiles_n;
    if (tiles > 0x7fffffffu) return 1;
    if (((size_t)g.M * g.lda + g.K) * 4 >= (size_t(1) << 32) || ((size_t)g.K * g.ldb + g.N) * 4 >= (size_t(1) << 32)) return 1;   // 32-bit byte offsets
    g.swizzle = (g_kq_swizzle && g.tiles_m >= 8 && tiles >= 64) ? 4 : 0;   // XCD-aware bands, as launch_dmas
    dim3 grid((unsigned)tiles, 1, batch);
    if (g.k_chunks) {   // (launch_plan: the batch is K-chunks of one product) chunk-major over the XCDs
        if (g.k_chunks != batch || tiles * batch > 0x7fffffffu) g.k_chunks = 0;
        else {
            g.swizzle = 0;
            grid = dim3((unsigned)(tiles * batch), 1, 1);
        }
    }
    const bool edge = g.M % bm || g.N % bn;
    launch_kq_any(shape, g, grid, edge, vec, np::stream(), std::make_integer_sequence<int, kKqShapeCount>{});
    NP_LAUNCH_CHECK("sgemm_kq_kernel");
    return NP_OK;
}
int g_force_kq = -1;   // np_sgemm_set_variant(-(2000 + shape)): every product the k-quartered kernel takes goes through it (A/B, tests); -999: off

int g_force_dmas_shape = -1, g_force_dmas_S = 1;   // np_sgemm_set_variant(-(1000 + 100 * shape + S)): every tiled product through this form (A/B, tests); -999: off

// != 0: the matrices being launched are a PIECE of a batch of this many (np_comm's per-piece pipeline): planned as that
// batch, so that every piece — a single matrix included — runs the kernel configuration the whole batch would have
// run, and the pipelined result is bit-identical to the one-call form (np::sgemm_batched_piece)
thread_local size_t g_plan_batch = 0;   // (per host thread, like g_progress)

int launch_planned(GemmArgs g, size_t launch_batch, bool vec) {
    const size_t M = g.M, N = g.N, K = g.K;
    const size_t batch = g_plan_batch > launch_batch ? g_plan_batch : launch_batch;   // what the plan is made for
    const bool dma_ok = dma_takes(M, N, K, batch, vec);   // M, N edges: sgemm_dma_kernel<EDGE>; K % 16, odd K / N: fixed up in the last K-tile
    // C as a window of a wider matrix (the main block of a peeled product, launch_peeled): the split-K plans fold their
    // partials into a dense C, so they — and the pad path, which may take one — are left out
    const bool dense_c = g.ldc == g.N && g.ldb == g.N;
#ifdef NP_TUNING   // tuning builds only (python -m numpower_amd.build --tuning): plan tracing
    static const bool debug = getenv("NP_SGEMM_PLAN_DEBUG") != nullptr;
#else
    constexpr bool debug = false;
#endif
    // stream-K: equal shares of K-TILES instead of whole tiles, when the tile count does not fill the machine evenly
    // (K % 16, ragged M / N and operands of any alignment included, like every use of the LDS-DMA kernel).  Not under
    // a progress request (np_comm's pipeline counts whole tiles) and not for batches (blockIdx.z is the batch there).
    const bool sk_allowed = batch == 1 && g_streamk >= 0 && !g.progress && g.K_last == 0;
    unsigned sk_grid = 0;
    double t_sk = 1e300;
    if (sk_allowed && dma_ok) {
        t_sk = streamk_model(M, N, K, &sk_grid) / (vec ? 1.0 : 0.93);
        if (debug && t_sk < 1e299) fprintf(stderr, "[np_sgemm] %zux%zux%zu stream-K model %.1f us on %u workgroups\n", M, N, K, t_sk * 1e6, sk_grid);
    }
    Plan p = plan_sgemm(M, N, K, batch, dma_ok, false, vec, dense_c, t_sk);
    const bool take_sk = t_sk < 1e299 && (g_streamk > 0 || t_sk < 0.99 * p.t);
    if (!vec && batch == 1 && g_splitk && dense_c && N >= 1) {   // padded copies + the aligned kernel: still the faster form for the largest products
        const size_t Kp = (K + 15) / 16 * 16, Np = (N + 3) / 4 * 4;
        Plan pp = plan_sgemm(M, Np, Kp, 1, true, true);
        double t_pad = pp.t;
        if (sk_allowed) {
            unsigned G = 0;
            const double t = streamk_model(M, Np, Kp, &G);
            if (t < 0.99 * t_pad) t_pad = t;
        }
        // two pad launches: read + write of both operands
        t_pad += 2.0 * (double)((M * Kp + Kp * Np) * sizeof(float)) / 4e12 + 2 * 3e-6;
        const double t_as_is = take_sk ? t_sk : p.t;
        if (debug)
            fprintf(stderr, "[np_sgemm] %zux%zux%zu pad candidate: cfg %d tail_rows %u S %u model %.1f us vs unpadded cfg %d tail %u S %u %.1f us\n",
                    M, N, K, pp.cfg, pp.tail_rows, pp.S, t_pad * 1e6, p.cfg, p.tail_rows, p.S, t_as_is * 1e6);
        if (t_pad < t_as_is || g_force_pad) {
            if (debug)
                fprintf(stderr, "[np_sgemm] %zux%zux%zu -> padded to K %zu N %zu: cfg %d tail_rows %u S %u model %.1f us (unpadded %.1f)\n",
                        M, N, K, Kp, Np, pp.cfg, pp.tail_rows, pp.S, pp.t * 1e6, p.t * 1e6);
            return launch_padded(g, pp, Kp, Np);
        }
    }
#ifdef NP_TUNING
    // tools/gemm_plan_sweep.py: NP_SGEMM_PLAN="cfg,tail_rows,S" forces a plan, NP_SGEMM_PLAN_DEBUG prints the
    // choice.  Compiled into tuning builds only: the shipped library reads no environment variable, so a
    // stray one cannot change which kernels a product call runs.
    static const char *forced = getenv("NP_SGEMM_PLAN");
    if (forced && batch == 1) {
        int c = 2, r = 0, S = 1;
        if (sscanf(forced, "%d,%d,%d", &c, &r, &S) == 3 && c >= 0 && c < 3 && (c != 0 || dma_ok)) {
            const size_t tm = (M + kCfg[c].bm - 1) / kCfg[c].bm;
            if (r < 0 || (size_t)r > tm) r = (int)tm;
            p = Plan{c, (unsigned)r, (unsigned)S, K, 0.0};
            if (r > 0 && S >= 2) p.Kc = ((K + S - 1) / S + 15) / 16 * 16; else p.tail_rows = 0;
        }
    }
#endif
    if (debug)
        fprintf(stderr, "[np_sgemm] %zux%zux%zu batch %zu -> cfg %d tail_rows %u S %u Kc %zu model %.1f us\n", M, N, K,
                batch, p.cfg, p.tail_rows, p.S, p.Kc, p.t * 1e6);
    if (take_sk) return launch_streamk(g, sk_grid);
    return launch_plan(p, g, launch_batch, vec);
}

int launch_plan(const Plan &p, GemmArgs g, size_t batch, bool vec) {
    const size_t M = g.M, N = g.n_store ? g.n_store : g.N, K = g.K;   // N: row length of C and of the partials
    if (p.tail_rows == 0 && p.cfg >= kFirstMidCfg && p.S > 1) {
        const int rc = launch_dmas(p.cfg - kFirstMidCfg, g, (unsigned)batch, p.S);
        if (rc != 1) return rc;
    }
    if (p.tail_rows == 0) return launch_cfg(p.cfg, g, (unsigned)batch, vec);
    const TileCfg &T = kCfg[p.cfg];
    const size_t tm = (M + T.bm - 1) / T.bm;
    const size_t m1 = (tm - p.tail_rows) * T.bm, m2 = M - m1;
    if (m1) {   // leading rows: whole-K workgroups
        GemmArgs lead = g;
        lead.M = (unsigned)m1;
        if (int rc = launch_cfg(p.cfg, lead, 1, vec)) return rc;
    }
    const size_t full = K / p.Kc, rem = K - full * p.Kc, chunks = full + (rem ? 1 : 0);
    np::Scratch w;
    if (int rc = w.alloc(chunks * m2 * N * sizeof(float))) return rc;
    float *W = (float *)w.ptr;
    GemmArgs part = g;
    part.A = g.A + m1 * g.lda;
    part.M = (unsigned)m2;
    part.K = (unsigned)p.Kc;
    part.C = W;
    part.stride_a = p.Kc;
    part.stride_b = p.Kc * g.ldb;
    part.stride_c = m2 * N;
    // chunks start at multiples of 16 floats: alignment of the bases is unchanged.  The remainder
    // chunk rides in the same launch as the last batch entry (K_last): as a launch of its own it
    // would cost a whole extra work-unit time on a mostly idle machine.
    part.K_last = (unsigned)rem;
    if (p.cfg >= kFirstKqCfg && g_kq_chunk_major) part.k_chunks = (unsigned)chunks;
    if (int rc = launch_cfg(p.cfg, part, (unsigned)chunks, vec && rem % 4 == 0)) return rc;
    return np_reduce_axis(NP_SUM, W, 1, chunks, m2 * N, g.C + m1 * N, 0);
}

// C[b] (M x N, row stride ldc) = A[b] (M x K, row stride lda) * B[b] (K x N, row stride ldb)
int launch_sgemm_ld(size_t batch, size_t M, size_t N, size_t K, const float *A, size_t lda, size_t sa,
                    const float *B, size_t ldb, size_t sb, float *C, size_t ldc, size_t sc) {
    if (M > 0x7fffffffu || N > 0x7fffffffu || K > 0x7fffffffu || lda > 0x7fffffffu || ldb > 0x7fffffffu || ldc > 0x7fffffffu ||
        batch > 65535)
        return np::fail(NP_ERR_INVALID, "np_sgemm: dimension too large");
    GemmArgs g;
    g.A = A; g.B = B; g.C = C;
    g.M = (unsigned)M; g.N = (unsigned)N; g.K = (unsigned)K; g.K_last = 0; g.n_store = 0;
    g.lda = (unsigned)lda; g.ldb = (unsigned)ldb; g.ldc = (unsigned)ldc;
    g.stride_a = sa; g.stride_b = sb; g.stride_c = sc;
    g.tiles_m = g.tiles_n = 0;
    g.prio_period = 0;
    g.k_chunks = 0;
    g.probe = g_probe;
    g.progress = g_progress.counters;
    g.piece_base = g_progress.base;
    g.piece_extra = g_progress.extra;
    const bool vec = (K % 4 == 0) && (lda % 4 == 0) && (N % 4 == 0) && (ldb % 4 == 0) && aligned16(A) && aligned16(B) &&
                     (sa % 4 == 0) && (sb % 4 == 0);
    // variant = tile_code + 10 * swizzle_group ; 0 = default
#ifdef NP_TUNING
    if (g_variant >= 1000) {   // timing ablations of the pipelined kernel (wrong results!)
        switch (g_variant - 1000) {
            case 3: return launch_sgemm_pipe<128, 128, 1 + 2>(g, (unsigned)batch, vec);
            case 5: return launch_sgemm_pipe<128, 128, 1 + 4>(g, (unsigned)batch, vec);
            case 9: return launch_sgemm_pipe<128, 128, 1 + 8>(g, (unsigned)batch, vec);
            case 17: return launch_sgemm_pipe<128, 128, 1 + 16>(g, (unsigned)batch, vec);
            case 15: return launch_sgemm_pipe<128, 128, 1 + 2 + 4 + 8>(g, (unsigned)batch, vec);
            case 31: return launch_sgemm_pipe<128, 128, 1 + 2 + 4 + 8 + 16>(g, (unsigned)batch, vec);
            default: break;
        }
    }
#endif
    if (g_force_kq >= 0) {
        const int rc = launch_kq(g_force_kq, g, (unsigned)batch, vec);
        if (rc != 1) return rc;
    }
    if (g_force_dmas_shape >= 0 && !g.progress && (vec || g_dma_any_alignment)) {
        const int rc = launch_dmas(g_force_dmas_shape, g, (unsigned)batch, (unsigned)g_force_dmas_S);
        if (rc != 1) return rc;
    }
    const int tile = g_variant % 10;
    g.swizzle = (unsigned)(g_variant / 10);
    switch (tile) {
        case 1: return launch_sgemm_tile<128, 128, 16, 4>(g, (unsigned)batch, vec);
        case 2: return launch_sgemm_tile<128, 128, 32, 2>(g, (unsigned)batch, vec);
        case 3: return launch_sgemm_tile<128, 128, 16, 2>(g, (unsigned)batch, vec);
        case 4: return launch_sgemm_tile<64, 64, 16, 4>(g, (unsigned)batch, vec);
        case 5: return launch_sgemm_pipe<128, 128, 0>(g, (unsigned)batch, vec);
        case 6: return launch_sgemm_pipe<128, 128, 1>(g, (unsigned)batch, vec);
        case 7:
            if (N >= 4 && K >= 4 && (vec || g_dma_any_alignment)) return launch_cfg(0, g, (unsigned)batch, vec);   // any size: tests
            return launch_sgemm_pipe<128, 128, 1>(g, (unsigned)batch, vec);
        default: break;
    }
    return launch_planned(g, batch, vec);
}

int launch_sgemm(size_t batch, size_t M, size_t N, size_t K, const float *A, size_t lda, size_t sa,
                 const float *B, size_t sb, float *C, size_t sc) {
    return launch_sgemm_ld(batch, M, N, K, A, lda, sa, B, N, sb, C, N, sc);
}

// ---- peeling a thin ragged edge ----
// 4097^3 is 17 x 33 tiles of 256 x 128 of which a whole tile row holds ONE row of C and a whole tile column one
// column: 9 % of the matrix-core work is spent on padding, whatever the schedule.  When M % 256 (<= 32 rows) or N % 128
// (<= 2 columns) is that thin and the product large, the edge is peeled off instead:
//   main block   C[0:M0, 0:N0] = A[0:M0, :] . B[:, 0:N0]   whole tiles, C and B addressed as windows of the full
//                matrices (ldb = ldc = N; the LDS-DMA kernel reads rows of any alignment)
//   row edge     C[M0:M, :]    = A[M0:M, :] . B             an (M - M0) x K by K x N product: reads B once
//   column edge  C[0:M0, N0:N] = A[0:M0, :] . B[:, N0:N]    the columns gathered into a K x c matrix, the thin product
//                (reads A once), the result scattered into C's columns (np_copy2d both ways)
// taken when the planner's model of main + edges beats the whole product by 1.5 %: the edges are HBM-bound reads of one
// operand each (the row edge on sgemm_fewrows_kernel: 3.7 TB/s at 4097^2, 5.9 at 8192^2; the column edge on the thin
// kernels: 5.9 TB/s for one column, ~3 for two — profiles/r03/gemm_fringe_probe.log).
bool g_fewrows = true;   // np_sgemm_set_variant(-12): M <= 64 products against a large B go to the tiled kernels as before — no sgemm_fewrows_kernel / sgemm_skinny_kernel (A/B), (-13): back
int g_peel = 1;   // np_sgemm_set_variant(-9) = 0: never peel, (-10) = 1: when the model says so (default), (-11) = 2: whenever an edge is thin enough (tests)

// the planner's estimate for one product, whichever form launch_planned would pick
double estimate_product(size_t M, size_t N, size_t K, bool vec, bool dense_c) {
    const bool dma_ok = dma_takes(M, N, K, 1, vec);
    double t = plan_sgemm(M, N, K, 1, dma_ok, false, vec, dense_c).t;
    if (dma_ok && g_streamk >= 0) {
        unsigned G = 0;
        const double t_sk = streamk_model(M, N, K, &G) / (vec ? 1.0 : 0.93);
        if (t_sk < 0.99 * t) t = t_sk;
    }
    if (!vec && g_splitk && dense_c) {
        const size_t Kp = (K + 15) / 16 * 16, Np = (N + 3) / 4 * 4;
        double t_pad = plan_sgemm(M, Np, Kp, 1, true, true).t;
        if (g_streamk >= 0) {
            unsigned G = 0;
            const double t_sk = streamk_model(M, Np, Kp, &G);
            if (t_sk < 0.99 * t_pad) t_pad = t_sk;
        }
        t_pad += 2.0 * (double)((M * Kp + Kp * Np) * sizeof(float)) / 4e12 + 2 * 3e-6;
        if (t_pad < t) t = t_pad;
    }
    return t;
}

// 1 = not peeled (the caller carries on), else the status of the peeled product
int try_peeled(size_t M, size_t N, size_t K, const float *A, const float *B, float *C) {
    if (!g_peel || g_variant != 0 || g_progress.counters || (g_peel == 1 && 2.0 * (double)M * (double)N * (double)K < 2e10)) return 1;
    const size_t r = M % 256, c = N % 128;
    const size_t m_cut = (r && r <= 32 && M > 2048) ? M - r : M, n_cut = (c && c <= 2 && N > 2048) ? N - c : N;
    if (m_cut == M && n_cut == N) return 1;
    const bool aligned = aligned16(A) && aligned16(B);
    const bool vec_full = K % 4 == 0 && N % 4 == 0 && aligned;
    const double t_full = estimate_product(M, N, K, vec_full, true);
    double best = g_peel == 2 ? 1e300 : 0.985 * t_full;   // (the model is ~5 % pessimistic about the peeled form: 5121^3 modelled 0.967, measured 0.93)
    size_t M0 = M, N0 = N;
    for (int pick = 1; pick < 4; ++pick) {   // rows only, columns only, both
        const size_t m0 = (pick & 1) ? m_cut : M, n0 = (pick & 2) ? n_cut : N;
        if ((m0 == M && (pick & 1)) || (n0 == N && (pick & 2))) continue;
        double t = estimate_product(m0, n0, K, vec_full, n0 == N);
        if (m0 != M) t += (double)K * (double)N * 4.0 / (M - m0 <= 8 ? 4.5e12 : 3.6e12) + 6e-6;   // sgemm_fewrows_kernel (<= 8 rows) / sgemm_skinny_kernel + the fold
        if (n0 != N) t += (double)m0 * (double)K * 4.0 / (N - n0 == 1 ? 5e12 : 3e12) + 10e-6;   // sgemv_kernel / the MFMA thin kernel, + the two column copies
        if (t < best) { best = t; M0 = m0; N0 = n0; }
    }
    if (M0 == M && N0 == N) return 1;
    if (int rc = launch_sgemm_ld(1, M0, N0, K, A, K, 0, B, N, 0, C, N, 0)) return rc;
    if (M0 != M)
        if (int rc = np_sgemm_strided_batched(1, M - M0, N, K, A + M0 * K, 0, B, 0, C + M0 * N, 0)) return rc;
    if (N0 != N) {
        const size_t cw = N - N0;
        np::Scratch tb, tc;
        if (int rc = tb.alloc(K * cw * sizeof(float))) return rc;
        if (int rc = tc.alloc(M0 * cw * sizeof(float))) return rc;
        if (int rc = np_copy2d((float *)tb.ptr, cw, B + N0, N, cw, K)) return rc;
        if (int rc = np_sgemm_strided_batched(1, M0, cw, K, A, 0, (const float *)tb.ptr, 0, (float *)tc.ptr, 0)) return rc;
        if (int rc = np_copy2d(C + N0, N, (const float *)tc.ptr, cw, cw, M0)) return rc;
    }
    return NP_OK;
}

// N <= 32: the GEMV-with-several-right-hand-sides kernels.  Returns 1 when the shape is left to the
// tiled kernels (few rows and a short K: nothing to gain).
template <int NV>
int launch_thin_nv(size_t M, size_t N, size_t K, const float *A, const float *B, float *C) {
    hipStream_t s = np::stream();
    const size_t target = (size_t)np::num_cus() * 8;
    // measured (profiles/r01/skinny_gemm.log): with N <= 4 the per-k loads of B are few enough that the
    // kernel stays HBM-bound (10^7 x 3 x 3: 0.53 -> 0.045 ms); with 8-32 accumulators it turns
    // load-issue-bound and loses to the tiled kernels, so those shapes are left to the planner
    if (N == 1 && M >= 2048 && K >= 256) return np_sgemv(M, K, A, B, C);   // a matrix . vector product: a wave per row, float4 loads
    // (a long K on few rows: the MFMA kernel over K-chunks below reads A at 3.3 TB/s where this one's 4-byte loads reach 2)
    if (M >= 2048 && N <= 4 && K * N <= 16384 && (K < 1024 || M >= 65536)) {   // B stays in the vector L1 / L2 while every row streams past it
        size_t L = 1;
        while (L * 8 < K && L < 64) L *= 2;    // ~8 k per lane
        const size_t blocks = (M * L + 255) / 256;
        if (blocks > 0x7fffffffu) return 1;
#define NP_TH(L_) sgemm_thin_kernel<NV, L_><<<(unsigned)blocks, 256, 0, s>>>(A, B, C, M, (unsigned)N, (unsigned)K)
        switch (L) {
            case 1: NP_TH(1); break;
            case 2: NP_TH(2); break;
            case 4: NP_TH(4); break;
            case 8: NP_TH(8); break;
            case 16: NP_TH(16); break;
            case 32: NP_TH(32); break;
            default: NP_TH(64); break;
        }
#undef NP_TH
        NP_LAUNCH_CHECK("sgemm_thin_kernel");
        return NP_OK;
    }
    if (M >= 2048 && (N > 4 || (K >= 1024 && M < 65536)) && N <= 32 && K >= 4) {
        const size_t rows_per_block = N <= 16 ? 64 : 128;
        const size_t blocks = (M + rows_per_block - 1) / rows_per_block;
        if (blocks > 0x7fffffffu) return 1;
        // Not enough rows to fill the machine with one workgroup per 64 / 128 rows (4096 x 8 x 4097: 64 workgroups, 90 us
        // for 67 MB): K is cut into chunks as for M < 2048 below, partials [chunk][M][N] folded by np_reduce_axis.
        if (blocks < target && K >= 1024) {
            size_t chunks = (target + blocks - 1) / blocks;
            if (chunks > K / 512) chunks = K / 512;
            if (chunks > 65535) chunks = 65535;
            if (chunks >= 2) {
                size_t kc = (K + chunks - 1) / chunks;
                kc = (kc + 63) / 64 * 64;
                chunks = (K + kc - 1) / kc;
                np::Scratch partial;
                if (int rc = partial.alloc(chunks * M * N * sizeof(float))) return rc;
                const dim3 grid((unsigned)blocks, (unsigned)chunks);
                if (N <= 16) sgemm_thin_mfma_kernel<16><<<grid, 256, 0, s>>>(A, B, (float *)partial.ptr, M, (unsigned)N, (unsigned)K, (unsigned)kc);
                else sgemm_thin_mfma_kernel<32><<<grid, 256, 0, s>>>(A, B, (float *)partial.ptr, M, (unsigned)N, (unsigned)K, (unsigned)kc);
                NP_LAUNCH_CHECK("sgemm_thin_mfma_kernel");
                return np_reduce_axis(NP_SUM, (const float *)partial.ptr, 1, chunks, M * N, C, 0);
            }
        }
        if (N <= 16) sgemm_thin_mfma_kernel<16><<<(unsigned)blocks, 256, 0, s>>>(A, B, C, M, (unsigned)N, (unsigned)K, (unsigned)K);
        else sgemm_thin_mfma_kernel<32><<<(unsigned)blocks, 256, 0, s>>>(A, B, C, M, (unsigned)N, (unsigned)K, (unsigned)K);
        NP_LAUNCH_CHECK("sgemm_thin_mfma_kernel");
        return NP_OK;
    }
    // measured (profiles/r01/skinny_gemm.log): wins for one or two row tiles and N > 16 (32 x 64 x 2e6: 0.334 -> 0.199 ms,
    // 32 x 32 x 2e6: 0.183 -> 0.114); at four row tiles, or N <= 16 (the 16x16x4 tile), the LDS-staged kernel below is ahead
    if (M > 16 && M <= 64 && N > 16 && K >= 16384) {
        const size_t row_tiles = (M + 31) / 32;
        size_t chunks = (target * 4 * 2 + row_tiles - 1) / row_tiles;      // wave-chunks
        const size_t max_chunks = K / 1024;
        if (chunks > max_chunks) chunks = max_chunks;
        if (chunks >= 8) {
            size_t kc = (K + chunks - 1) / chunks;
            kc = (kc + 31) / 32 * 32;
            chunks = (K + kc - 1) / kc;
            const size_t groups = (chunks + 3) / 4;
            if (thin_to_planner(row_tiles * groups)) return 1;
            if (groups <= 65535) {
                np::Scratch partial;
                if (int rc = partial.alloc(chunks * M * N * sizeof(float))) return rc;
                const dim3 grid((unsigned)row_tiles, (unsigned)groups);
                if (N <= 32) sgemm_fewrows_splitk_kernel<1><<<grid, 256, 0, s>>>(A, B, (float *)partial.ptr, (unsigned)M, (unsigned)N, (unsigned)K, (unsigned)kc);
                else sgemm_fewrows_splitk_kernel<2><<<grid, 256, 0, s>>>(A, B, (float *)partial.ptr, (unsigned)M, (unsigned)N, (unsigned)K, (unsigned)kc);
                NP_LAUNCH_CHECK("sgemm_fewrows_splitk_kernel");
                return np_reduce_axis(NP_SUM, (const float *)partial.ptr, 1, chunks, M * N, C, 0);
            }
        }
    }
    if (M > 16 && M < 2048 && K >= 16384) {
        // a few hundred rows, a very long K: the same kernel over K-chunks (the tiled split-K plan reads A at
        // 2.2 TB/s on 64-wide tiles that are mostly padding), partials [chunk][M][N] folded by np_reduce_axis
        const size_t rows_per_block = N <= 16 ? 64 : 128;
        const size_t blocks = (M + rows_per_block - 1) / rows_per_block;
        size_t chunks = (target * 2 + blocks - 1) / blocks;
        const size_t max_chunks = K / 2048;
        if (chunks > max_chunks) chunks = max_chunks;
        if (chunks > 65535) chunks = 65535;
        if (chunks >= 2) {
            size_t kc = (K + chunks - 1) / chunks;
            kc = (kc + 63) / 64 * 64;
            chunks = (K + kc - 1) / kc;
            if (thin_to_planner(blocks * chunks)) return 1;
            np::Scratch partial;
            if (int rc = partial.alloc(chunks * M * N * sizeof(float))) return rc;
            const dim3 grid((unsigned)blocks, (unsigned)chunks);
            if (N <= 16) sgemm_thin_mfma_kernel<16><<<grid, 256, 0, s>>>(A, B, (float *)partial.ptr, M, (unsigned)N, (unsigned)K, (unsigned)kc);
            else if (N <= 32) sgemm_thin_mfma_kernel<32><<<grid, 256, 0, s>>>(A, B, (float *)partial.ptr, M, (unsigned)N, (unsigned)K, (unsigned)kc);
            else sgemm_thin_mfma_kernel<32, 2><<<grid, 256, 0, s>>>(A, B, (float *)partial.ptr, M, (unsigned)N, (unsigned)K, (unsigned)kc);
            NP_LAUNCH_CHECK("sgemm_thin_mfma_kernel");
            return np_reduce_axis(NP_SUM, (const float *)partial.ptr, 1, chunks, M * N, C, 0);
        }
    }
    if (M <= 16 && N <= 8 && K >= 65536) {   // X^T X of a tall-skinny X
        size_t chunks = (2 * target + M - 1) / M;
        const size_t max_chunks = K / 4096;
        if (chunks > max_chunks) chunks = max_chunks;
        if (chunks < 1) chunks = 1;
        const size_t chunk_len = (K + chunks - 1) / chunks;
        chunks = (K + chunk_len - 1) / chunk_len;
        if (chunks > 65535) return 1;
        np::Scratch partial;
        if (int rc = partial.alloc(M * chunks * N * sizeof(float))) return rc;
        sgemm_thin_chunks_kernel<NV><<<dim3((unsigned)chunks, (unsigned)M), 256, 0, s>>>(
            A, B, (float *)partial.ptr, (unsigned)N, (unsigned)K, (unsigned)chunk_len);
        NP_LAUNCH_CHECK("sgemm_thin_chunks_kernel");
        return np_reduce_axis(NP_SUM, (const float *)partial.ptr, M, chunks, N, C, 0);
    }
    return 1;
}

// M <= 8 rows against a B of at least a few MB: sgemm_fewrows_kernel.  Returns 1 when the shape is left to the tiled kernels.
int launch_fewrows(size_t M, size_t N, size_t K, const float *A, const float *B, float *C) {
    // (B below ~24 MB: the tiled kernels are level or ahead — 1 x 2049 x 2049 14 us against 16, profiles/r03/gemm_fringe_probe.log)
    if (M > 8 || N < 256 || K < 256 || (double)K * (double)N < 6e6 || N > 0x7fffffffu || K > 0x7fffffffu) return 1;
    const size_t col_blocks = (N + 255) / 256;
    // ~6 workgroups per CU; a chunk is at least 16 rows (four per wave) and a multiple of 4
    size_t chunks = ((size_t)np::num_cus() * 6 + col_blocks - 1) / col_blocks;
    if (chunks > K / 16) chunks = K / 16;
    if (chunks < 1) chunks = 1;
    if (chunks > 65535) chunks = 65535;
    size_t chunk_len = ((K + chunks - 1) / chunks + 3) / 4 * 4;
    chunks = (K + chunk_len - 1) / chunk_len;
    np::Scratch partial;
    float *out = C;
    if (chunks > 1) {
        if (int rc = partial.alloc(chunks * M * N * sizeof(float))) return rc;
        out = (float *)partial.ptr;
    }
    const dim3 grid((unsigned)col_blocks, (unsigned)chunks);
    hipStream_t s = np::stream();
#define NP_FR(MV_) sgemm_fewrows_kernel<MV_><<<grid, 256, 0, s>>>(A, B, out, (unsigned)M, (unsigned)N, (unsigned)K, (unsigned)chunk_len)
    if (M == 1) NP_FR(1);
    else if (M == 2) NP_FR(2);
    else if (M <= 4) NP_FR(4);
    else NP_FR(8);
#undef NP_FR
    NP_LAUNCH_CHECK("sgemm_fewrows_kernel");
    if (chunks > 1) return np_reduce_axis(NP_SUM, out, 1, chunks, M * N, C, 0);
    return NP_OK;
}

// 9 <= M <= 32 rows against a large B: sgemm_skinny_kernel.  Returns 1 when the shape is left to the tiled kernels.
int launch_skinny(size_t M, size_t N, size_t K, const float *A, const float *B, float *C) {
    // (33 ... 64 rows — two 32-row blocks, 200 registers, two waves per SIMD — measured 134 us against the tiles' 93 on
    // 64 x 8192 x 8192: left to them; the RB = 2 instantiation stays for the day that is worth tuning)
    if (M < 9 || M > 32 || N < 512 || K < 512 || (double)K * (double)N < 16e6 || N > 0x7fffffffu || K > 0x7fffffffu) return 1;
    const size_t col_blocks = (N + 127) / 128;
    // ~3 workgroups per CU, as long as the partials stay under ~15 % of the bytes of B; chunks are multiples of 64 rows
    // (four waves x 16-row tiles)
    size_t chunks = ((size_t)np::num_cus() * 3 + col_blocks - 1) / col_blocks;
    const size_t cap = (size_t)(0.075 * (double)K / (double)M);
    if (chunks > cap) chunks = cap;
    if (chunks > K / 64) chunks = K / 64;
    if (chunks < 1) chunks = 1;
    if (chunks > 65535) chunks = 65535;
    size_t chunk_len = ((K + chunks - 1) / chunks + 63) / 64 * 64;
    chunks = (K + chunk_len - 1) / chunk_len;
    np::Scratch partial;
    float *out = C;
    if (chunks > 1) {
        if (int rc = partial.alloc(chunks * M * N * sizeof(float))) return rc;
        out = (float *)partial.ptr;
    }
    const dim3 grid((unsigned)col_blocks, (unsigned)chunks);
    hipStream_t s = np::stream();
    if (M <= 32)
        sgemm_skinny_kernel<1><<<grid, 256, 0, s>>>(A, B, out, (unsigned)M, (unsigned)N, (unsigned)K, (unsigned)chunk_len);
    else
        sgemm_skinny_kernel<2><<<grid, 256, 0, s>>>(A, B, out, (unsigned)M, (unsigned)N, (unsigned)K, (unsigned)chunk_len);
    NP_LAUNCH_CHECK("sgemm_skinny_kernel");
    if (chunks > 1) return np_reduce_axis(NP_SUM, out, 1, chunks, M * N, C, 0);
    return NP_OK;
}

int launch_thin(size_t M, size_t N, size_t K, const float *A, const float *B, float *C) {
    if (N <= 4) return launch_thin_nv<4>(M, N, K, A, B, C);
    if (N <= 8) return launch_thin_nv<8>(M, N, K, A, B, C);
    if (N <= 16) return launch_thin_nv<16>(M, N, K, A, B, C);
    return launch_thin_nv<32>(M, N, K, A, B, C);
}

}  // namespace

namespace np {

// The batched GEMM as ONE launch whose workgroups report progress (GemmArgs::progress): counters[c] += 1 for every
// finished tile of piece c, the np_comm_piece split of `batch` into `chunks`.  *tiles_per_matrix = how many tiles one
// batch entry contributes (piece c is complete at count(c) * tiles_per_matrix), or 0 when this shape's plan is not a
// single launch of the LDS-DMA kernel (small or unaligned matrices) — nothing has been launched then and the caller
// issues one launch per piece instead.
int sgemm_batched_with_progress(size_t batch, size_t M, size_t N, size_t K, const float *A, size_t stride_a, const float *B,
                                size_t stride_b, float *C, size_t stride_c, unsigned *counters, int chunks,
                                unsigned *tiles_per_matrix) {
    *tiles_per_matrix = 0;
    // batch >= 2: the planner never splits K or pads there, the plan is one launch_cfg (plan_sgemm / launch_planned)
    if (batch < 2 || batch > 65535 || M == 0 || N == 0 || K == 0 || chunks < 1 || (size_t)chunks > batch || g_variant != 0 ||
        !A || !B || !C || !counters)
        return NP_OK;
    if (int rc = np::ensure_init()) return rc;
    // only the LDS-DMA kernel reports progress: ask the planner BEFORE anything is launched
    const size_t lda = K;
    const bool vec = (K % 4 == 0) && (lda % 4 == 0) && (N % 4 == 0) && aligned16(A) && aligned16(B) && (stride_a % 4 == 0) &&
                     (stride_b % 4 == 0);
    const bool dma_ok = dma_takes(M, N, K, batch, vec);
    if (!dma_ok) return NP_OK;
    const Plan p = plan_sgemm(M, N, K, batch, dma_ok, false, vec);
    if (p.cfg != 0 || p.tail_rows != 0) return NP_OK;
    g_progress = ProgressRequest{counters, (unsigned)(batch / (size_t)chunks), (unsigned)(batch % (size_t)chunks), 0, 0};
    const int rc = launch_sgemm(batch, M, N, K, A, lda, stride_a, B, stride_b, C, stride_c);
    const ProgressRequest done = g_progress;
    g_progress = ProgressRequest{};
    if (rc) return rc;
    if (done.launches != 1)
        return np::fail(NP_ERR_DEVICE, "np_sgemm: internal error: a progress-reporting GEMM made %u launches", done.launches);
    *tiles_per_matrix = done.tiles_per_matrix;
    return NP_OK;
}

// `count` matrices that are a piece of a batch of `whole`: the kernels and tile configuration np_sgemm_strided_batched
// would pick for the whole batch — a one-matrix piece must not wander off to the single-product planner (split-K,
// stream-K, peeled edges ...), or the pipelined result would differ from the one-call form in its last bits.
int sgemm_batched_piece(size_t count, size_t whole, size_t M, size_t N, size_t K, const float *A, size_t stride_a, const float *B,
                        size_t stride_b, float *C, size_t stride_c) {
    g_plan_batch = whole;
    const int rc = np_sgemm_strided_batched(count, M, N, K, A, stride_a, B, stride_b, C, stride_c);
    g_plan_batch = 0;
    return rc;
}

}  // namespace np

extern "C" {

// debug: device buffer of 8 x u64 per workgroup, filled by the next np_sgemm launches (null = off)
int np_debug_sgemm_probe(void *dev_buf) {
    g_probe = (unsigned long long *)dev_buf;
    return NP_OK;
}

// debug: what the planner would run for one M x N x K product (aligned, dense operands) on a device of `cus` CUs (0 = the
// current device) — out[0..2] = cfg / tail_rows / S of the chosen tiled plan, out[3] its modelled us, out[4] 1 if stream-K is
// taken instead and out[5] its modelled us (1e300 = not applicable), out[6..8] = cfg / S / us of the best mid-size plan,
// out[9..10] = cfg / us of the best plan without the mid-size tiles.  Host arithmetic only: no device needed when cus != 0.
int np_sgemm_debug_plan(size_t M, size_t N, size_t K, size_t batch, int cus, double *out) {
    if (!out || !M || !N || !K || !batch || cus < 0) return np::fail(NP_ERR_INVALID, "np_sgemm_debug_plan: bad argument");
    if (cus == 0)
        if (int rc = np::ensure_init()) return rc;
    g_plan_cus = cus;
    const bool vec = K % 4 == 0 && N % 4 == 0;
    const bool dma_ok = dma_takes(M, N, K, batch, vec);
    unsigned G = 0;
    double t_sk = 1e300;
    if (batch == 1 && g_streamk >= 0 && dma_ok) t_sk = streamk_model(M, N, K, &G) / (vec ? 1.0 : 0.93);
    Plan mid, other;
    const Plan p = plan_sgemm(M, N, K, batch, dma_ok, false, vec, true, t_sk, &mid, &other);
    g_plan_cus = 0;
    out[0] = p.cfg; out[1] = p.tail_rows; out[2] = p.S; out[3] = p.t * 1e6;
    out[4] = t_sk < 1e299 && (g_streamk > 0 || t_sk < 0.99 * p.t);
    out[5] = t_sk < 1e299 ? t_sk * 1e6 : 1e300;
    out[6] = mid.cfg; out[7] = mid.S; out[8] = mid.t < 1e299 ? mid.t * 1e6 : 1e300;
    out[9] = other.cfg; out[10] = other.t < 1e299 ? other.t * 1e6 : 1e300;
    return NP_OK;
}

int np_sgemm_set_variant(int variant) {
    if (variant <= -30000) {
        const int code = -variant - 30000;
        if (code / 1000 >= kKqShapeCount) return np::fail(NP_ERR_INVALID, "np_sgemm_set_variant: -(30000 + 1000 * shape + S): no such shape");
        g_force_kq_split_shape = code / 1000;
        g_force_kq_split_S = code % 1000;
        return NP_OK;
    }
    if (variant <= -999) {   // -(1000 + 100 * shape + S): sgemm_dmas_kernel with that tile shape and S K-chunks wherever it applies; -999: off
        if (variant == -999) {
            g_force_dmas_shape = -1;
            g_force_kq = -1;
            return NP_OK;
        }
        if (variant <= -2000) {   // -(2000 + shape): sgemm_kq_kernel wherever it applies
            if (variant < -2000 - 63) return np::fail(NP_ERR_INVALID, "np_sgemm_set_variant: -(2000 + shape): no such shape");
            g_force_kq = -variant - 2000;
            return NP_OK;
        }
        const int code = -variant - 1000;
        if (code / 100 > 5 || code % 100 < 1) return np::fail(NP_ERR_INVALID, "np_sgemm_set_variant: -(1000 + 100 * shape + S) with shape 0..5, S >= 1");
        g_force_dmas_shape = code / 100;
        g_force_dmas_S = code % 100;
        return NP_OK;
    }
    if (variant <= -40 && variant >= -99) {   // -(40 + q): see g_thin_underfilled_to_planner
        g_thin_underfilled_to_planner = -variant - 40;
        return NP_OK;
    }
    if (variant <= -100) {   // -(100 + p): priority alternation between co-resident workgroups, p K-tiles per phase (p = 0: off)
        g_prio_period = (unsigned)(-variant - 100);
        g_prio_min_k = g_prio_period == 16 ? 2048u : 0u;
        return NP_OK;
    }
    if (variant < 0) {   // -1: whole-K plans only, -2: default planner, -3: default + forced operand padding, -4 / -5: stream-K always / never
        if (variant == -16 || variant == -17) {
            g_mid_swizzle = variant == -17;
            return NP_OK;
        }
        if (variant == -18 || variant == -19) {
            g_mid_waves = variant == -19;
            return NP_OK;
        }
        if (variant == -20 || variant == -21) {
            g_kq_tiles = variant == -21;
            return NP_OK;
        }
        if (variant == -25 || variant == -26) {
            g_kq_chunk_major = variant == -26;
            return NP_OK;
        }
        if (variant == -24) {   // (A/B) the K-chunked k-quartered plan wherever one exists
            g_kq_splitk = 2;
            return NP_OK;
        }
        if (variant == -22 || variant == -23) {   // -22: deep-K products on the register-staged tiles as before round 5, -23: back (the K-chunked k-quartered plans)
            g_kq_splitk = variant == -23;
            return NP_OK;
        }
        if (variant == -14 || variant == -15) {   // -14: no mid-size LDS-DMA tiles (the plans of round 3), -15: back
            g_mid_tiles = variant == -15;
            return NP_OK;
        }
        if (variant == -12 || variant == -13) {   // -12: no sgemm_fewrows_kernel (M <= 8 on the tiled kernels, as before), -13: back
            g_fewrows = variant == -13;
            return NP_OK;
        }
        if (variant <= -9 && variant >= -11) {   // -9: never peel a thin ragged edge off a large product, -11: always when one is thin enough, -10: back to the default (the model decides)
            g_peel = variant == -9 ? 0 : variant == -11 ? 2 : 1;
            return NP_OK;
        }
        if (variant <= -6 && variant >= -8) {   // -6: LDS-DMA kernel for float4-loadable operands only (the pad-copy path for the rest), -8: for any operands of any size, -7: back to the default
            g_dma_any_alignment = variant == -6 ? 0 : variant == -8 ? 2 : 1;
            return NP_OK;
        }
        g_splitk = variant != -1;
        g_force_pad = variant == -3;
        g_streamk = variant == -4 ? 1 : (variant == -5 || variant == -1) ? -1 : 0;   // -1: one whole tile per workgroup, nothing else
        return NP_OK;
    }
    // variants >= 1000 switch parts of the pipelined kernel OFF to time them (tools/gemm_ab.py): the
    // products they compute are wrong by construction, so they exist in tuning builds (-DNP_TUNING) only
#ifdef NP_TUNING
    if (variant >= 1000 && !getenv("NP_ALLOW_ABLATION"))
        return np::fail(NP_ERR_INVALID, "np_sgemm_set_variant: ablation variants need NP_ALLOW_ABLATION=1");
#else
    if (variant >= 1000)
        return np::fail(NP_ERR_INVALID, "np_sgemm_set_variant: ablation variants exist only in tuning builds (-DNP_TUNING)");
#endif
    g_variant = variant;
    return NP_OK;
}

int np_sgemm(size_t M, size_t N, size_t K, const float *A, const float *B, float *C) {
    return np_sgemm_strided_batched(1, M, N, K, A, 0, B, 0, C, 0);
}

int np_sgemm_strided_batched(size_t batch, size_t M, size_t N, size_t K, const float *A,
                             size_t stride_a, const float *B, size_t stride_b, float *C,
                             size_t stride_c) {
    if (batch == 0 || M == 0 || N == 0) return NP_OK;
    if (!C) return np::fail(NP_ERR_INVALID, "np_sgemm: null output");
    if (int rc = np::ensure_init()) return rc;
    if (K == 0) {
        // empty inner dimension: C = 0 (beta = 0)
        for (size_t b = 0; b < batch; ++b)
            if (int rc = np_memset0(C + b * stride_c, M * N * sizeof(float))) return rc;
        return NP_OK;
    }
    if (!A || !B) return np::fail(NP_ERR_INVALID, "np_sgemm: null input");
    if (g_plan_batch > 1) {   // a piece of a larger batch: straight to the tiled kernels, planned as that batch (launch_planned)
        for (size_t b0 = 0; b0 < batch; b0 += 65535) {
            const size_t nb = batch - b0 < 65535 ? batch - b0 : 65535;
            if (int rc = launch_sgemm(nb, M, N, K, A + b0 * stride_a, K, stride_a, B + b0 * stride_b, stride_b, C + b0 * stride_c, stride_c))
                return rc;
        }
        return NP_OK;
    }
    if (batch == 1 && M <= 0x7fffffffu && N <= 0x7fffffffu && K <= 0x7fffffffu) {
        const int rc = try_peeled(M, N, K, A, B, C);
        if (rc != 1) return rc;   // 1 = not peeled
    }
    // (N up to 64 for the split-K thin path only: a few hundred rows x a very long K, two 32-column blocks)
    if (batch == 1 && (N <= 32 || (N <= 64 && M < 2048 && K >= 16384)) && g_variant == 0 && K <= 0x7fffffffu) {
        const int rc = launch_thin(M, N, K, A, B, C);
        if (rc != 1) return rc;   // 1 = shape not taken
    }
    if (batch == 1 && M <= 8 && g_variant == 0 && g_fewrows) {
        const int rc = launch_fewrows(M, N, K, A, B, C);
        if (rc != 1) return rc;   // 1 = shape not taken
    }
    if (batch == 1 && M >= 9 && M <= 32 && g_variant == 0 && g_fewrows) {
        const int rc = launch_skinny(M, N, K, A, B, C);
        if (rc != 1) return rc;   // 1 = shape not taken
    }
    if (batch == 1 && M <= 16 && K <= 64 && N >= 65536 && g_variant == 0) {
        size_t blocks = (N / 4 + 255) / 256;
        const size_t cap = (size_t)np::num_cus() * 64;
        if (blocks > cap) blocks = cap;
        if (blocks < 1) blocks = 1;
        if (M <= 4)
            sgemm_thin_left_kernel<4><<<(unsigned)blocks, 256, 0, np::stream()>>>(A, B, C, (unsigned)M, N, (unsigned)K);
        else if (M <= 8)
            sgemm_thin_left_kernel<8><<<(unsigned)blocks, 256, 0, np::stream()>>>(A, B, C, (unsigned)M, N, (unsigned)K);
        else
            sgemm_thin_left_kernel<16><<<(unsigned)blocks, 256, 0, np::stream()>>>(A, B, C, (unsigned)M, N, (unsigned)K);
        NP_LAUNCH_CHECK("sgemm_thin_left_kernel");
        return NP_OK;
    }
    // blockIdx.z carries the batch index: more than 65535 matrices go in slabs
    for (size_t b0 = 0; b0 < batch; b0 += 65535) {
        const size_t nb = batch - b0 < 65535 ? batch - b0 : 65535;
        if (int rc = launch_sgemm(nb, M, N, K, A + b0 * stride_a, K, stride_a, B + b0 * stride_b, stride_b,
                                  C + b0 * stride_c, stride_c))
            return rc;
    }
    return NP_OK;
}

int np_sgemm_strided_batched_piece(size_t count, size_t whole, size_t M, size_t N, size_t K, const float *A, size_t stride_a,
                                   const float *B, size_t stride_b, float *C, size_t stride_c) {
    if (whole < count) return np::fail(NP_ERR_INVALID, "np_sgemm_strided_batched_piece: a piece of %zu matrices of a batch of %zu", count, whole);
    return np::sgemm_batched_piece(count, whole, M, N, K, A, stride_a, B, stride_b, C, stride_c);
}

int np_sgemv(size_t M, size_t N, const float *A, const float *x, float *y) {
    if (M == 0) return NP_OK;
    if (!A || !x || !y) return np::fail(NP_ERR_INVALID, "np_sgemv: null pointer");
    if (M > 0x7fffffffu || N > 0x7fffffffu)
        return np::fail(NP_ERR_INVALID, "np_sgemv: dimension too large");
    if (int rc = np::ensure_init()) return rc;
    const size_t target = (size_t)np::num_cus() * 8;
    if (N >= 4 && N <= 63 && M >= 262144 && aligned16(A)) {   // (N = 3 and small M: the lane-group kernel below is ahead)
        // slabs of R rows, R a multiple of 256 (so that R * N % 4 == 0 and every thread has whole rows), ~32 KB of LDS
        const unsigned pitch = (unsigned)N | 1u;
        unsigned R = (8192u / pitch) / 256u * 256u;
        if (R < 256) R = 256;
        const size_t blocks = (M + R - 1) / R;
        const size_t lds = ((size_t)R * pitch + N) * sizeof(float);
        if (blocks <= 0x7fffffffu && lds <= 64 * 1024) {
            const unsigned magic = (unsigned)((0x100000000ull + N - 1) / N);
            sgemv_staged_rows_kernel<<<(unsigned)blocks, 256, lds, np::stream()>>>(A, x, y, M, (unsigned)N, R, magic);
            NP_LAUNCH_CHECK("sgemv_staged_rows_kernel");
            return NP_OK;
        }
    }
    if (N <= 256 && M >= 1024) {
        size_t L = 1;
        while (L * 8 < N) L *= 2;          // ~8 elements per lane
        const size_t blocks = (M * L + 255) / 256;
        if (blocks <= 0x7fffffffu) {
#define NP_SR(L_) sgemv_short_rows_kernel<L_><<<(unsigned)blocks, 256, 0, np::stream()>>>(A, x, y, M, (unsigned)N)
            switch (L) {
                case 1: NP_SR(1); break;
                case 2: NP_SR(2); break;
                case 4: NP_SR(4); break;
                case 8: NP_SR(8); break;
                case 16: NP_SR(16); break;
                default: NP_SR(32); break;
            }
#undef NP_SR
            NP_LAUNCH_CHECK("sgemv_short_rows_kernel");
            return NP_OK;
        }
    }
    if (M >= 2 && M <= 16 && N >= 65536) {   // a few long rows: a workgroup per chunk of x takes all of them (x is read once)
        size_t chunks = 2 * target;
        const size_t max_chunks = N / 1024;
        if (chunks > max_chunks) chunks = max_chunks;
        const size_t chunk_len = ((N + chunks - 1) / chunks + 3) / 4 * 4;
        chunks = (N + chunk_len - 1) / chunk_len;
        np::Scratch partial;
        if (int rc = partial.alloc(M * chunks * sizeof(float))) return rc;
        sgemv_fewrows_chunks_kernel<<<(unsigned)chunks, 256, 0, np::stream()>>>(A, x, (float *)partial.ptr, (unsigned)M, (unsigned)N,
                                                                              (unsigned)chunk_len);
        NP_LAUNCH_CHECK("sgemv_fewrows_chunks_kernel");
        return np_reduce_axis(NP_SUM, (const float *)partial.ptr, M, chunks, 1, y, 0);
    }
    if (M < 2 * target && N >= 16384 && M <= 65535) {   // one wave per row would leave most of the chip idle
        size_t chunks = (2 * target + M - 1) / M;
        const size_t max_chunks = N / 4096;
        if (chunks > max_chunks) chunks = max_chunks;
        if (chunks >= 2) {
            const size_t chunk_len = ((N + chunks - 1) / chunks + 3) / 4 * 4;
            chunks = (N + chunk_len - 1) / chunk_len;
            np::Scratch partial;
            if (int rc = partial.alloc(M * chunks * sizeof(float))) return rc;
            sgemv_chunks_kernel<<<dim3((unsigned)chunks, (unsigned)M), 256, 0, np::stream()>>>(
                A, x, (float *)partial.ptr, (unsigned)N, (unsigned)chunk_len);
            NP_LAUNCH_CHECK("sgemv_chunks_kernel");
            return np_reduce_axis(NP_SUM, (const float *)partial.ptr, M, chunks, 1, y, 0);
        }
    }
    const int vec = (N % 4 == 0) && aligned16(A) && aligned16(x);
    sgemv_kernel<<<(unsigned)((M + 3) / 4), 256, 0, np::stream()>>>(A, x, y, (unsigned)M,
                                                                  (unsigned)N, vec);
    NP_LAUNCH_CHECK("sgemv_kernel");
    return NP_OK;
}

}  // extern "C"
