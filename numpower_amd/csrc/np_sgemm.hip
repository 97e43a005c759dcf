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
#include "np_internal.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

struct GemmArgs {
    const float *A, *B;
    float *C;
    unsigned M, N, K;
    unsigned lda, ldb, ldc;
    size_t stride_a, stride_b, stride_c;   // batch strides (elements)
    unsigned tiles_m, tiles_n;
    unsigned swizzle;                       // 0 = row-major tile order, else XCD-aware grouping
};

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
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    ra[r][e] = (gm < g.M && gk + e < g.K) ? A[(size_t)gm * g.lda + gk + e] : 0.0f;
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
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    rb[r][e] = (gk < g.K && gn + e < g.N) ? B[(size_t)gk * g.ldb + gn + e] : 0.0f;
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
        for (unsigned k = lane; k < N; k += 64) acc = fmaf(a[k], x[k], acc);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) y[row] = acc;
}

int g_variant = 0;

inline bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

template <int BM, int BN, int BK, int MINW>
int launch_sgemm_tile(GemmArgs g, unsigned batch, bool vec) {
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (g.N + BN - 1) / BN;
    const bool edge = (g.M % BM) || (g.N % BN) || (g.K % BK);
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

int launch_sgemm(size_t batch, size_t M, size_t N, size_t K, const float *A, size_t sa,
                 const float *B, size_t sb, float *C, size_t sc) {
    if (M > 0x7fffffffu || N > 0x7fffffffu || K > 0x7fffffffu || batch > 65535)
        return np::fail(NP_ERR_INVALID, "np_sgemm: dimension too large");
    GemmArgs g;
    g.A = A; g.B = B; g.C = C;
    g.M = (unsigned)M; g.N = (unsigned)N; g.K = (unsigned)K;
    g.lda = (unsigned)K; g.ldb = (unsigned)N; g.ldc = (unsigned)N;
    g.stride_a = sa; g.stride_b = sb; g.stride_c = sc;
    g.tiles_m = g.tiles_n = 0;
    const bool vec = (K % 4 == 0) && (N % 4 == 0) && aligned16(A) && aligned16(B) &&
                     (sa % 4 == 0) && (sb % 4 == 0);
    // variant = tile_code + 10 * swizzle_group ; 0 = default
    const int tile = g_variant % 10;
    g.swizzle = (unsigned)(g_variant / 10);
    switch (tile) {
        case 1: return launch_sgemm_tile<128, 128, 16, 4>(g, (unsigned)batch, vec);
        case 2: return launch_sgemm_tile<128, 128, 32, 2>(g, (unsigned)batch, vec);
        case 3: return launch_sgemm_tile<128, 128, 16, 2>(g, (unsigned)batch, vec);
        case 4: return launch_sgemm_tile<64, 64, 16, 4>(g, (unsigned)batch, vec);
        default: break;
    }
    // heuristic: small outputs get 64x64 tiles so that more than a handful of CUs have work
    const size_t big_tiles = ((M + 127) / 128) * ((N + 127) / 128) * batch;
    if (big_tiles < (size_t)np::num_cus() && (M > 64 || N > 64))
        return launch_sgemm_tile<64, 64, 16, 4>(g, (unsigned)batch, vec);
    if (M <= 64 && N <= 64) return launch_sgemm_tile<64, 64, 16, 4>(g, (unsigned)batch, vec);
    return launch_sgemm_tile<128, 128, 16, 4>(g, (unsigned)batch, vec);
}

}  // namespace

extern "C" {

int np_sgemm_set_variant(int variant) {
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
    return launch_sgemm(batch, M, N, K, A, stride_a, B, stride_b, C, stride_c);
}

int np_sgemv(size_t M, size_t N, const float *A, const float *x, float *y) {
    if (M == 0) return NP_OK;
    if (!A || !x || !y) return np::fail(NP_ERR_INVALID, "np_sgemv: null pointer");
    if (M > 0x7fffffffu || N > 0x7fffffffu)
        return np::fail(NP_ERR_INVALID, "np_sgemv: dimension too large");
    if (int rc = np::ensure_init()) return rc;
    const int vec = (N % 4 == 0) && aligned16(A) && aligned16(x);
    sgemv_kernel<<<(unsigned)((M + 3) / 4), 256, 0, np::stream()>>>(A, x, y, (unsigned)M,
                                                                  (unsigned)N, vec);
    NP_LAUNCH_CHECK("sgemv_kernel");
    return NP_OK;
}

}  // extern "C"
