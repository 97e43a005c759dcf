// Elementwise kernels of libnp_hip.so: binary ops with fused scalar/row/column broadcast, the
// unary float_* family, and fill.  HBM-bound streaming kernels: 16 B per lane per access
// (global_load/store_dwordx4), UNROLL independent accesses in flight per lane, an uncapped grid
// (every lane moves UNROLL float4 and retires), non-temporal loads and stores.  No LDS, no MFMA:
// nothing here has reuse.
//
// Reference behaviour restated (file:line relative to the reference tree):
//   binary ops      src/ndmath/arithmetics.c:160-926  (+ CUDA kernels cuda_math.cu:593-633)
//   broadcast       src/ndarray.c:1172-1294           (materialised there, index math here)
//   unary ops       src/ndmath/double_math.c:9-265    (+ CUDA kernels cuda_math.cu:207-585)
//   fill            cuda_math.cu:829,912
#include <math.h>

#include "np_internal.h"
#include "np_elementwise_ops.h"

// Bit-level parity with the reference's CPU results needs every multiply, add and divide rounded
// on its own: no implicit FMA contraction anywhere in this file (the two places where the
// reference's build DOES fuse — mod's AVX body and the rsqrt Newton step — say so with an
// explicit __fmaf_rn).  `/` and sqrtf are IEEE-correct under hipcc's defaults; the __f*_rn
// helpers are plain operators in this ROCm, kept only as documentation of intent.
#pragma clang fp contract(off)

namespace {


// Row length of a broadcast whose float4s may straddle rows (cols % 4 != 0): `cols` = 0 means "not
// ragged" (the fast path below), else the per-element (row, col) come from fast_div.
template <typename I>
struct Ragged {
    I cols;
    unsigned m, s1, s2;
    // A ROW operand too long to stay cached (X + r with 7 x 10^7: each of the 7 result rows re-reads the 40 MB vector,
    // 4.8 TB/s where X + col runs 6.3): rowblock_rows = rows of the result switches the work order from row-major to
    // column blocks, XCD by XCD — workgroup w (dispatched to XCD w % 8) takes 256 float4 slots of column block
    // (w / 8 / rows) * 8 + w % 8 in row (w / 8) % rows: the workgroups that need the same 4 KiB piece of the vector follow
    // each other on ONE XCD and find it in that XCD's L2 (sharing it across consecutive workgroups, i.e. across the eight
    // private L2s, changed nothing: profiles/r04/misc_sweep.log).  wspace = rows x column blocks (rounded up to 8) x 256
    // work items.  0 = off (the plain order).
    unsigned rowblock_rows;
    I wspace;
};

template <typename I>
__device__ __forceinline__ I ragged_row(const Ragged<I> &r, I e) {
    if constexpr (sizeof(I) == 4) return (I)fast_div((unsigned)e, r.m, r.s1, r.s2);
    return e / r.cols;
}

// Operand fetch for the vector path: `v` is the float4 index into the rows x cols output,
// cols4 = cols / 4.  With cols % 4 == 0 a float4 never straddles a row: ROW is one float4 load, COL
// one scalar.  Otherwise (rg.cols != 0, a uniform branch) each of the four elements gets its own
// row / column index; the row / column vector itself is cache-resident.
template <int KIND, bool NT, typename I>
__device__ __forceinline__ v4f fetch4(const float *p, I v, I cols4, float splat, const Ragged<I> &rg) {
    if constexpr (KIND == NP_FULL) return ld4<NT>(p + (size_t)v * 4);
    if constexpr (KIND == NP_SCALAR) return v4f{splat, splat, splat, splat};
    if constexpr (KIND == NP_ROW) {
        if (rg.cols == 0) return *(const v4f_u *)(p + (size_t)(v % cols4) * 4);
        // ragged: ONE division per float4 (the position of its first element), then a dword-aligned float4 load when the
        // four elements stay in one row — all but one float4 in cols / 4 — and a walk with wrap-around for the others
        const I e0 = v * 4;
        I c = e0 - ragged_row(rg, e0) * rg.cols;
        if (c + 3 < rg.cols) return *(const v4f_u *)(p + (size_t)c);
        v4f r;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            r[k] = p[(size_t)c];
            c = c + 1 == rg.cols ? (I)0 : c + 1;
        }
        return r;
    }
    if constexpr (KIND == NP_COL) {
        if (rg.cols == 0) {
            const float s = p[(size_t)(v / cols4)];
            return v4f{s, s, s, s};
        }
        // (one division per float4 + a walk, as for ROW above, measured 2 % slower here than four divisions: the column
        // operand needs no second load in the common case either way — tools/ragged_ab.py)
        v4f r;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = p[(size_t)ragged_row(rg, (I)(v * 4 + k))];
        return r;
    }
    return v4f{0, 0, 0, 0};
}

// ------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------

// Vector path: each lane handles UNROLL float4 per trip, spaced one grid apart so that every
// wave-level access is a contiguous 1 KiB segment.
template <int OP, int AK, int BK, bool QUIRK, int UNROLL, bool NT, typename I, bool POWREG = false>
__global__ __launch_bounds__(256) void binary_vec_kernel(const float *__restrict__ a,
                                                         const float *__restrict__ b,
                                                         float *__restrict__ out, I nvec, I cols4,
                                                         I tail_start, I n, I body_end, float ha,
                                                         float hb, Ragged<I> rg) {
    const I stride = (I)gridDim.x * blockDim.x;
    const I tid = (I)blockIdx.x * blockDim.x + threadIdx.x;
    // scalar operands: device pointer, or (null pointer) a value handed over by the host
    float sa = 0.0f, sb = 0.0f;
    if constexpr (AK == NP_SCALAR) sa = a ? a[0] : ha;
    if constexpr (BK == NP_SCALAR) sb = b ? b[0] : hb;
    // pow: every wave stages its own copy of the log2 table (512 B) in LDS — no workgroup barrier: a wave's
    // LDS operations complete in order, so its reads below see its own writes
    constexpr bool LDSTAB = OP == NP_POW && !POWREG;
    __shared__ PowLogEntry pow_tab_lds[LDSTAB ? 4 : 1][LDSTAB ? 32 : 1];
    PowTabLds pow_tab = (PowTabLds)&pow_tab_lds[LDSTAB ? (threadIdx.x >> 6) : 0][0];
    PowTabRegs pow_regs{0, 0};
    if constexpr (LDSTAB) {
        const unsigned lane = threadIdx.x & 63u;
        if (lane < 32u) pow_tab_lds[threadIdx.x >> 6][lane] = kPowLogTab[lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr (POWREG) pow_regs.load(threadIdx.x & 63u);

    // POWREG: the loop and the arithmetic run with the whole wave active (the table lives in its lanes); lanes past
    // the end compute 1^1 and store nothing
    constexpr bool ROWBLOCK = (AK == NP_ROW || BK == NP_ROW) && sizeof(I) == 4 && OP != NP_POW;
    const I limit = ROWBLOCK && rg.rowblock_rows ? rg.wspace : nvec;
    // work item -> float4 slot (see Ragged::rowblock_rows); ok = the slot exists
    auto slot_of = [&](I w, bool &ok) -> I {
        if constexpr (ROWBLOCK) {
            if (rg.rowblock_rows) {   // uniform
                const unsigned wg = __builtin_amdgcn_readfirstlane((unsigned)w >> 8);   // a workgroup's 256 items share it
                const unsigned j = wg >> 3, grp = j / rg.rowblock_rows, row = j - grp * rg.rowblock_rows;
                const unsigned c = (grp * 8u + (wg & 7u)) * 256u + ((unsigned)w & 255u);
                ok = w < limit && c < (unsigned)cols4;
                return (I)(row * (unsigned)cols4 + c);
            }
        }
        ok = w < nvec;
        return w;
    };
    auto more = [&](I base) { return POWREG ? __builtin_amdgcn_ballot_w64(base < nvec) != 0ull : base < limit; };
    for (I base = tid; more(base); base += stride * UNROLL) {
        v4f va[UNROLL], vb[UNROLL];
        I slot[UNROLL];
        bool live[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            slot[u] = slot_of(base + (I)u * stride, live[u]);
            const I v = slot[u];
            if constexpr (POWREG) va[u] = vb[u] = v4f{1.0f, 1.0f, 1.0f, 1.0f};
            if (live[u]) {
                va[u] = fetch4<AK, NT, I>(a, v, cols4, sa, rg);
                vb[u] = fetch4<BK, NT, I>(b, v, cols4, sb, rg);
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const I v = slot[u];
            if constexpr (POWREG) {
                const float px[4] = {va[u][0], va[u][1], va[u][2], va[u][3]};
                const float py[4] = {vb[u][0], vb[u][1], vb[u][2], vb[u][3]};
                float pr[4];
                pow_n<4>(px, py, pr, pow_regs);
                if (live[u]) st4<NT>(out + (size_t)v * 4, v4f{pr[0], pr[1], pr[2], pr[3]});
            } else if (live[u]) {
                v4f r;
                const I e = v * 4;
                if constexpr (OP == NP_POW) {
                    const float px[4] = {va[u][0], va[u][1], va[u][2], va[u][3]};
                    const float py[4] = {vb[u][0], vb[u][1], vb[u][2], vb[u][3]};
                    float pr[4];
                    pow_n<4>(px, py, pr, pow_tab);
                    r = v4f{pr[0], pr[1], pr[2], pr[3]};
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        r[k] = binary_apply<OP, QUIRK>(va[u][k], vb[u][k], (e + k) < body_end);
                }
                st4<NT>(out + (size_t)v * 4, r);
            }
        }
    }
    // ragged tail (n % 4 elements): FULL/SCALAR operand kinds, or a ragged broadcast
    if (blockIdx.x == 0) {
        const I i = tail_start + threadIdx.x;
        if (i < n) {
            auto at = [&](const float *p, int kind, float splat) -> float {
                if (kind == NP_SCALAR) return splat;
                if (kind == NP_FULL) return p[i];
                const I row = ragged_row(rg, i);
                return kind == NP_ROW ? p[(size_t)(i - row * rg.cols)] : p[(size_t)row];
            };
            out[i] = binary_apply<OP, QUIRK>(at(a, AK, sa), at(b, BK, sb), i < body_end);
        }
    }
}

// Scalar path: any operand kinds, any shape/alignment.
template <int OP, bool QUIRK, typename I>
__global__ __launch_bounds__(256) void binary_scalar_kernel(const float *__restrict__ a, int ak,
                                                            const float *__restrict__ b, int bk,
                                                            float *__restrict__ out, I n, I cols,
                                                            I body_end, float ha, float hb) {
    const I stride = (I)gridDim.x * blockDim.x;
    for (I i = (I)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float x, y;
        switch (ak) {
            case NP_FULL: x = a[i]; break;
            case NP_SCALAR: x = a ? a[0] : ha; break;
            case NP_ROW: x = a[i % cols]; break;
            default: x = a[i / cols]; break;
        }
        switch (bk) {
            case NP_FULL: y = b[i]; break;
            case NP_SCALAR: y = b ? b[0] : hb; break;
            case NP_ROW: y = b[i % cols]; break;
            default: y = b[i / cols]; break;
        }
        out[i] = binary_apply<OP, QUIRK>(x, y, i < body_end);
    }
}

template <int OP, int UNROLL, bool NT, typename I>
__global__ __launch_bounds__(256) void unary_vec_kernel(const float *in, float *out, I nvec,
                                                        I tail_start, I n, float p0, float p1) {
    const I stride = (I)gridDim.x * blockDim.x;
    const I tid = (I)blockIdx.x * blockDim.x + threadIdx.x;
    for (I base = tid; base < nvec; base += stride * UNROLL) {
        v4f vx[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const I v = base + (I)u * stride;
            if (v < nvec) vx[u] = ld4<NT>(in + (size_t)v * 4);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const I v = base + (I)u * stride;
            if (v < nvec) {
                v4f r;
#pragma unroll
                for (int k = 0; k < 4; ++k) r[k] = unary_apply<OP>(vx[u][k], p0, p1);
                st4<NT>(out + (size_t)v * 4, r);
            }
        }
    }
    if (blockIdx.x == 0) {
        const I i = tail_start + threadIdx.x;
        if (i < n) out[i] = unary_apply<OP>(in[i], p0, p1);
    }
}

template <int OP, typename I>
__global__ __launch_bounds__(256) void unary_scalar_kernel(const float *in, float *out, I n,
                                                           float p0, float p1) {
    const I stride = (I)gridDim.x * blockDim.x;
    for (I i = (I)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = unary_apply<OP>(in[i], p0, p1);
}

template <typename I>
__global__ __launch_bounds__(256) void fill_kernel(float *__restrict__ out, float value, I n,
                                                   I head, I nvec) {
    // head = elements before the first 16-byte boundary, then nvec float4, then the rest
    const I stride = (I)gridDim.x * blockDim.x;
    const I tid = (I)blockIdx.x * blockDim.x + threadIdx.x;
    const v4f v{value, value, value, value};
    for (I i = tid; i < nvec; i += stride) __builtin_nontemporal_store(v, (v4f *)(out + head + (size_t)i * 4));
    if (blockIdx.x == 0) {
        if (threadIdx.x < head) out[threadIdx.x] = value;
        const I t = head + nvec * 4 + threadIdx.x;
        if (t < n) out[t] = value;
    }
}

// ------------------------------------------------------------------------------------------
// launch configuration
// ------------------------------------------------------------------------------------------

// Plain device-to-device copy of 4-byte words (np::device_copy): the same float4 / non-temporal stream
// as the unary kernels; hipMemcpyAsync's blit reaches 5.1-5.3 TB/s on large buffers, this 6.0-6.4.
template <typename I>
__global__ __launch_bounds__(256) void copy_vec_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                       I nvec, I n) {
    const I stride = (I)gridDim.x * blockDim.x;
    const I tid = (I)blockIdx.x * blockDim.x + threadIdx.x;
    for (I base = tid; base < nvec; base += 2 * stride) {
        const I v1 = base + stride;
        const v4f x0 = ld4<true>(in + (size_t)base * 4);
        v4f x1{0, 0, 0, 0};
        if (v1 < nvec) x1 = ld4<true>(in + (size_t)v1 * 4);
        st4<true>(out + (size_t)base * 4, x0);
        if (v1 < nvec) st4<true>(out + (size_t)v1 * 4, x1);
    }
    if (blockIdx.x == 0) {
        const I t = nvec * 4 + threadIdx.x;
        if (t < n) out[t] = in[t];
    }
}

// out[i][j] = 0 + a[i] * b[j]: cblas_sger on a zeroed matrix (linalg.c:741-742) — the "0 +" matters
// only for the sign of zero products (-0 + +0 = +0).  Write-bound: 4 B/elem.
template <bool VEC, typename I>
__global__ __launch_bounds__(256) void outer_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                    float *__restrict__ out, I rows, I cols) {
    const I stride = (I)gridDim.x * blockDim.x;
    const I tid = (I)blockIdx.x * blockDim.x + threadIdx.x;
    if constexpr (VEC) {
        const I cols4 = cols / 4, nvec = rows * cols4;
        for (I v = tid; v < nvec; v += stride) {
            const I i = v / cols4, j4 = v - i * cols4;
            const float x = a[i];
            const v4f y = *(const v4f *)(b + (size_t)j4 * 4);
            v4f r;
#pragma unroll
            for (int k = 0; k < 4; ++k) r[k] = __fmaf_rn(x, y[k], 0.0f);
            __builtin_nontemporal_store(r, (v4f *)(out + (size_t)v * 4));
        }
    } else {
        const I n = rows * cols;
        for (I e = tid; e < n; e += stride) {
            const I i = e / cols;
            out[e] = __fmaf_rn(a[i], b[e - i * cols], 0.0f);
        }
    }
}

int g_variant = 0;   // see np_elementwise_set_variant
constexpr bool kColsWideDefault = false;      // fused_chain_cols_kernel: column-block width / residency (A/B: tools/fused_cols_ab.py)
constexpr size_t kColsWgPerCuDefault = 4;   // 92.5 us vs 94-108 at 8 (profiles/r02/fused_cols_ab.log)

struct LaunchCfg {
    int unroll;          // float4 per lane per trip
    int blocks_per_cu;   // grid = min(needed, CUs * blocks_per_cu)
    bool nt;             // non-temporal loads/stores
};

// Default = what was measured fastest on MI355X for 1e8-float streams (tools/explore/add_bw.hip,
// tools/ew_ab.py; profiles/r01/add_bw_explore.log, profiles/r01/ew_ab.log): no grid cap — every
// lane moves UNROLL = 2 float4 and retires, ~49k workgroups keep the dispatcher ahead of the
// memory system — with non-temporal loads and stores.  Capping the grid at a few workgroups per
// CU and grid-striding was 3-8 % slower, cached (non-nt) accesses 5-10 % slower, UNROLL 1/4/8
// 1-8 % slower than 2 on random data (add 6.28, exp 6.60, add+row 6.47 TB/s at UNROLL 2).
// variant = unroll_code + 10*bpc_code + 100*nt ; 0 = default
LaunchCfg cfg_from_variant(int variant) {
    LaunchCfg c{2, 0, true};
    if (variant <= 0) return c;
    const int u = variant % 10, b = (variant / 10) % 10, nt = (variant / 100) % 10;
    if (u == 1) c.unroll = 1;
    if (u == 2) c.unroll = 2;
    if (u == 4) c.unroll = 4;
    if (u == 8) c.unroll = 8;
    c.blocks_per_cu = b * 2;   // 0 = uncapped, 1..9 -> 2..18 workgroups per CU
    c.nt = (nt != 0);
    return c;
}

unsigned grid_for(size_t work_items, int per_thread, int blocks_per_cu) {
    const size_t threads = (work_items + per_thread - 1) / per_thread;
    size_t blocks = (threads + 255) / 256;
    if (blocks_per_cu > 0) {
        const size_t cap = (size_t)np::num_cus() * blocks_per_cu;
        if (blocks > cap) blocks = cap;
    }
    if (blocks > 0x7fffffffu) blocks = 0x7fffffffu;
    if (blocks < 1) blocks = 1;
    // a lane's UNROLL accesses are one grid apart: keep that distance off the powers of two (2^27 elements would
    // put them exactly 256 MiB apart, on the same HBM channel — np::capped_grid has the measurement)
    if (blocks >= 64) blocks |= 1;
    return (unsigned)blocks;
}

inline bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

// FULL (op) ROW with whole float4 columns (cols % 4 == 0, at least one workgroup of them): a lane owns ONE float4 column of a
// block of RPT rows — the row operand's four values are loaded once and stay in registers, the RPT loads of the full operand
// are independent and in flight together, one multiply-high division per lane.  binary_vec_kernel's flat walk pays a 32-bit
// modulo and a second (cache-resident) load per float4 for the same result (profiles/r04/bcast2d_ab.log).  Same binary_apply, same bits.  blockIdx.x = column block (fastest: neighbouring workgroups stream
// neighbouring memory), blockIdx.y strides over the row blocks.
template <int OP, bool QUIRK, bool ROW_IS_A, int RPT>
__global__ __launch_bounds__(256) void binary_rows2d_kernel(const float *__restrict__ full, const float *__restrict__ row,
                                                            float *__restrict__ out, unsigned rows, unsigned cols, unsigned body_end,
                                                            unsigned items, unsigned div_m, unsigned div_s1, unsigned div_s2) {
    // work item = (row block, float4 column), column fastest, no padding of the last column block: ONE multiply-high division
    // per lane (for RPT float4s) instead of a modulo per float4
    const unsigned id = blockIdx.x * 256u + threadIdx.x;
    if (id >= items) return;
    const unsigned cols4 = cols / 4, rb = fast_div(id, div_m, div_s1, div_s2), c0 = (id - rb * cols4) * 4, r0 = rb * RPT;
    const v4f rv = *(const v4f_u *)(row + c0);
    v4f x[RPT];
#pragma unroll
    for (int u = 0; u < RPT; ++u)
        if (r0 + u < rows) x[u] = ld4<true>(full + (size_t)(r0 + u) * cols + c0);
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        if (r0 + u < rows) {
            const unsigned e = (r0 + u) * cols + c0;
            v4f r;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                r[k] = ROW_IS_A ? binary_apply<OP, QUIRK>(rv[k], x[u][k], (e + k) < body_end) : binary_apply<OP, QUIRK>(x[u][k], rv[k], (e + k) < body_end);
            st4<true>(out + (size_t)(r0 + u) * cols + c0, r);
        }
    }
}

// ---- binary dispatch ----

template <int OP, int AK, int BK, bool QUIRK, typename I>
int launch_binary_vec(const float *a, const float *b, float *out, size_t n, size_t cols,
                      size_t body_end, float ha, float hb) {
    const LaunchCfg c = cfg_from_variant(g_variant);
    const I nvec = (I)(n / 4), cols4 = (I)(cols / 4), tail = (I)(n / 4 * 4);
    unsigned grid = grid_for(n / 4 + 1, c.unroll, c.blocks_per_cu);
    hipStream_t s = np::stream();
    // FULL (op) ROW on whole float4 columns: the 2-D form (binary_rows2d_kernel; variant 8100: off).  Rows shorter than one
    // workgroup of float4s and vectors too long to stay cached (the column-block order below) keep the flat walk.
    if constexpr (((AK == NP_ROW && BK == NP_FULL) || (AK == NP_FULL && BK == NP_ROW)) && OP != NP_POW && sizeof(I) == 4) {
        // (measured in alternation, profiles/r04/bcast2d_ab.log: row (op) X +8-10 % everywhere; X (op) row +-1 % on rows of a few
        // thousand floats — the flat walk's modulo and second load are hidden there — and +2-4 % on rows of 10^5 and more)
        if (cols % 32 == 0 && cols >= 1024 && (AK == NP_ROW || cols >= 65536) && cols < (size_t(1) << 22) && n / cols >= 2 && g_variant == 0 &&
            !np::g_bcast2d_off) {
            constexpr int RPT = 2;   // (as binary_vec_kernel's two float4 per lane: fatter lanes stream slower on this machine, 8 rows per lane lost 4 %)
            const size_t rows = n / cols, row_blocks = (rows + RPT - 1) / RPT, items = row_blocks * (cols / 4);   // (< 2^31: n is)
            unsigned dm, d1, d2;
            fast_div_magic(cols / 4, dm, d1, d2);
            const unsigned grid2 = (unsigned)((items + 255) / 256);
            if constexpr (AK == NP_ROW)
                binary_rows2d_kernel<OP, QUIRK, true, RPT><<<grid2, 256, 0, s>>>(b, a, out, (unsigned)rows, (unsigned)cols, (unsigned)body_end,
                                                                                   (unsigned)items, dm, d1, d2);
            else
                binary_rows2d_kernel<OP, QUIRK, false, RPT><<<grid2, 256, 0, s>>>(a, b, out, (unsigned)rows, (unsigned)cols, (unsigned)body_end,
                                                                                    (unsigned)items, dm, d1, d2);
            NP_LAUNCH_CHECK("binary_rows2d_kernel");
            return NP_OK;
        }
    }
    Ragged<I> rg{0, 0, 0, 0, 0, 0};
    // a ROW operand of >= 16 MB under a result of several rows: column-block order (Ragged::rowblock_rows; variant 8000: off).
    // 7 x 10^7 4.79 -> 5.71 TB/s, 3 x 3*10^7 4.84 -> 5.73; an 8 MB vector (40 x 2*10^6) is served by the Infinity Cache either
    // way and loses 2 % to the index arithmetic (tools/rowblock_ab.py, profiles/r04/rowblock_ab.log)
    if ((AK == NP_ROW || BK == NP_ROW) && OP != NP_POW && sizeof(I) == 4 && cols % 4 == 0 && cols >= (size_t(1) << 22) && n / cols >= 2 &&
        n / cols < (size_t(1) << 16) && g_variant != 8000) {
        const size_t rows = n / cols, col_blocks = ((cols / 4 + 255) / 256 + 7) / 8 * 8;
        if (rows * col_blocks * 256 < (size_t(1) << 31)) {
            rg.rowblock_rows = (unsigned)rows;
            rg.wspace = (I)(rows * col_blocks * 256);
            grid = grid_for(rows * col_blocks * 256, c.unroll, c.blocks_per_cu);
        }
    }
    if ((AK == NP_ROW || AK == NP_COL || BK == NP_ROW || BK == NP_COL) && cols % 4 != 0) {
        rg.cols = (I)cols;
        if (cols > 1 && cols <= 0xffffffffull) {   // fast_div constants (cols == 1: q = n)
            unsigned l = 0;
            while ((1ull << l) < cols) ++l;
            rg.m = (unsigned)((((1ull << 32) * ((1ull << l) - cols)) / cols) + 1);
            rg.s1 = 1;
            rg.s2 = l - 1;
        }
    }
#define NP_BV(U, NT)                                                                       \
    binary_vec_kernel<OP, AK, BK, QUIRK, U, NT, I><<<grid, 256, 0, s>>>(a, b, out, nvec,   \
                                                                         cols4, tail, (I)n, \
                                                                         (I)body_end, ha, hb, rg)
    // only the measured-useful variants are instantiated (the full unroll x nt sweep lives in
    // tools/explore/add_bw.hip): UNROLL 2 (default) and 4, non-temporal
    if (OP == NP_POW && g_variant != 9000 && c.unroll == 2) {   // the log2 table in registers (9000: the per-wave LDS copy, tools/pow_grid_ab.py)
        if constexpr (OP == NP_POW)
            binary_vec_kernel<OP, AK, BK, QUIRK, 2, true, I, true><<<grid, 256, 0, s>>>(a, b, out, nvec, cols4, tail, (I)n,
                                                                                       (I)body_end, ha, hb, rg);
    } else if (c.unroll == 4)
        NP_BV(4, true);
    else if (c.unroll == 1)
        NP_BV(1, true);
    else
        NP_BV(2, true);
#undef NP_BV
    NP_LAUNCH_CHECK("binary_vec_kernel");
    return NP_OK;
}

template <int OP, bool QUIRK, typename I>
int launch_binary_scalar(const float *a, int ak, const float *b, int bk, float *out, size_t n,
                         size_t cols, size_t body_end, float ha, float hb) {
    const unsigned grid = grid_for(n, 4, 16);
    binary_scalar_kernel<OP, QUIRK, I>
        <<<grid, 256, 0, np::stream()>>>(a, ak, b, bk, out, (I)n, (I)cols, (I)body_end, ha, hb);
    NP_LAUNCH_CHECK("binary_scalar_kernel");
    return NP_OK;
}

template <int OP, bool QUIRK, typename I>
int dispatch_binary_kinds(const float *a, int ak, const float *b, int bk, float *out, size_t rows,
                          size_t cols, size_t body_end, float ha, float hb) {
    const size_t n = rows * cols;
    // vector path: ROW/COL operands need cols % 4 == 0 so that a float4 never straddles a row;
    // pointers may have any (4-byte) alignment.
    const bool bcast = (ak == NP_ROW || ak == NP_COL || bk == NP_ROW || bk == NP_COL);
    // no alignment requirement (ld4 / st4 are dword-aligned float4 accesses) and no row-length
    // requirement (ragged broadcasts index per element, see fetch4)
    const bool vec = n >= 4;
    (void)bcast;
    if (vec) {
#define NP_BK(AK_, BK_)                                                                     \
    if (ak == AK_ && bk == BK_)                                                             \
        return launch_binary_vec<OP, AK_, BK_, QUIRK, I>(a, b, out, n, cols, body_end, ha, hb)
        NP_BK(NP_FULL, NP_FULL);
        NP_BK(NP_FULL, NP_SCALAR);
        NP_BK(NP_SCALAR, NP_FULL);
        NP_BK(NP_FULL, NP_ROW);
        NP_BK(NP_ROW, NP_FULL);
        NP_BK(NP_FULL, NP_COL);
        NP_BK(NP_COL, NP_FULL);
#undef NP_BK
    }
    return launch_binary_scalar<OP, QUIRK, I>(a, ak, b, bk, out, n, cols, body_end, ha, hb);
}

template <int OP, bool QUIRK>
int dispatch_binary_index(const float *a, int ak, const float *b, int bk, float *out, size_t rows,
                          size_t cols, size_t body_end, float ha, float hb) {
    const size_t n = rows * cols;
    if (n < (size_t(1) << 31)) {
        if (body_end > n) body_end = n;
        return dispatch_binary_kinds<OP, QUIRK, uint32_t>(a, ak, b, bk, out, rows, cols, body_end, ha, hb);
    }
    return dispatch_binary_kinds<OP, QUIRK, uint64_t>(a, ak, b, bk, out, rows, cols, body_end, ha, hb);
}

template <int OP>
int dispatch_binary_quirk(const float *a, int ak, const float *b, int bk, float *out, size_t rows,
                          size_t cols, unsigned flags, size_t body_end, float ha, float hb) {
    constexpr bool has_quirk = (OP == NP_MULTIPLY || OP == NP_MOD || OP == NP_EQUAL || OP == NP_NOT_EQUAL);
    if constexpr (has_quirk) {
        if (flags & NP_QUIRK_AVX_BODY)
            return dispatch_binary_index<OP, true>(a, ak, b, bk, out, rows, cols, body_end, ha, hb);
    }
    return dispatch_binary_index<OP, false>(a, ak, b, bk, out, rows, cols, 0, ha, hb);
}

// ---- unary dispatch ----

template <int OP, typename I>
int launch_unary(const float *in, float *out, size_t n, float p0, float p1) {
    hipStream_t s = np::stream();
    if (n >= 4) {   // any alignment: see ld4 / st4
        const LaunchCfg c = cfg_from_variant(g_variant);
        const I nvec = (I)(n / 4), tail = (I)(n / 4 * 4);
        const unsigned grid = grid_for(n / 4 + 1, c.unroll, c.blocks_per_cu);
#define NP_UV(U, NT) \
    unary_vec_kernel<OP, U, NT, I><<<grid, 256, 0, s>>>(in, out, nvec, tail, (I)n, p0, p1)
        if (c.unroll == 4)
            NP_UV(4, true);
        else if (c.unroll == 1)
            NP_UV(1, true);
        else
            NP_UV(2, true);
#undef NP_UV
        NP_LAUNCH_CHECK("unary_vec_kernel");
        return NP_OK;
    }
    const unsigned grid = grid_for(n, 4, 16);
    unary_scalar_kernel<OP, I><<<grid, 256, 0, s>>>(in, out, (I)n, p0, p1);
    NP_LAUNCH_CHECK("unary_scalar_kernel");
    return NP_OK;
}

template <int OP>
int dispatch_unary(const float *in, float *out, size_t n, float p0, float p1) {
    if (n < (size_t(1) << 31)) return launch_unary<OP, uint32_t>(in, out, n, p0, p1);
    return launch_unary<OP, uint64_t>(in, out, n, p0, p1);
}

// ------------------------------------------------------------------------------------------
// fused elementwise chains
// ------------------------------------------------------------------------------------------
//
// acc = in[0]; for each op: acc = f(acc) | acc (op) in[k] | in[k] (op) acc; out = acc — ONE pass
// over HBM for a whole expression like exp(a) * b + 2 instead of one pass (and one temporary) per
// PHP-level op.  Every step goes through the same binary_apply / unary_apply bodies as the
// stand-alone kernels (contraction is off in this file), so a fused chain is bit-identical to the
// unfused sequence.  Inputs are full arrays of n elements or host scalars.
constexpr int FUSED_MAX_OPS = 12;
constexpr int FUSED_MAX_IN = 6;

// What the kernel interprets (built on the host from np_fused_op): operands are resolved to
// pointers / values up front so a step costs one uniform load of its descriptor.
enum { FUSED_SRC_SCALAR = 0, FUSED_SRC_STREAM = 1, FUSED_SRC_INPUT0 = 2 };
// how a streamed operand is indexed by the flat element index e of the rows x cols result
enum { FUSED_IDX_FULL = 0, FUSED_IDX_ROW = 1 /* e % cols */, FUSED_IDX_COL = 2 /* e / cols */, FUSED_IDX_ZERO = 3 };
struct FusedStep {
    int kind, op, swap, quirk;
    float p0, p1, scalar;
    int src_kind;
    const float *prefetch;   // array whose load is started when this step consumes its streamed operand
    size_t body_end;
    int prefetch_idx;        // FUSED_IDX_* of `prefetch`
    int pad;
};
struct FusedArgs {
    int n_ops;
    float scalar0;
    const float *in0;              // null -> scalar0
    const float *first_prefetch;   // first streamed operand of the chain (null: none)
    size_t cols;                   // row length of the result; 0 when no operand is broadcast
    int first_prefetch_idx;
    int sink;                      // -1: store the chain value; NP_SUM / NP_PROD / NP_MIN / NP_MAX: reduce it
    unsigned *ticket;              // full reduction on a small grid: the last workgroup folds the partials (np_internal.h) ...
    float *result;                 // ... into this device float; null ticket: np::fold_partials does, in a second launch
    unsigned div_m, div_s1, div_s2;   // e / cols for 32-bit e without a division (see fast_div)
    FusedStep ops[FUSED_MAX_OPS];
};
// The kernel indexes ops[] with a run-time k.  On a by-value kernel parameter that makes the compiler
// copy the whole struct to scratch (private memory) first; reading it through the kernarg segment
// pointer (constant address space) keeps every descriptor fetch a scalar load.
typedef const __attribute__((address_space(4))) FusedArgs *FusedArgsK;

constexpr bool binary_has_quirk(int op) {
    return op == NP_MULTIPLY || op == NP_MOD || op == NP_EQUAL || op == NP_NOT_EQUAL;
}

// One op applied to all N elements a thread holds: the op switch below runs once per op per
// thread-trip (uniform, scalar branch), not once per element.
template <int OP, int N>
__device__ __forceinline__ void unary_all(float (&acc)[N], float p0, float p1) {
#pragma unroll
    for (int e = 0; e < N; ++e) acc[e] = unary_apply<OP>(acc[e], p0, p1);
}

// pow inside the chain interpreter goes through ONE out-of-line copy: inlined, its fp64 temporaries for 8
// elements took the full interpreter from 124 to 139 VGPRs (4 -> 3 waves per SIMD) for every chain
// that merely might contain a pow.
__device__ __attribute__((noinline)) float pow_outlined(float x, float y) { return fast_pow(x, y); }

template <int OP, int N>
__device__ __forceinline__ void binary_all(float (&acc)[N], const float (&oth)[N], bool swap, bool quirk,
                                           const bool (&body)[N]) {
    if constexpr (OP == NP_POW) {
#pragma unroll
        for (int e = 0; e < N; ++e) acc[e] = swap ? pow_outlined(oth[e], acc[e]) : pow_outlined(acc[e], oth[e]);
        return;
    }
    // `swap` and `quirk` are uniform: one branch each per step, not a select per element
    if (binary_has_quirk(OP) && quirk) {
        if (swap) {
#pragma unroll
            for (int e = 0; e < N; ++e) acc[e] = binary_apply<OP, binary_has_quirk(OP)>(oth[e], acc[e], body[e]);
        } else {
#pragma unroll
            for (int e = 0; e < N; ++e) acc[e] = binary_apply<OP, binary_has_quirk(OP)>(acc[e], oth[e], body[e]);
        }
    } else if (swap) {
#pragma unroll
        for (int e = 0; e < N; ++e) acc[e] = binary_apply<OP, false>(oth[e], acc[e], false);
    } else {
#pragma unroll
        for (int e = 0; e < N; ++e) acc[e] = binary_apply<OP, false>(acc[e], oth[e], false);
    }
}

// Register allocation is static: a kernel that CAN run powf or tanf on 8 elements is allocated for
// them even when the chain is exp -> multiply -> add, and pays in occupancy.  So there are two
// interpreters: LIGHT knows only the ops whose bodies are a handful of instructions (the common
// arithmetic chains), the full one knows everything; np_fused_chain picks per chain.
constexpr bool fused_light_unary(int op) {
    switch (op) {
        case NP_ABS: case NP_SQRT: case NP_EXP: case NP_EXP2: case NP_LOG: case NP_LOG2: case NP_LOG10:
        case NP_DEGREES: case NP_RADIANS: case NP_RINT: case NP_FIX: case NP_FLOOR: case NP_CEIL:
        case NP_TRUNC: case NP_NEGATE: case NP_SIGN: case NP_CLIP: case NP_ROUND: case NP_RSQRT:
        case NP_POSITIVE: case NP_RECIPROCAL: case NP_UNARY_SQUARE:
            return true;
        default:
            return false;
    }
}
// (adding the hand-written log1p / hyperbolic ops here cost the exp -> multiply -> add chain 3-5 %,
// adding expm1 / tanh / arc* 10 %: the set stays minimal)
constexpr bool fused_light_binary(int op) { return op != NP_MOD && op != NP_POW && op != NP_ARCTAN2; }

template <int N, bool LIGHT>
__device__ __forceinline__ void unary_dispatch(int op, float (&acc)[N], float p0, float p1) {
#define NP_UD(OP_)                                                                     \
    case OP_:                                                                          \
        if constexpr (!LIGHT || fused_light_unary(OP_)) unary_all<OP_, N>(acc, p0, p1); \
        break
    switch (op) {
        NP_UD(NP_ABS); NP_UD(NP_SQRT); NP_UD(NP_EXP); NP_UD(NP_EXP2); NP_UD(NP_EXPM1); NP_UD(NP_LOG);
        NP_UD(NP_LOG2); NP_UD(NP_LOG10); NP_UD(NP_LOG1P); NP_UD(NP_LOGB); NP_UD(NP_SIN); NP_UD(NP_COS);
        NP_UD(NP_TAN); NP_UD(NP_ARCSIN); NP_UD(NP_ARCCOS); NP_UD(NP_ARCTAN); NP_UD(NP_DEGREES);
        NP_UD(NP_RADIANS); NP_UD(NP_SINH); NP_UD(NP_COSH); NP_UD(NP_TANH); NP_UD(NP_ARCSINH);
        NP_UD(NP_ARCCOSH); NP_UD(NP_ARCTANH); NP_UD(NP_RINT); NP_UD(NP_FIX); NP_UD(NP_FLOOR);
        NP_UD(NP_CEIL); NP_UD(NP_TRUNC); NP_UD(NP_SINC); NP_UD(NP_NEGATE); NP_UD(NP_SIGN); NP_UD(NP_CLIP);
        NP_UD(NP_ROUND); NP_UD(NP_RSQRT); NP_UD(NP_POSITIVE); NP_UD(NP_RECIPROCAL); NP_UD(NP_UNARY_SQUARE);
        default: break;
    }
#undef NP_UD
}

template <int N, bool LIGHT>
__device__ __forceinline__ void binary_dispatch(int op, float (&acc)[N], const float (&oth)[N], bool swap,
                                                bool quirk, const bool (&body)[N]) {
#define NP_BD(OP_)                                                                                        \
    case OP_:                                                                                             \
        if constexpr (!LIGHT || fused_light_binary(OP_)) binary_all<OP_, N>(acc, oth, swap, quirk, body); \
        break
    switch (op) {
        NP_BD(NP_ADD); NP_BD(NP_SUBTRACT); NP_BD(NP_MULTIPLY); NP_BD(NP_DIVIDE); NP_BD(NP_MOD); NP_BD(NP_POW);
        NP_BD(NP_ARCTAN2); NP_BD(NP_EQUAL); NP_BD(NP_NOT_EQUAL); NP_BD(NP_GREATER); NP_BD(NP_GREATER_EQUAL);
        NP_BD(NP_LESS); NP_BD(NP_LESS_EQUAL); NP_BD(NP_MAXIMUM); NP_BD(NP_MINIMUM);
        default: break;
    }
#undef NP_BD
}

// U slots of G contiguous elements per thread (G = 4: one float4 per slot), N = U*G values held in
// registers.  Slot v of a span starts at element elem0 + v*G.  Operand arrays are streamed: the
// load of the next array operand is issued as soon as the previous one has been consumed, so it is
// in flight while the steps in between compute; with input 0 that keeps two loads outstanding per
// thread without holding every input in registers (which cost occupancy: 6 inputs x 8 values).
// KEEP: the kernel variant for chains that name a full array MORE THAN ONCE (f(x) * y + x, (x - y) * y): full arrays are read with
// plain loads, so that the second read of a line finds it in the L2 — non-temporal loads do not leave it there and the array
// was streamed from HBM twice (10^8 elements: x * y + x 247 us = 16 B/elem -> 192 us; tools/sq_chain_probe.py).  A kernel of its
// own: the choice as a run-time branch in the common kernel cost the other chains 3-7 % (registers, occupancy 8 -> 7).
template <int U, int G, typename I, bool RAGGED = false, bool KEEP = false>
__device__ __forceinline__ void fused_fetch(float (&dst)[U * G], const float *p, int idx, const I (&first)[U],
                                            const I (&row)[U], const I (&col)[U], const bool (&live)[U],
                                            I ragged_cols = 0) {
    if (idx == FUSED_IDX_FULL) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (G == 4) {
                v4f t = v4f{0.0f, 0.0f, 0.0f, 0.0f};
                if constexpr (KEEP) {
                    if (live[u]) t = *(const v4f_u *)(p + (size_t)first[u]);
                } else {
                    if (live[u]) t = __builtin_nontemporal_load((const v4f_u *)(p + (size_t)first[u]));
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[u * 4 + e] = t[e];
            } else {
                dst[u] = live[u] ? (KEEP ? p[(size_t)first[u]] : __builtin_nontemporal_load(p + (size_t)first[u])) : 0.0f;
            }
        }
    } else if (idx == FUSED_IDX_ROW) {
        // one row of `cols` floats, re-read by every result row: cache-resident, plain loads.  With cols % 4 != 0
        // (ragged_cols = cols, else 0) a slot may straddle two rows: those few slots walk the row with wrap-around.
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (G == 4) {
                v4f t = v4f{0.0f, 0.0f, 0.0f, 0.0f};
                if (live[u]) {
                    if (!RAGGED || col[u] + 3 < ragged_cols) {
                        t = *(const v4f_u *)(p + (size_t)col[u]);
                    } else {
                        I c = col[u];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            t[e] = p[(size_t)c];
                            c = c + 1 == ragged_cols ? (I)0 : c + 1;
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[u * 4 + e] = t[e];
            } else {
                dst[u] = live[u] ? p[(size_t)col[u]] : 0.0f;
            }
        }
    } else {
        // one value per result row (column operand) or one value in all (0-d device scalar)
        if (RAGGED && G == 4 && idx == FUSED_IDX_COL) {   // uniform: a slot that runs into the next row takes that row's value for its tail
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool straddles = live[u] && col[u] + 3 >= ragged_cols;
                const float t0 = live[u] ? p[(size_t)row[u]] : 0.0f;
                const float t1 = straddles ? p[(size_t)row[u] + 1] : t0;
#pragma unroll
                for (int e = 0; e < G; ++e) dst[u * G + e] = (straddles && col[u] + e >= ragged_cols) ? t1 : t0;
            }
            return;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float t = live[u] ? p[idx == FUSED_IDX_COL ? (size_t)row[u] : (size_t)0] : 0.0f;
#pragma unroll
            for (int e = 0; e < G; ++e) dst[u * G + e] = t;
        }
    }
}

// VACC: the sink accumulates per register element (rv[u * G + e], for column reductions where every
// element of a slot belongs to a different result) instead of into the one scalar racc.
template <int U, int G, bool LIGHT, typename I, bool VACC, bool RAGGED = false, bool KEEP = false>
__device__ __forceinline__ void fused_span_impl(FusedArgsK f, float *__restrict__ out, I elem0, I nslots, I tid,
                                                I stride, float &racc, float (&rv)[U * G]) {
    constexpr int N = U * G;
    const int sink = f->sink;
    const int n_ops = f->n_ops;
    const float *in0 = f->in0, *first_prefetch = f->first_prefetch;
    const float scalar0 = f->scalar0;
    const int first_prefetch_idx = f->first_prefetch_idx;
    const I cols = (I)f->cols;
    const I ragged_cols = RAGGED ? cols : (I)0;   // float4 slots of a broadcast with cols % 4 != 0 may straddle rows (a kernel variant of its own:
                                                  // with the extra code in the common kernel, aligned exp(X) + col lost 9 %)
    const unsigned div_m = f->div_m, div_s1 = f->div_s1, div_s2 = f->div_s2;
    for (I base = 0; base < nslots; base += stride * U) {
        I first[U], row[U], col[U];
        bool live[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const I v = base + (I)u * stride + tid;   // coalesced per u
            live[u] = v < nslots;
            first[u] = elem0 + v * G;
            row[u] = col[u] = 0;
            if (cols) {   // uniform: only chains with a broadcast operand need (row, col)
                if constexpr (sizeof(I) == 4)
                    row[u] = (I)fast_div((unsigned)first[u], div_m, div_s1, div_s2);
                else
                    row[u] = first[u] / cols;
                col[u] = first[u] - row[u] * cols;
            }
        }
        // the chain value starts as input 0 and lives in acc from the first load on: input 0 is not kept in a second
        // set of registers for the rare step that names it again (x * x ...) — that step re-reads it (an L2 hit in the KEEP
        // variant of the kernel, which such chains get), and every other chain saves N registers and N moves per trip
        float acc[N], nxt[N];
        if (in0) {
            fused_fetch<U, G, I, false, KEEP>(acc, in0, FUSED_IDX_FULL, first, row, col, live);
        } else {
#pragma unroll
            for (int e = 0; e < N; ++e) acc[e] = scalar0;
        }
        if (first_prefetch) {
            fused_fetch<U, G, I, RAGGED, KEEP>(nxt, first_prefetch, first_prefetch_idx, first, row, col, live, ragged_cols);
        } else {
#pragma unroll
            for (int e = 0; e < N; ++e) nxt[e] = 0.0f;
        }
        for (int k = 0; k < n_ops; ++k) {
            struct {
                int kind, op, swap, quirk, src_kind, prefetch_idx;
                float p0, p1, scalar;
                const float *prefetch;
                size_t body_end;
            } o = {f->ops[k].kind, f->ops[k].op, f->ops[k].swap, f->ops[k].quirk, f->ops[k].src_kind,
                   f->ops[k].prefetch_idx, f->ops[k].p0, f->ops[k].p1, f->ops[k].scalar, f->ops[k].prefetch,
                   f->ops[k].body_end};
            if (o.kind == NP_FUSED_UNARY) {
                unary_dispatch<N, LIGHT>(o.op, acc, o.p0, o.p1);
                continue;
            }
            float oth[N];
            bool body[N];
            // one uniform branch per step (written per element, the compiler emitted a branch ladder
            // for every one of the N selects)
            if (o.src_kind == FUSED_SRC_STREAM) {
#pragma unroll
                for (int e = 0; e < N; ++e) oth[e] = nxt[e];
            } else if (o.src_kind == FUSED_SRC_INPUT0) {
                if (in0) {
                    fused_fetch<U, G, I, false, KEEP>(oth, in0, FUSED_IDX_FULL, first, row, col, live);
                } else {
#pragma unroll
                    for (int e = 0; e < N; ++e) oth[e] = scalar0;
                }
            } else {
#pragma unroll
                for (int e = 0; e < N; ++e) oth[e] = o.scalar;
            }
#pragma unroll
            for (int e = 0; e < N; ++e) {
                // body_end is a multiple of 8 and float4 slots start at multiples of 4: one flag per slot
                body[e] = (size_t)first[e / G] < o.body_end;
            }
            if (o.src_kind == FUSED_SRC_STREAM && o.prefetch)
                fused_fetch<U, G, I, RAGGED, KEEP>(nxt, o.prefetch, o.prefetch_idx, first, row, col, live, ragged_cols);
            binary_dispatch<N, LIGHT>(o.op, acc, oth, o.swap != 0, o.quirk != 0, body);
        }
        if constexpr (VACC) {
            if (sink == NP_SUM) {
#pragma unroll
                for (int e = 0; e < N; ++e) rv[e] += live[e / G] ? acc[e] : 0.0f;
            } else if (sink == NP_PROD) {
#pragma unroll
                for (int e = 0; e < N; ++e) rv[e] *= live[e / G] ? acc[e] : 1.0f;
            } else if (sink == NP_MIN) {
#pragma unroll
                for (int e = 0; e < N; ++e)
                    if (live[e / G] && acc[e] < rv[e]) rv[e] = acc[e];
            } else {
#pragma unroll
                for (int e = 0; e < N; ++e)
                    if (live[e / G] && acc[e] > rv[e]) rv[e] = acc[e];
            }
            continue;
        }
        if (sink >= 0) {
            // reduction at the end of the chain: the value never goes to memory (uniform branch;
            // same combine rules as np_reduce_all: NaN never replaces in min / max)
            if (sink == NP_SUM) {
#pragma unroll
                for (int e = 0; e < N; ++e) racc += live[e / G] ? acc[e] : 0.0f;
            } else if (sink == NP_PROD) {
#pragma unroll
                for (int e = 0; e < N; ++e) racc *= live[e / G] ? acc[e] : 1.0f;
            } else if (sink == NP_MIN) {
#pragma unroll
                for (int e = 0; e < N; ++e)
                    if (live[e / G] && acc[e] < racc) racc = acc[e];
            } else {
#pragma unroll
                for (int e = 0; e < N; ++e)
                    if (live[e / G] && acc[e] > racc) racc = acc[e];
            }
            continue;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!live[u]) continue;
            if constexpr (G == 4)
                __builtin_nontemporal_store(v4f{acc[u * 4], acc[u * 4 + 1], acc[u * 4 + 2], acc[u * 4 + 3]},
                                            (v4f_u *)(out + (size_t)first[u]));
            else
                out[first[u]] = acc[u];
        }
    }
}

template <int U, int G, bool LIGHT, typename I, bool RAGGED = false, bool KEEP = false>
__device__ __forceinline__ void fused_span(FusedArgsK f, float *__restrict__ out, I elem0, I nslots, I tid,
                                           I stride, float &racc) {
    float unused[U * G];
    fused_span_impl<U, G, LIGHT, I, false, RAGGED, KEEP>(f, out, elem0, nslots, tid, stride, racc, unused);
}

__device__ __forceinline__ float sink_identity(int sink) {
    return sink == NP_SUM ? 0.0f : sink == NP_PROD ? 1.0f : sink == NP_MIN ? INFINITY : -INFINITY;
}
__device__ __forceinline__ float sink_combine(int sink, float a, float b) {   // np::dev::r_combine with a run-time op
    if (sink == NP_SUM) return a + b;
    if (sink == NP_PROD) return a * b;
    if (sink == NP_MIN) return (b < a) ? b : a;
    return (b > a) ? b : a;
}

// Chain ending in a reduction over the LAST axis: out[r] = reduce_c chain(r, c).  One wave per row
// (BLOCK = false; rows of up to a few thousand elements, no barrier anywhere) or one workgroup per
// row (BLOCK = true).  scale = 1 / cols for a mean (applied as a division, like NDArray::mean).
template <int G, bool LIGHT, bool BLOCK, typename I, int UU = 2>
__global__ __launch_bounds__(256) void fused_chain_rows_kernel(FusedArgs by_value, float *__restrict__ out, I rows, I cols,
                                                               unsigned L, I chunk_slots, float mean_div) {
    (void)by_value;
    FusedArgsK f = (FusedArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    const int sink = f->sink;
    // wave mode: groups of L lanes (a power of two <= 64) own a row each, so short rows still fill the wave
    const I lane = BLOCK ? threadIdx.x : (threadIdx.x & (L - 1));
    const I width = BLOCK ? 256 : L;
    const I groups = 64 / L;
    const I first_row = BLOCK ? (I)blockIdx.x : ((I)blockIdx.x * 4 + (threadIdx.x >> 6)) * groups + (threadIdx.x & 63) / L;
    const I row_stride = BLOCK ? (I)gridDim.x : (I)gridDim.x * 4 * groups;
    __shared__ float lds4[4];
    // workgroup mode may cut a row into gridDim.y chunks of chunk_slots slots (a handful of very long rows):
    // out then holds [row][chunk] partials for np_reduce_axis to fold
    const I slot0 = BLOCK ? (I)blockIdx.y * chunk_slots : 0;
    const I nslots = BLOCK ? (slot0 + chunk_slots < cols / G ? chunk_slots : cols / G - slot0) : cols / G;
    const I out_stride = BLOCK ? (I)gridDim.y : 1, out_off = BLOCK ? (I)blockIdx.y : 0;
    for (I r = first_row; r < rows; r += row_stride) {
        float racc = sink_identity(sink);
        fused_span<UU, G, LIGHT, I>(f, out, r * cols + slot0 * G, nslots, lane, width, racc);
        if constexpr (G == 4) {   // a row of cols % 4 != 0 floats: its last 1-3 elements, one per lane, behind the float4 slots
            const I tail = cols - (cols / 4) * 4;
            if (tail != 0 && (!BLOCK || blockIdx.y == gridDim.y - 1))
                fused_span<1, 1, LIGHT, I>(f, out, r * cols + (cols / 4) * 4, tail, lane, width, racc);
        }
        if constexpr (BLOCK) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) racc = sink_combine(sink, racc, __shfl_down(racc, off, 64));
        } else {
            for (unsigned off = L >> 1; off > 0; off >>= 1) racc = sink_combine(sink, racc, __shfl_xor(racc, (int)off, 64));
        }
        if constexpr (BLOCK) {
            if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = racc;
            __syncthreads();
            if (threadIdx.x == 0) {
                float v = lds4[0];
                for (int w = 1; w < 4; ++w) v = sink_combine(sink, v, lds4[w]);
                out[(size_t)r * out_stride + out_off] = mean_div != 0.0f ? v / mean_div : v;
            }
            __syncthreads();
        } else if (lane == 0) {
            out[r] = mean_div != 0.0f ? racc / mean_div : racc;
        }
    }
}

// Chain ending in a reduction over the FIRST axis of a rows x cols result: out[c] = reduce_r chain(r, c).
// A workgroup owns 64 slots of G columns; its four waves take every fourth row of the block's row chunk
// (blockIdx.y), each lane accumulating its own G columns (VACC), and are combined through LDS.  With
// more than one row chunk the partials [chunk][cols] are folded by np_reduce_axis.
template <int G, bool LIGHT, bool WIDE, typename I, int CU = 2>
__global__ __launch_bounds__(256) void fused_chain_cols_kernel(FusedArgs by_value, float *__restrict__ out, I rows, I cols,
                                                               I rows_per_chunk, float mean_div) {
    (void)by_value;
    FusedArgsK f = (FusedArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    const int sink = f->sink;
    const I lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const I slots_per_row = cols / G;             // cols % G == 0
    const I r0 = (I)blockIdx.y * rows_per_chunk;
    const I r1 = r0 + rows_per_chunk < rows ? r0 + rows_per_chunk : rows;
    float rv[CU * G], racc = 0.0f;   // CU rows in flight per lane, one accumulator set per row slot
#pragma unroll
    for (int e = 0; e < CU * G; ++e) rv[e] = sink_identity(sink);
    if constexpr (WIDE) {
        // the whole workgroup walks down the chunk row by row: 256 slots = 4 KiB of one row per step (one
        // DRAM-friendly contiguous segment instead of four 1 KiB segments of four different rows), every
        // thread owns its G columns for the whole chunk, so nothing has to be combined across waves
        const I slot = (I)blockIdx.x * 256 + threadIdx.x;
        if (slot < slots_per_row) {
            fused_span_impl<CU, G, LIGHT, I, true>(f, out, r0 * cols, (r1 - r0) * slots_per_row, slot, slots_per_row, racc, rv);
#pragma unroll
            for (int e = 0; e < G; ++e) {
                float v = rv[e];
                for (int u = 1; u < CU; ++u) v = sink_combine(sink, v, rv[u * G + e]);
                out[(size_t)blockIdx.y * cols + (size_t)slot * G + e] = mean_div != 0.0f ? v / mean_div : v;
            }
        }
        return;
    }
    const I slot = (I)blockIdx.x * 64 + lane;
    // slot index v of the span <-> (row r0 + v / slots_per_row, slot v % slots_per_row): this thread's
    // slots are v = (wave + 4 i) * slots_per_row + slot
    if (slot < slots_per_row)
        fused_span_impl<CU, G, LIGHT, I, true>(f, out, r0 * cols, (r1 - r0) * slots_per_row, wave * slots_per_row + slot,
                                              4 * slots_per_row, racc, rv);
    // cross-wave combine through LDS, lane fastest: a wave's 64 stores / loads of one element land on 64 different banks
    // (lane-major [lane][e] put lanes l and l + 16 on one bank: 74 % of this kernel's LDS cycles were conflicts)
    __shared__ float part[4][G][64];
#pragma unroll
    for (int e = 0; e < G; ++e) {
        float v = rv[e];
        for (int u = 1; u < CU; ++u) v = sink_combine(sink, v, rv[u * G + e]);
        part[wave][e][lane] = v;
    }
    __syncthreads();
    float v[G];
#pragma unroll
    for (int e = 0; e < G; ++e) {
        v[e] = part[0][e][lane];
        for (int w = 1; w < 4; ++w) v[e] = sink_combine(sink, v[e], part[w][e][lane]);
    }
    // One row of results per chunk of rows; with several chunks np_reduce_axis folds them in a second launch.  (Round 3
    // folded them here, in the last workgroup of each column block to finish — memory-side partials, a ticket per block,
    // profiles/r03/fused_cols_ab.log: 96 us against 75 + the ~9 us second launch.  The 64 partial rows arrive one
    // memory round trip after the other at the end of a kernel that has nothing left to hide them behind.)
    if (wave == 0 && slot < slots_per_row) {
#pragma unroll
        for (int e = 0; e < G; ++e) out[(size_t)blockIdx.y * cols + (size_t)slot * G + e] = mean_div != 0.0f ? v[e] / mean_div : v[e];
    }
}

template <bool VEC, int U, bool LIGHT, typename I, bool RAGGED = false, bool KEEP = false>
__global__ __launch_bounds__(256) void fused_chain_kernel(FusedArgs by_value, float *__restrict__ out, I n) {
    (void)by_value;   // first kernel argument: lives at offset 0 of the kernarg segment
    FusedArgsK f = (FusedArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    const I stride = (I)gridDim.x * blockDim.x;
    const I tid = (I)blockIdx.x * blockDim.x + threadIdx.x;
    const int sink = f->sink;
    float racc = sink == NP_SUM ? 0.0f : sink == NP_PROD ? 1.0f : sink == NP_MIN ? INFINITY : -INFINITY;
    if constexpr (VEC) {
        const I nvec = n / 4;
        fused_span<U, 4, LIGHT, I, RAGGED, KEEP>(f, out, (I)0, nvec, tid, stride, racc);
        fused_span<1, 1, LIGHT, I>(f, out, nvec * 4, n - nvec * 4, tid, stride, racc);   // ragged tail (< 4 elements)
    } else {
        fused_span<U, 1, LIGHT, I>(f, out, (I)0, n, tid, stride, racc);   // 4-byte aligned views
    }
    if (sink >= 0) {   // one partial per workgroup, folded in a fixed order (deterministic) by the last workgroup or a second launch
        __shared__ float lds4[4];
        unsigned *ticket = f->ticket;
        float *result = f->result;
#define NP_FOLD(OP_) np::dev::fold_in_last_workgroup<OP_>(np::dev::block_reduce<OP_>(racc, lds4), out, ticket, result, 1.0f, lds4)
        if (sink == NP_SUM) NP_FOLD(NP_SUM);
        else if (sink == NP_PROD) NP_FOLD(NP_PROD);
        else if (sink == NP_MIN) NP_FOLD(NP_MIN);
        else NP_FOLD(NP_MAX);
#undef NP_FOLD
    }
}

}  // namespace


// Chain ending in a reduction over the LAST axis of a matrix with very many SHORT rows (sum(exp(X), 1) over 10 classes):
// the row kernels above give a lane group to each row and load 4 bytes per lane at the row stride.  Here — as in
// reduce_rows_staged (np_reduce.hip), whose fold order this repeats, so that the fused result stays bit-identical to the
// op-by-op one — a workgroup walks a contiguous slab of R rows as the flat float4 stream it is, runs the chain on each
// float4 (every operand kind can be fetched there: the element's row and column are known), parks the chain values in
// LDS (row pitch len | 1 words) and then every thread folds its rows.  R is a multiple of 256, so slabs start on 16 bytes.
struct StagedCtx {
    FusedArgsK f;
    float *slab;          // LDS
    size_t base, row0;    // flat index / row index of the slab's first element
    unsigned len, pitch, magic;
};

// N values (one float4 or one element) of the slab through the chain; e = slab-relative flat index of the first.
// (Out of line — one copy of the interpreter, called once per float4 with all eight loads of a thread issued first —
// it cost 2.3x: register save / restore around every call, 0.129 -> 0.29 ms on 10^7 x 10.)
template <int N, bool LIGHT>
__device__ __forceinline__ void staged_run(const StagedCtx &cx, unsigned e, v4f x) {
    FusedArgsK f = cx.f;
    const float *in0 = f->in0;
    const unsigned len = cx.len;
    float acc[N];
    unsigned r[N], c[N];
    r[0] = __umulhi(e, cx.magic);                // e / len (magic = ceil(2^32 / len), exact for e < 2^16)
    c[0] = e - r[0] * len;
#pragma unroll
    for (int j = 1; j < N; ++j) {
        const bool wrap = c[j - 1] + 1 == len;
        c[j] = wrap ? 0u : c[j - 1] + 1;
        r[j] = r[j - 1] + (wrap ? 1u : 0u);
    }
    const size_t ge = cx.base + e;
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = x[j];
    auto fetch = [&](const float *p, int idx, float (&dst)[N]) {
        if (idx == FUSED_IDX_FULL) {
            if constexpr (N == 4) {
                const v4f t = *(const v4f_u *)(p + ge);
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[j] = t[j];
            } else {
                dst[0] = p[ge];
            }
        } else {
#pragma unroll
            for (int j = 0; j < N; ++j)
                dst[j] = p[idx == FUSED_IDX_ROW ? (size_t)c[j] : idx == FUSED_IDX_COL ? cx.row0 + r[j] : (size_t)0];
        }
    };
    const float *stream = f->first_prefetch;
    int stream_idx = f->first_prefetch_idx;
    const int n_ops = f->n_ops;
    for (int k = 0; k < n_ops; ++k) {
        const int kind = f->ops[k].kind, op = f->ops[k].op;
        if (kind == NP_FUSED_UNARY) {
            unary_dispatch<N, LIGHT>(op, acc, f->ops[k].p0, f->ops[k].p1);
            continue;
        }
        float oth[N];
        bool body[N];
        const int src = f->ops[k].src_kind;
        if (src == FUSED_SRC_STREAM) {
            fetch(stream, stream_idx, oth);
            stream = f->ops[k].prefetch;
            stream_idx = f->ops[k].prefetch_idx;
        } else if (src == FUSED_SRC_INPUT0) {
            fetch(in0, FUSED_IDX_FULL, oth);
        } else {
#pragma unroll
            for (int j = 0; j < N; ++j) oth[j] = f->ops[k].scalar;
        }
        const size_t body_end = f->ops[k].body_end;
#pragma unroll
        for (int j = 0; j < N; ++j) body[j] = ge + j < body_end;
        binary_dispatch<N, LIGHT>(op, acc, oth, f->ops[k].swap != 0, f->ops[k].quirk != 0, body);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) cx.slab[r[j] * cx.pitch + c[j]] = acc[j];
}

template <int SINK, bool LIGHT>
__global__ __launch_bounds__(256) void fused_chain_rows_staged_kernel(FusedArgs by_value, float *__restrict__ out,
                                                                      size_t rows_total, unsigned len, unsigned R,
                                                                      unsigned magic, float mean_div, unsigned T) {
    (void)by_value;
    extern __shared__ __attribute__((aligned(16))) float slab[];
    StagedCtx cx;
    cx.f = (FusedArgsK)__builtin_amdgcn_kernarg_segment_ptr();
    cx.slab = slab;
    cx.len = len;
    cx.pitch = len | 1u;
    cx.magic = magic;
    cx.row0 = (size_t)blockIdx.x * R;
    cx.base = cx.row0 * len;
    const unsigned pitch = cx.pitch;
    const size_t row0 = cx.row0;
    const unsigned rows = (unsigned)(rows_total - row0 < R ? rows_total - row0 : R);
    const unsigned total = rows * len;
    const float *in0 = cx.f->in0;
    // two float4s of input 0 in flight per trip (the chain is a long stretch of code the compiler does not move loads
    // across): sum(exp(X), 1) on 10^7 x 10 0.129 ms with one, 0.109 with two, 0.123 with four (four inlined copies of the
    // interpreter)
    const unsigned nvec = total / 4;
    for (unsigned v = threadIdx.x; v < nvec; v += 512) {
        const bool two = v + 256 < nvec;
        const v4f x0 = __builtin_nontemporal_load((const v4f *)(in0 + cx.base + (size_t)v * 4));
        const v4f x1 = two ? __builtin_nontemporal_load((const v4f *)(in0 + cx.base + (size_t)(v + 256) * 4)) : v4f{0, 0, 0, 0};
        staged_run<4, LIGHT>(cx, v * 4, x0);
        if (two) staged_run<4, LIGHT>(cx, (v + 256) * 4, x1);
    }
    for (unsigned e = nvec * 4 + threadIdx.x; e < total; e += 256) staged_run<1, LIGHT>(cx, e, v4f{in0[cx.base + e], 0, 0, 0});
    __syncthreads();
    if (T > 1) {
        // rows of 49 ... 1024 floats: only a few rows fit a slab, so a group of T lanes (a power of two <= 64) folds each
        // row — lane t takes elements t, t + T, ... (consecutive lanes, consecutive LDS words), then an xor-shuffle tree
        const unsigned g = threadIdx.x / T, t = threadIdx.x & (T - 1), per_pass = 256u / T;
        for (unsigned r = g; r < rows; r += per_pass) {
            const float *p = slab + r * pitch;
            float a = np::dev::r_identity<SINK>();
            for (unsigned c = t; c < len; c += T) a = np::dev::r_combine<SINK>(a, p[c]);
            for (unsigned off = T >> 1; off > 0; off >>= 1) a = np::dev::r_combine<SINK>(a, __shfl_xor(a, (int)off, 64));
            if (t == 0) out[row0 + r] = mean_div != 0.0f ? __fdiv_rn(a, mean_div) : a;
        }
        return;
    }
    for (unsigned r = threadIdx.x; r < rows; r += 256) {
        const float *p = slab + r * pitch;
        float a0 = np::dev::r_identity<SINK>(), a1 = a0;
        unsigned c = 0;
        for (; c + 1 < len; c += 2) {
            a0 = np::dev::r_combine<SINK>(a0, p[c]);
            a1 = np::dev::r_combine<SINK>(a1, p[c + 1]);
        }
        if (c < len) a0 = np::dev::r_combine<SINK>(a0, p[c]);
        float v = np::dev::r_combine<SINK>(a0, a1);
        if (mean_div != 0.0f) v = __fdiv_rn(v, mean_div);
        out[row0 + r] = v;
    }
}

// very many short rows: the staged kernel (the same window as reduce_rows_staged in np_reduce.hip, which the op-by-op
// path takes for these shapes — the two must agree for the fused result to stay bit-identical)
static bool fused_rows_staged_shape(size_t rows, size_t cols) {
    return cols > 4 && cols <= 48 && rows * cols >= (size_t(8) << 20);
}
// Rows of 49 ... 512 floats of a large array that the lane-group kernel packs badly take the same staged kernel, a lane
// group folding each row of the slab: rows whose length is not a multiple of 4 (the lane-group kernel walks their last
// 1-3 elements in a pass of their own) and rows whose float4 slots do not fill the lane groups (100 floats = 25 slots on
// 16 lanes).  Measured, same box (profiles/r03/fused_mid_rows_ab.log): 2e6 x 50 2.5 -> 4.3 TB/s, 400000 x 250 2.9 -> 3.8,
// 800000 x 127 3.0 -> 3.9, 1e6 x 100 3.2 -> 3.6; well-packed rows stay where they are (64 / 256 / 1000 / 1024 floats:
// 3.8-4.8 TB/s there against 3.3-3.8 staged).  np_elementwise_set_variant(4000) switches the window off (A/B).
constexpr size_t kStagedMidMax = 512;
static bool fused_rows_staged_mid_shape(size_t rows, size_t cols) {
    if (g_variant == 4000 || cols <= 48 || cols > kStagedMidMax || rows < 4096 || rows * cols < (size_t(8) << 20)) return false;
    if (cols % 4 != 0) return true;
    const size_t slots = cols / 4;
    size_t L = 64;                                   // the lane-group kernel's choice (fused_chain_impl, axis_mode 1)
    while (L > 4 && L >= slots) L >>= 1;
    const size_t trips = (slots + 2 * L - 1) / (2 * L);
    return (double)slots / (double)(trips * 2 * L) < 0.9;
}

static bool fused_axis_shape_ok(size_t rows, size_t cols, int axis) {
    if (rows * cols >= (size_t(1) << 32)) return false;
    if (axis == 1 && (fused_rows_staged_shape(rows, cols) || fused_rows_staged_mid_shape(rows, cols))) return true;
    if (axis == 1) return cols >= 16 && (rows >= 128 || rows * cols >= (size_t(1) << 20));   // a lane group / wave / workgroup (or several) per row
    return rows >= 32 && cols / (cols % 4 == 0 ? 4 : 1) >= 32;   // first axis: a lane per column slot, waves interleaved over rows
}

// sink < 0: out receives rows*cols values; else out is a device float receiving the reduction — or,
// with axis_mode 1 / 0, rows / cols floats: the reduction over the last / first axis (mean_div != 0:
// divide the result by it)
static int fused_chain_impl(const float *const *inputs, const int *input_kinds, int n_inputs,
                            const np_fused_op *ops, int n_ops, float *out, size_t rows, size_t cols, int sink,
                            int axis_mode = -1, float mean_div = 0.0f) {
    const size_t n = rows * cols;
    if (n_inputs < 1 || n_inputs > FUSED_MAX_IN)
        return np::fail(NP_ERR_INVALID, "np_fused_chain: 1..%d inputs supported", FUSED_MAX_IN);
    if (n_ops < 0 || n_ops > FUSED_MAX_OPS)
        return np::fail(NP_ERR_INVALID, "np_fused_chain: at most %d ops per chain", FUSED_MAX_OPS);
    if (!inputs || !input_kinds || (n_ops > 0 && !ops) || !out)
        return np::fail(NP_ERR_INVALID, "np_fused_chain: null pointer");
    if (n == 0) return NP_OK;
    if (int rc = np::ensure_init()) return rc;
    FusedArgs f;
    f.n_ops = n_ops;
    bool vec = true;   // dword-aligned float4 accesses: pointers may start anywhere
    bool broadcast = false;
    for (int i = 0; i < n_inputs; ++i) {
        if (!inputs[i]) return np::fail(NP_ERR_INVALID, "np_fused_chain: null input %d", i);
        switch (input_kinds[i]) {
            case NP_FULL: break;
            // (float4 slots also when cols % 4 != 0, from 4 columns up: a slot then straddles at most two rows — fused_fetch)
            case NP_ROW: vec = vec && (cols % 4 == 0 || (cols >= 4 && axis_mode < 0)); broadcast = true; break;
            case NP_COL: vec = vec && (cols % 4 == 0 || (cols >= 4 && axis_mode < 0)); broadcast = true; break;
            case NP_SCALAR:
            case NP_HOST_SCALAR: break;
            default: return np::fail(NP_ERR_INVALID, "np_fused_chain: unknown operand kind %d", input_kinds[i]);
        }
    }
    if (input_kinds[0] != NP_FULL && input_kinds[0] != NP_HOST_SCALAR)
        return np::fail(NP_ERR_INVALID, "np_fused_chain: input 0 must be NP_FULL or NP_HOST_SCALAR");
    f.in0 = input_kinds[0] == NP_FULL ? inputs[0] : nullptr;
    f.scalar0 = input_kinds[0] == NP_FULL ? 0.0f : *inputs[0];
    f.first_prefetch = nullptr;
    f.first_prefetch_idx = FUSED_IDX_FULL;
    // does the chain name a full array more than once (as the operand of several steps, or input 0 again)?  Such a chain runs
    // on the interpreter's KEEP variant (see fused_fetch); np_elementwise_set_variant(7002): as every other chain (A/B)
    bool names_twice = false;
    if (g_variant != 7002) {
        for (int i = 0; i < n_inputs && !names_twice; ++i) {
            if (input_kinds[i] != NP_FULL) continue;
            int uses = (i == 0) ? 1 : 0;                       // input 0 is the chain's first value
            for (int k = 0; k < n_ops; ++k)
                if (ops[k].kind == NP_FUSED_BINARY && ops[k].operand >= 0 && ops[k].operand < n_inputs &&
                    input_kinds[ops[k].operand] == NP_FULL && inputs[ops[k].operand] == inputs[i])
                    ++uses;
            names_twice = uses >= 2;
        }
    }
    f.cols = broadcast ? cols : 0;
    f.sink = sink;
    f.ticket = nullptr;
    f.result = nullptr;
    f.div_m = f.div_s1 = f.div_s2 = 0;   // cols == 1: q = n
    if (broadcast && cols > 1 && cols <= 0xffffffffull) {
        unsigned l = 0;
        while ((1ull << l) < cols) ++l;
        f.div_m = (unsigned)((((1ull << 32) * ((1ull << l) - cols)) / cols) + 1);
        f.div_s1 = 1;
        f.div_s2 = l - 1;
    }
    np::Scratch partials;
    float *result = out;
    unsigned reduce_blocks = 0;
    if (sink >= 0 && axis_mode < 0) {
        // grid-stride loop over a capped grid: one workgroup-reduce + partial per block, so few,
        // long-lived blocks (NP_FUSED_RBPC blocks per CU for tools/fused_ab.py)
#ifdef NP_TUNING   // tools/fused_ab.py; the shipped library reads no environment variable
        static const int rbpc = getenv("NP_FUSED_RBPC") ? atoi(getenv("NP_FUSED_RBPC")) : 16;
#else
        constexpr int rbpc = 16;
#endif
        const size_t want = (n / 4 + 255) / 256 + 1, cap = (size_t)np::num_cus() * (size_t)rbpc;
        reduce_blocks = (unsigned)np::capped_grid(want, cap);
        if (int rc = partials.alloc(reduce_blocks * sizeof(float))) return rc;
        out = (float *)partials.ptr;
        if (reduce_blocks <= np::kFoldInKernelMaxBlocks) {
            f.ticket = np::next_ticket();
            f.result = result;
        }
    }
    FusedStep *last_stream = nullptr;
    bool light = true;
    // the same chain as np_fused_static.hip wants it, while it still may be one of the compiled ones: 1-3 steps on a full
    // input 0, float4-divisible rows under a broadcast, 32-bit indices, no AVX-body quirk other than multiply's (variant 7000: interpreter only)
    np::FusedStaticDesc sd{};
    bool compiled = n_ops >= 1 && n_ops <= 3 && f.in0 && vec && (!broadcast || cols % 4 == 0) && n < (size_t(1) << 31) &&
                    g_variant != 7000 && (!names_twice || g_variant == 7001);   // (the compiled kernels stream every array: 7001 = A/B)
    sd.n_ops = n_ops;
    sd.in0 = f.in0;
    sd.bcast_cols = broadcast ? (unsigned)cols : 0u;
    sd.div_m = f.div_m;
    sd.div_s1 = f.div_s1;
    sd.div_s2 = f.div_s2;
    for (int k = 0; k < n_ops; ++k) {
        const np_fused_op &o = ops[k];
        FusedStep &d = f.ops[k];
        d = FusedStep{};
        d.kind = o.kind;
        d.op = o.op;
        d.p0 = o.p0;
        d.p1 = o.p1;
        if (o.kind == NP_FUSED_UNARY) {
            if (o.op < 0 || o.op >= NP_UNARY_OP_COUNT) return np::fail(NP_ERR_INVALID, "np_fused_chain: unknown unary op %d", o.op);
            if (o.op == NP_ROUND) d.p0 = powf(10.0f, o.p0);   // as np_unary
            light = light && fused_light_unary(o.op);
        } else if (o.kind == NP_FUSED_BINARY) {
            if (o.op < 0 || o.op >= NP_BINARY_OP_COUNT) return np::fail(NP_ERR_INVALID, "np_fused_chain: unknown binary op %d", o.op);
            if (o.operand < 0 || o.operand >= n_inputs) return np::fail(NP_ERR_INVALID, "np_fused_chain: operand index out of range");
            if (o.op == NP_POW && !o.swap && input_kinds[o.operand] == NP_HOST_SCALAR && *inputs[o.operand] == 2.0f) {
                // value ** 2 with a PHP number: np_binary computes it as x * x (below, case NP_POW), so the chain does
                d.kind = NP_FUSED_UNARY;
                d.op = NP_UNARY_SQUARE;
                if (compiled) {
                    sd.kind[k] = d.kind;
                    sd.op[k] = d.op;           // not on the menu of np_fused_static.hip: the interpreter runs this chain
                }
                continue;
            }
            light = light && fused_light_binary(o.op);
            d.swap = o.swap;
            d.quirk = (o.flags & NP_QUIRK_AVX_BODY) ? 1 : 0;
            d.body_end = o.body_end;
            if (input_kinds[o.operand] == NP_HOST_SCALAR) {
                d.src_kind = FUSED_SRC_SCALAR;
                d.scalar = *inputs[o.operand];
            } else if (inputs[o.operand] == inputs[0] && input_kinds[o.operand] == NP_FULL && f.in0) {
                d.src_kind = FUSED_SRC_INPUT0;   // already in registers
            } else {
                d.src_kind = FUSED_SRC_STREAM;
                const int kind = input_kinds[o.operand];
                const int idx = kind == NP_FULL ? FUSED_IDX_FULL : kind == NP_ROW ? FUSED_IDX_ROW
                              : kind == NP_COL ? FUSED_IDX_COL : FUSED_IDX_ZERO;
                if (last_stream) {
                    last_stream->prefetch = inputs[o.operand];
                    last_stream->prefetch_idx = idx;
                } else {
                    f.first_prefetch = inputs[o.operand];
                    f.first_prefetch_idx = idx;
                }
                last_stream = &d;
            }
        } else {
            return np::fail(NP_ERR_INVALID, "np_fused_chain: unknown op kind %d", o.kind);
        }
        if (compiled) {
            sd.kind[k] = d.kind;
            sd.op[k] = d.op;
            sd.swap[k] = d.swap;
            sd.p0[k] = d.p0;
            sd.p1[k] = d.p1;
            sd.scalar[k] = d.scalar;
            sd.quirk[k] = 0;
            sd.body_end[k] = 0;
            if (d.kind == NP_FUSED_BINARY) {
                // the AVX-body quirk: the compiled kernels know multiply's (what the binding's chains carry); mod / equal are not on their menu
                if (d.quirk && binary_has_quirk(d.op)) {
                    if (d.op == NP_MULTIPLY) {
                        sd.quirk[k] = 1;
                        sd.body_end[k] = (unsigned)(d.body_end < n ? d.body_end : n);
                    } else {
                        compiled = false;
                    }
                }
                const int kind = input_kinds[o.operand];
                sd.operand[k] = d.src_kind == FUSED_SRC_SCALAR ? nullptr : d.src_kind == FUSED_SRC_INPUT0 ? f.in0 : inputs[o.operand];
                sd.idx[k] = d.src_kind != FUSED_SRC_STREAM ? 0 : kind == NP_FULL ? 0 : kind == NP_ROW ? 1 : kind == NP_COL ? 2 : 3;
            }
        }
    }
    // float4 slots per thread-trip: measured (tools/fused_ab.py, profiles/r01/fused_ab.log) — the
    // LIGHT interpreter is best at 1 slot (49 VGPRs, 8 waves/SIMD), the full one at 2.
    // NP_FUSED_U / NP_FUSED_FULL override for that A/B.
#ifdef NP_TUNING
    static const int fu_env = getenv("NP_FUSED_U") ? atoi(getenv("NP_FUSED_U")) : 0;
    static const bool force_full = getenv("NP_FUSED_FULL") != nullptr;
#else
    constexpr int fu_env = 0;
    constexpr bool force_full = false;
#endif
    hipStream_t s = np::stream();
    if (force_full) light = false;
    const bool staged_mid = axis_mode == 1 && fused_rows_staged_mid_shape(rows, cols) && f.in0 && ((uintptr_t)f.in0 & 15u) == 0;
    if (axis_mode == 1 && (fused_rows_staged_shape(rows, cols) || staged_mid)) {
        // (needs input 0 as a 16-byte aligned full array; anything else — a chain that starts from a scalar, a view at an
        // odd offset — is materialised and handed to np_reduce_axis like the shapes fused_axis_shape_ok turns away)
        if (!f.in0 || ((uintptr_t)f.in0 & 15u) != 0) {
            np::Scratch tmp;
            if (int rc = tmp.alloc(n * sizeof(float))) return rc;
            if (int rc = fused_chain_impl(inputs, input_kinds, n_inputs, ops, n_ops, (float *)tmp.ptr, rows, cols, -1)) return rc;
            const int rop = mean_div != 0.0f ? NP_MEAN : sink;
            return np_reduce_axis(rop, (const float *)tmp.ptr, rows, cols, 1, out, 0);
        }
        const unsigned pitch = (unsigned)cols | 1u;
        unsigned R = (8192u / pitch) / 256u * 256u, T = 1;
        if (R < 256) R = 256;
        if (staged_mid) {
            // ~32 KB of LDS per workgroup (4-5 resident per CU keep the streaming phase's loads in flight); R % 4 == 0 keeps
            // every slab's first element 16-byte aligned; T lanes per row, the largest power of two that R rows leave
            R = (8192u / pitch) / 4u * 4u;
            if (R < 4) R = 4;
            T = 64;
            while (T > 1 && T * R > 256u) T >>= 1;
        }
        const size_t blocks = (rows + R - 1) / R;
        if (blocks > 0x7fffffffu) return np::fail(NP_ERR_INVALID, "np_fused_chain_reduce_axis: too many rows");
        const size_t lds = (size_t)R * pitch * sizeof(float);
        const unsigned magic = (unsigned)((0x100000000ull + cols - 1) / cols);
        // (the LIGHT interpreter — only ops whose bodies are a handful of instructions — where the chain allows: fewer
        // registers, more waves, more loads in flight; variant 4001 keeps the full one for the A/B)
#define NP_FRS(SINK_)                                                                                                                        \
    do {                                                                                                                                     \
        if (light && g_variant != 4001)                                                                                                      \
            fused_chain_rows_staged_kernel<SINK_, true><<<(unsigned)blocks, 256, lds, s>>>(f, out, rows, (unsigned)cols, R, magic, mean_div, T);  \
        else                                                                                                                                 \
            fused_chain_rows_staged_kernel<SINK_, false><<<(unsigned)blocks, 256, lds, s>>>(f, out, rows, (unsigned)cols, R, magic, mean_div, T); \
    } while (0)
        if (sink == NP_SUM) NP_FRS(NP_SUM);
        else if (sink == NP_PROD) NP_FRS(NP_PROD);
        else if (sink == NP_MIN) NP_FRS(NP_MIN);
        else NP_FRS(NP_MAX);
#undef NP_FRS
        NP_LAUNCH_CHECK("fused_chain_rows_staged_kernel");
        return NP_OK;
    }
    if (axis_mode == 1) {
        // last axis: a wave per row while rows are plentiful and short enough to leave a wave busy, else a workgroup
        const bool block = cols >= 16384 || rows < (size_t)np::num_cus() * 16;
        // float4 slots also for rows of cols % 4 != 0 floats (dword-aligned accesses; the kernel takes the 1-3 leftover
        // elements of each row one per lane): such rows used to run one ELEMENT per slot
        const size_t slots = cols / 4;
        // The lane-group (wave) mode is bound by instruction issue, not by HBM (profiles/r03/rows_pmc.txt: SALU + VALU
        // activity scales with the time; rows of 64 floats issue 50 % more instructions per element than rows of 4000).
        // For LIGHT chains over rows of 128 ... 512 floats the interpreter's per-trip dispatch is amortised over 4 float4
        // slots per lane instead of 2, the lane group sized to ~4 slots per lane: 128 / 256 / 500 / 512 floats +7-12 %
        // (192, 384: neutral); shorter and longer rows lose 3-18 % to the extra registers and keep 2
        // (profiles/r03/fused_rows_u4_ab.log; variant 5000 = 2 everywhere).
        const bool four = light && !block && cols >= 128 && cols <= 512 && g_variant != 5000;
        unsigned L = 64;                                   // ~2 (or ~4) slots per lane per trip
        while (L > 4 && (size_t)L * (four ? 2 : 1) >= slots) L >>= 1;
        size_t grid = block ? rows : (rows + 4 * (64 / L) - 1) / (4 * (64 / L));
        const size_t cap = (size_t)np::num_cus() * 16;
        if (grid > cap) grid = cap;
        // a handful of very long rows: several workgroups per row (>= 1024 slots each), partials folded afterwards
        size_t chunks = 1;
        if (block && rows < (size_t)np::num_cus() * 4) {
            chunks = ((size_t)np::num_cus() * 8 + rows - 1) / rows;
            const size_t max_chunks = slots / 1024 ? slots / 1024 : 1;
            if (chunks > max_chunks) chunks = max_chunks;
            if (chunks > 65535) chunks = 65535;
        }
        const size_t chunk_slots = (slots + chunks - 1) / chunks;
        chunks = (slots + chunk_slots - 1) / chunk_slots;
        np::Scratch partial;
        float *dst = out;
        if (chunks > 1) {
            if (int rc = partial.alloc(rows * chunks * sizeof(float))) return rc;
            dst = (float *)partial.ptr;
        }
        const float div = chunks > 1 ? 0.0f : mean_div;
        if (compiled && !block && chunks == 1 && cols % 4 == 0 && np::fused_static_covers(sd, sink, 1)) {
            // a compiled chain (np_fused_static.hip) in the same wave-per-row geometry: ~2 slots per lane per trip
            unsigned Lc = 64;
            while (Lc > 4 && (size_t)Lc >= slots) Lc >>= 1;
            size_t gridc = (rows + 4 * (64 / Lc) - 1) / (4 * (64 / Lc));
            if (gridc > cap) gridc = cap;
            return np::fused_static_rows(sd, dst, rows, cols, Lc, div, (unsigned)gridc);
        }
        const dim3 grid2((unsigned)grid, (unsigned)chunks);
#define NP_FR(G_, LIGHT_, BLOCK_) fused_chain_rows_kernel<G_, LIGHT_, BLOCK_, uint32_t><<<grid2, 256, 0, s>>>(f, dst, (uint32_t)rows, (uint32_t)cols, L, (uint32_t)chunk_slots, div)
        if (four) fused_chain_rows_kernel<4, true, false, uint32_t, 4><<<grid2, 256, 0, s>>>(f, dst, (uint32_t)rows, (uint32_t)cols, L, (uint32_t)chunk_slots, div);
        else if (light) { if (block) NP_FR(4, true, true); else NP_FR(4, true, false); }
        else { if (block) NP_FR(4, false, true); else NP_FR(4, false, false); }
#undef NP_FR
        NP_LAUNCH_CHECK("fused_chain_rows_kernel");
        if (chunks > 1) {
            if (int rc = np_reduce_axis(sink, dst, rows, chunks, 1, out, 0)) return rc;
            if (mean_div != 0.0f) return np_binary(NP_DIVIDE, out, NP_FULL, &mean_div, NP_HOST_SCALAR, out, 1, rows, 0, 0);
        }
        return NP_OK;
    }
    if (axis_mode == 0) {
        const size_t g = cols % 4 == 0 ? 4 : 1;
        // tuning knobs (np_elementwise_set_variant): 1000 + k = k workgroups per CU, 64-slot column blocks;
        // 2000 + k = the same with 256-slot (whole-workgroup) column blocks
        const bool wide = g_variant >= 2000 && g_variant < 3000 ? true : g_variant >= 1000 && g_variant < 2000 ? false : kColsWideDefault;
        const size_t wg_per_cu = ((g_variant >= 1000 && g_variant < 3000) || g_variant / 1000 == 12 || g_variant / 1000 == 14) && g_variant % 1000
                                     ? (size_t)(g_variant % 1000) : kColsWgPerCuDefault;   // (12000 + k / 14000 + k: compiled chains, 2 / 4 rows in flight)
        const size_t block_slots = wide ? 256 : 64;
        const size_t col_blocks = (cols / g + block_slots - 1) / block_slots;
        size_t chunks = ((size_t)np::num_cus() * wg_per_cu + col_blocks - 1) / col_blocks;
        const size_t max_chunks = rows / 32 ? rows / 32 : 1;
        if (chunks > max_chunks) chunks = max_chunks;
        if (chunks > 65535) chunks = 65535;
        const size_t rows_per_chunk = (rows + chunks - 1) / chunks;
        chunks = (rows + rows_per_chunk - 1) / rows_per_chunk;
        np::Scratch partial;
        float *dst = out;
        if (chunks > 1) {
            if (int rc = partial.alloc(chunks * cols * sizeof(float))) return rc;
            dst = (float *)partial.ptr;
        }
        const float div = chunks > 1 ? 0.0f : mean_div;
        const dim3 grid((unsigned)col_blocks, (unsigned)chunks);
        if (compiled && g == 4 && !wide && np::fused_static_covers(sd, sink, 0)) {
            // a compiled chain (np_fused_static.hip): the same geometry, straight-line code
            if (int rc = np::fused_static_cols(sd, dst, rows, cols, rows_per_chunk, div, (unsigned)col_blocks, (unsigned)chunks,
                                               g_variant == 7004 || g_variant / 1000 == 14 ? 4 : 2))   // (A/B: 7004, 14000 + wg per CU)
                return rc;
            if (chunks > 1) {
                if (int rc = np_reduce_axis(sink, dst, 1, chunks, cols, out, 0)) return rc;
                if (mean_div != 0.0f) return np_binary(NP_DIVIDE, out, NP_FULL, &mean_div, NP_HOST_SCALAR, out, 1, cols, 0, 0);
            }
            return NP_OK;
        }
#define NP_FCOL(G_, LIGHT_, WIDE_) fused_chain_cols_kernel<G_, LIGHT_, WIDE_, uint32_t><<<grid, 256, 0, s>>>(f, dst, (uint32_t)rows, (uint32_t)cols, (uint32_t)rows_per_chunk, div)
        // variants 6004 / 6001 / 6003: the LIGHT float4 kernel with 4 / 1 / 3 rows in flight per lane (A/B, tools/fused_cols_ab.py)
        if (g == 4 && light && !wide && g_variant == 6004)
            fused_chain_cols_kernel<4, true, false, uint32_t, 4><<<grid, 256, 0, s>>>(f, dst, (uint32_t)rows, (uint32_t)cols, (uint32_t)rows_per_chunk, div);
        else if (g == 4 && light && !wide && g_variant == 6003)
            fused_chain_cols_kernel<4, true, false, uint32_t, 3><<<grid, 256, 0, s>>>(f, dst, (uint32_t)rows, (uint32_t)cols, (uint32_t)rows_per_chunk, div);
        else if (g == 4 && light && !wide && g_variant == 6001)
            fused_chain_cols_kernel<4, true, false, uint32_t, 1><<<grid, 256, 0, s>>>(f, dst, (uint32_t)rows, (uint32_t)cols, (uint32_t)rows_per_chunk, div);
        else if (wide) {
            if (g == 4) { if (light) NP_FCOL(4, true, true); else NP_FCOL(4, false, true); }
            else { if (light) NP_FCOL(1, true, true); else NP_FCOL(1, false, true); }
        } else {
            if (g == 4) { if (light) NP_FCOL(4, true, false); else NP_FCOL(4, false, false); }
            else { if (light) NP_FCOL(1, true, false); else NP_FCOL(1, false, false); }
        }
#undef NP_FCOL
        NP_LAUNCH_CHECK("fused_chain_cols_kernel");
        if (chunks > 1) {
            if (int rc = np_reduce_axis(sink, dst, 1, chunks, cols, out, 0)) return rc;
            if (mean_div != 0.0f) return np_binary(NP_DIVIDE, out, NP_FULL, &mean_div, NP_HOST_SCALAR, out, 1, cols, 0, 0);
        }
        return NP_OK;
    }
    if (compiled && np::fused_static_covers(sd, sink, -1)) {
        const unsigned grid = reduce_blocks ? reduce_blocks : grid_for(n / 4 + 1, 2, 0);
        if (int rc = np::fused_static_flat(sd, out, n, sink, grid, f.ticket, f.result)) return rc;
        if (sink >= 0 && !f.ticket) return np::fold_partials(sink, (const float *)partials.ptr, reduce_blocks, result);
        return NP_OK;
    }
    const bool small = n < (size_t(1) << 31);
#define NP_FC(VEC_, U_, LIGHT_)                                                                          \
    do {                                                                                                 \
        const unsigned grid = reduce_blocks ? reduce_blocks : grid_for(VEC_ ? n / 4 + 1 : n, U_, 0);     \
        if (small)                                                                                       \
            fused_chain_kernel<VEC_, U_, LIGHT_, uint32_t><<<grid, 256, 0, s>>>(f, out, (uint32_t)n);    \
        else                                                                                             \
            fused_chain_kernel<VEC_, U_, LIGHT_, uint64_t><<<grid, 256, 0, s>>>(f, out, (uint64_t)n);    \
    } while (0)
    // two float4 slots per lane for both interpreters: with the 8-instruction exp a wave's fixed cost (launch, kernarg
    // and descriptor loads, one scalar branch per step) weighs more than its registers — exp(X) + row 0.161 -> 0.139-0.143
    // ms, exp(a)*b+2 0.204-0.213 -> 0.197-0.203 (round 1, with the 14-instruction expf, one slot was ahead for the
    // light interpreter; tools/fused_ab.py with FUSED_AB_BCAST=1, profiles/r02/fused_u_ab.log)
    const int fu = fu_env ? fu_env : 2;
    if (vec && broadcast && cols % 4 != 0) {   // float4 slots that may straddle rows: the RAGGED variant of the two-slot kernels
        const unsigned grid = reduce_blocks ? reduce_blocks : grid_for(n / 4 + 1, 2, 0);
        if (small) {
            if (light) fused_chain_kernel<true, 2, true, uint32_t, true><<<grid, 256, 0, s>>>(f, out, (uint32_t)n);
            else fused_chain_kernel<true, 2, false, uint32_t, true><<<grid, 256, 0, s>>>(f, out, (uint32_t)n);
        } else {
            if (light) fused_chain_kernel<true, 2, true, uint64_t, true><<<grid, 256, 0, s>>>(f, out, (uint64_t)n);
            else fused_chain_kernel<true, 2, false, uint64_t, true><<<grid, 256, 0, s>>>(f, out, (uint64_t)n);
        }
    } else if (!vec) {
        if (light) NP_FC(false, 2, true); else NP_FC(false, 2, false);
    } else if (names_twice && small && fu == 2 && g_variant != 7001) {
        const unsigned grid = reduce_blocks ? reduce_blocks : grid_for(n / 4 + 1, 2, 0);
        if (light) fused_chain_kernel<true, 2, true, uint32_t, false, true><<<grid, 256, 0, s>>>(f, out, (uint32_t)n);
        else fused_chain_kernel<true, 2, false, uint32_t, false, true><<<grid, 256, 0, s>>>(f, out, (uint32_t)n);
    } else if (fu == 1) {
        if (light) NP_FC(true, 1, true); else NP_FC(true, 1, false);
    } else {
        if (light) NP_FC(true, 2, true); else NP_FC(true, 2, false);
    }
#undef NP_FC
    NP_LAUNCH_CHECK("fused_chain_kernel");
    if (sink >= 0 && !f.ticket) return np::fold_partials(sink, (const float *)partials.ptr, reduce_blocks, result);
    return NP_OK;
}

namespace np {
int g_bcast2d_off = 0;
int device_copy(void *dst, const void *src, size_t bytes) {
    const size_t n = bytes / 4;
    const unsigned grid = grid_for(n / 4 + 1, 2, 0);
    if (n < (size_t(1) << 31))
        copy_vec_kernel<uint32_t><<<grid, 256, 0, np::stream()>>>((const float *)src, (float *)dst, (uint32_t)(n / 4), (uint32_t)n);
    else
        copy_vec_kernel<uint64_t><<<grid, 256, 0, np::stream()>>>((const float *)src, (float *)dst, (uint64_t)(n / 4), (uint64_t)n);
    NP_LAUNCH_CHECK("copy_vec_kernel");
    return NP_OK;
}
}  // namespace np

extern "C" {

int np_fused_chain(const float *const *inputs, const int *input_kinds, int n_inputs,
                   const np_fused_op *ops, int n_ops, float *out, size_t rows, size_t cols) {
    return fused_chain_impl(inputs, input_kinds, n_inputs, ops, n_ops, out, rows, cols, -1);
}

int np_fused_chain_reduce(const float *const *inputs, const int *input_kinds, int n_inputs,
                          const np_fused_op *ops, int n_ops, int reduce_op, size_t rows, size_t cols,
                          float *host_out) {
    if (!host_out) return np::fail(NP_ERR_INVALID, "np_fused_chain_reduce: null output");
    if (reduce_op != NP_SUM && reduce_op != NP_PROD && reduce_op != NP_MIN && reduce_op != NP_MAX &&
        reduce_op != NP_MEAN)
        return np::fail(NP_ERR_INVALID, "np_fused_chain_reduce: unknown reduction %d", reduce_op);
    if (rows * cols == 0) return np::fail(NP_ERR_INVALID, "np_fused_chain_reduce: empty input");
    if (int rc = np::ensure_init()) return rc;
    np::ResultCall call;
    float *slot = call.slot;
    if (!slot) return NP_ERR_ALLOC;
    const int sink = reduce_op == NP_MEAN ? NP_SUM : reduce_op;
    if (int rc = fused_chain_impl(inputs, input_kinds, n_inputs, ops, n_ops, slot, rows, cols, sink))
        return rc;
    if (int rc = call.wait()) return rc;
    const float v = slot[0];
    *host_out = reduce_op == NP_MEAN ? v / (float)(rows * cols) : v;
    return NP_OK;
}

// Same, the result left on the device (1 float): nothing waits, the call can be issued back to back like any kernel.
int np_fused_chain_reduce_dev(const float *const *inputs, const int *input_kinds, int n_inputs, const np_fused_op *ops, int n_ops,
                              int reduce_op, size_t rows, size_t cols, float *dev_out) {
    if (!dev_out) return np::fail(NP_ERR_INVALID, "np_fused_chain_reduce_dev: null output");
    if (reduce_op != NP_SUM && reduce_op != NP_PROD && reduce_op != NP_MIN && reduce_op != NP_MAX && reduce_op != NP_MEAN)
        return np::fail(NP_ERR_INVALID, "np_fused_chain_reduce_dev: unknown reduction %d", reduce_op);
    if (rows * cols == 0) return np::fail(NP_ERR_INVALID, "np_fused_chain_reduce_dev: empty input");
    if (int rc = np::ensure_init()) return rc;
    const int sink = reduce_op == NP_MEAN ? NP_SUM : reduce_op;
    if (int rc = fused_chain_impl(inputs, input_kinds, n_inputs, ops, n_ops, dev_out, rows, cols, sink)) return rc;
    if (reduce_op == NP_MEAN) {
        const float count = (float)(rows * cols);
        return np_binary(NP_DIVIDE, dev_out, NP_FULL, &count, NP_HOST_SCALAR, dev_out, 1, 1, 0, 0);
    }
    return NP_OK;
}

int np_fused_chain_reduce_axis(const float *const *inputs, const int *input_kinds, int n_inputs,
                               const np_fused_op *ops, int n_ops, int reduce_op, size_t rows, size_t cols, int axis,
                               float *out) {
    if (!out) return np::fail(NP_ERR_INVALID, "np_fused_chain_reduce_axis: null output");
    if (axis != 0 && axis != 1) return np::fail(NP_ERR_INVALID, "np_fused_chain_reduce_axis: axis %d of a rows x cols chain", axis);
    if (reduce_op != NP_SUM && reduce_op != NP_PROD && reduce_op != NP_MIN && reduce_op != NP_MAX &&
        reduce_op != NP_MEAN)
        return np::fail(NP_ERR_INVALID, "np_fused_chain_reduce_axis: unknown reduction %d", reduce_op);
    if (rows * cols == 0) return np::fail(NP_ERR_INVALID, "np_fused_chain_reduce_axis: empty input");
    if (int rc = np::ensure_init()) return rc;
    const int sink = reduce_op == NP_MEAN ? NP_SUM : reduce_op;
    const float mean_div = reduce_op == NP_MEAN ? (float)(axis == 1 ? cols : rows) : 0.0f;
    if (fused_axis_shape_ok(rows, cols, axis))
        return fused_chain_impl(inputs, input_kinds, n_inputs, ops, n_ops, out, rows, cols, sink, axis, mean_div);
    // shapes the sink kernels would run mostly idle on: one fused pass into a temporary, then the
    // stand-alone axis reduction
    np::Scratch tmp;
    if (int rc = tmp.alloc(rows * cols * sizeof(float))) return rc;
    if (int rc = fused_chain_impl(inputs, input_kinds, n_inputs, ops, n_ops, (float *)tmp.ptr, rows, cols, -1)) return rc;
    return axis == 1 ? np_reduce_axis(reduce_op, (const float *)tmp.ptr, rows, cols, 1, out, 0)
                     : np_reduce_axis(reduce_op, (const float *)tmp.ptr, 1, rows, cols, out, 0);
}

int np_elementwise_set_variant(int variant) {
    if (variant == 8100 || variant == 8101) {   // 8100: broadcasts on the flat kernels as before round 4's 2-D forms (A/B), 8101: back
        np::g_bcast2d_off = variant == 8100;
        return NP_OK;
    }
    g_variant = variant;
    return NP_OK;
}

int np_binary(int op, const float *a, int a_kind, const float *b, int b_kind, float *out,
              size_t rows, size_t cols, unsigned flags, size_t body_end) {
    if (op < 0 || op >= NP_BINARY_OP_COUNT)
        return np::fail(NP_ERR_INVALID, "np_binary: unknown op %d", op);
    if (a_kind < NP_FULL || a_kind > NP_HOST_SCALAR || b_kind < NP_FULL || b_kind > NP_HOST_SCALAR)
        return np::fail(NP_ERR_INVALID, "np_binary: unknown operand kind (%d, %d)", a_kind, b_kind);
    if (rows == 0 || cols == 0) return NP_OK;
    if (!a || !b || !out) return np::fail(NP_ERR_INVALID, "np_binary: null pointer");
    if (int rc = np::ensure_init()) return rc;
    // host scalars travel by value (kernel argument); the kernels see kind SCALAR + null pointer
    float ha = 0.0f, hb = 0.0f;
    if (a_kind == NP_HOST_SCALAR) { ha = *a; a = nullptr; a_kind = NP_SCALAR; }
    if (b_kind == NP_HOST_SCALAR) { hb = *b; b = nullptr; b_kind = NP_SCALAR; }
    switch (op) {
        case NP_ADD: return dispatch_binary_quirk<NP_ADD>(a, a_kind, b, b_kind, out, rows, cols, flags, body_end, ha, hb);
        case NP_SUBTRACT: return dispatch_binary_quirk<NP_SUBTRACT>(a, a_kind, b, b_kind, out, rows, cols, flags, body_end, ha, hb);
        case NP_MULTIPLY: return dispatch_binary_quirk<NP_MULTIPLY>(a, a_kind, b, b_kind, out, rows, cols, flags, body_end, ha, hb);
        case NP_DIVIDE: return dispatch_binary_quirk<NP_DIVIDE>(a, a_kind, b, b_kind, out, rows, cols, flags, body_end, ha, hb);
        case NP_MOD: return dispatch_binary_quirk<NP_MOD>(a, a_kind, b, b_kind, out, rows, cols, flags, body_end, ha, hb);
        case NP_POW:
            // `$a ** 2`: x * x is the correctly rounded square (what powf returns in all but its rare
            // non-correctly-rounded cases) and has the same zeros / infinities / NaNs, at the cost of an add
            if (a_kind == NP_FULL && b_kind == NP_SCALAR && !b && hb == 2.0f)
                return dispatch_binary_quirk<NP_MULTIPLY>(a, NP_FULL, a, NP_FULL, out, rows, cols, 0u, 0, 0.0f, 0.0f);
            return dispatch_binary_quirk<NP_POW>(a, a_kind, b, b_kind, out, rows, cols, flags, body_end, ha, hb);
        case NP_EQUAL: return dispatch_binary_quirk<NP_EQUAL>(a, a_kind, b, b_kind, out, rows, cols, flags, body_end, ha, hb);
        case NP_NOT_EQUAL: return dispatch_binary_quirk<NP_NOT_EQUAL>(a, a_kind, b, b_kind, out, rows, cols, flags, body_end, ha, hb);
        case NP_GREATER: return dispatch_binary_quirk<NP_GREATER>(a, a_kind, b, b_kind, out, rows, cols, flags, body_end, ha, hb);
        case NP_GREATER_EQUAL: return dispatch_binary_quirk<NP_GREATER_EQUAL>(a, a_kind, b, b_kind, out, rows, cols, flags, body_end, ha, hb);
        case NP_LESS: return dispatch_binary_quirk<NP_LESS>(a, a_kind, b, b_kind, out, rows, cols, flags, body_end, ha, hb);
        case NP_LESS_EQUAL: return dispatch_binary_quirk<NP_LESS_EQUAL>(a, a_kind, b, b_kind, out, rows, cols, flags, body_end, ha, hb);
        case NP_MAXIMUM: return dispatch_binary_quirk<NP_MAXIMUM>(a, a_kind, b, b_kind, out, rows, cols, flags, body_end, ha, hb);
        case NP_MINIMUM: return dispatch_binary_quirk<NP_MINIMUM>(a, a_kind, b, b_kind, out, rows, cols, flags, body_end, ha, hb);
        default: return dispatch_binary_quirk<NP_ARCTAN2>(a, a_kind, b, b_kind, out, rows, cols, flags, body_end, ha, hb);
    }
}

int np_unary(int op, const float *in, float *out, size_t n, float p0, float p1) {
    if (op < 0 || op >= NP_UNARY_OP_COUNT)
        return np::fail(NP_ERR_INVALID, "np_unary: unknown op %d", op);
    if (n == 0) return NP_OK;
    if (!in || !out) return np::fail(NP_ERR_INVALID, "np_unary: null pointer");
    if (int rc = np::ensure_init()) return rc;
    // float_round: factor = powf(10, decimals), evaluated once with the host libm (the same
    // glibc powf the reference calls per element, double_math.c:255).
    if (op == NP_ROUND) p0 = powf(10.0f, p0);
#define NP_U(OP_) case OP_: return dispatch_unary<OP_>(in, out, n, p0, p1)
    switch (op) {
        NP_U(NP_ABS); NP_U(NP_SQRT); NP_U(NP_EXP); NP_U(NP_EXP2); NP_U(NP_EXPM1); NP_U(NP_LOG);
        NP_U(NP_LOG2); NP_U(NP_LOG10); NP_U(NP_LOG1P); NP_U(NP_LOGB); NP_U(NP_SIN); NP_U(NP_COS);
        NP_U(NP_TAN); NP_U(NP_ARCSIN); NP_U(NP_ARCCOS); NP_U(NP_ARCTAN); NP_U(NP_DEGREES);
        NP_U(NP_RADIANS); NP_U(NP_SINH); NP_U(NP_COSH); NP_U(NP_TANH); NP_U(NP_ARCSINH);
        NP_U(NP_ARCCOSH); NP_U(NP_ARCTANH); NP_U(NP_RINT); NP_U(NP_FIX); NP_U(NP_FLOOR);
        NP_U(NP_CEIL); NP_U(NP_TRUNC); NP_U(NP_SINC); NP_U(NP_NEGATE); NP_U(NP_SIGN); NP_U(NP_CLIP);
        NP_U(NP_ROUND); NP_U(NP_RSQRT); NP_U(NP_POSITIVE); NP_U(NP_RECIPROCAL);
        default: break;
    }
#undef NP_U
    return np::fail(NP_ERR_INVALID, "np_unary: unknown op %d", op);
}

int np_outer(const float *a, size_t m, const float *b, size_t n, float *out) {
    if (m == 0 || n == 0) return NP_OK;
    if (!a || !b || !out) return np::fail(NP_ERR_INVALID, "np_outer: null pointer");
    if (int rc = np::ensure_init()) return rc;
    const size_t total = m * n;
    const bool vec = n % 4 == 0 && aligned16(b) && aligned16(out);
    const unsigned grid = grid_for(vec ? total / 4 : total, vec ? 2 : 4, 0);
    hipStream_t s = np::stream();
    if (total < (size_t(1) << 31)) {
        if (vec)
            outer_kernel<true, uint32_t><<<grid, 256, 0, s>>>(a, b, out, (uint32_t)m, (uint32_t)n);
        else
            outer_kernel<false, uint32_t><<<grid, 256, 0, s>>>(a, b, out, (uint32_t)m, (uint32_t)n);
    } else {
        if (vec)
            outer_kernel<true, uint64_t><<<grid, 256, 0, s>>>(a, b, out, (uint64_t)m, (uint64_t)n);
        else
            outer_kernel<false, uint64_t><<<grid, 256, 0, s>>>(a, b, out, (uint64_t)m, (uint64_t)n);
    }
    NP_LAUNCH_CHECK("outer_kernel");
    return NP_OK;
}

int np_fill(float *dev_ptr, float value, size_t n) {
    if (n == 0) return NP_OK;
    if (!dev_ptr) return np::fail(NP_ERR_INVALID, "np_fill: null pointer");
    if (int rc = np::ensure_init()) return rc;
    // peel to a 16-byte boundary (row views start anywhere)
    size_t head = ((16 - ((uintptr_t)dev_ptr & 15u)) & 15u) / 4;
    if (head > n) head = n;
    const size_t nvec = (n - head) / 4;
    const unsigned grid = grid_for(nvec + 1, 2, 0);   // write-only stream: uncapped grid, non-temporal stores
    if (n < (size_t(1) << 31))
        fill_kernel<uint32_t><<<grid, 256, 0, np::stream()>>>(dev_ptr, value, (uint32_t)n,
                                                              (uint32_t)head, (uint32_t)nvec);
    else
        fill_kernel<uint64_t><<<grid, 256, 0, np::stream()>>>(dev_ptr, value, (uint64_t)n,
                                                              (uint64_t)head, (uint64_t)nvec);
    NP_LAUNCH_CHECK("fill_kernel");
    return NP_OK;
}

}  // extern "C"
