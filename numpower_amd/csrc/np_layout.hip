// Layout kernels of libnp_hip.so: transpose / axis permutation into a new
// contiguous buffer.  Bit-exact data movement, HBM-bound at 8 B/elem.
//
// Reference behaviour restated: NDArray_Transpose (src/manipulation.c:68-130) = copy, permute shape
// and strides, NDArray_ToContiguous (manipulation.c:381-421, a per-element strided copy; on the GPU
// a 2-D-only kernel, transposeCoalesced cuda_math.cu:136-150, launched with a fixed 16x16 grid and
// therefore only correct up to 256 x 256).
//
// Design: 2-D (and batched last-two-axes) transposes go through an LDS tile so that both the global
// read and the global write are 16-byte-per-lane coalesced row accesses; the LDS tile is padded by
// one float per row so the transposing writes spread over the banks.  Any other permutation uses a
// gather kernel (one output element per thread, coalesced writes).
#include <math.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "np_internal.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));
// a float4 that is only 4-byte aligned: global_load/store_dwordx4 take dword-aligned addresses, which
// is all a row of odd length offers
struct __attribute__((packed, aligned(4))) U4 { v4f v; };

constexpr int MAX_ND = 8;
// Where a plane sits: rows `in_pitch` floats apart in the input, output rows `out_pitch` apart; blockIdx.z runs over the
// batch axes (an N-D permutation whose plane axes are transposed: every other axis is a batch index with its own stride on
// either side).  The plain 2-D / batched case: pitches = cols / rows, one batch axis of rows * cols on both sides.
struct PlaneBatch {
    size_t in_pitch, out_pitch;
    unsigned nbatch;
    unsigned bshape[MAX_ND];
    size_t bin[MAX_ND], bout[MAX_ND];
};
__device__ __forceinline__ void plane_offsets(const PlaneBatch &p, unsigned batch, size_t &off_in, size_t &off_out) {
    off_in = off_out = 0;
#pragma unroll
    for (int d = MAX_ND - 1; d >= 0; --d) {
        if (d < (int)p.nbatch) {
            const unsigned q = batch / p.bshape[d], r = batch - q * p.bshape[d];
            off_in += (size_t)r * p.bin[d];
            off_out += (size_t)r * p.bout[d];
            batch = q;
        }
    }
}

// in: [batch][rows][cols] -> out: [batch][cols][rows].  TILE x TILE floats per workgroup; the LDS
// tile is padded to TILE+1 floats per row.  Global reads and writes are float4 per lane along
// rows: TILE*4 contiguous bytes per row segment on both sides (256 B at TILE 64, 512 B at 128).
// VEC = every row start 16-byte aligned: non-temporal on both sides, nothing is re-read.  Otherwise dword-aligned float4s
// (the matrix edge element by element) WITHOUT the hint: rows of odd length start anywhere in a 128-byte line, so the
// first and last line of every 512-byte segment are shared with the neighbouring tile, and the L2 has to merge the two
// partial writes — streaming stores take that away (8191 x 8193: 4.04 TB/s plain, 3.48 non-temporal, same box,
// profiles/r04/layout_ab_nt_on_ragged.log).
template <int TR, int TC, bool VEC>
__global__ __launch_bounds__(256) void transpose_tile_kernel(const float *__restrict__ in,
                                                             float *__restrict__ out, unsigned rows,
                                                             unsigned cols, unsigned tiles_x,
                                                             unsigned tiles_y, PlaneBatch pb, unsigned order = 0) {
    // a TR x TC tile of the input (TR rows, TC columns) becomes a TC x TR tile of the output
    constexpr int LDT = TR + 1;           // tile[c][r]
    constexpr int C4 = TC / 4;            // float4 columns per input tile row
    constexpr int RPP = 256 / C4;         // input tile rows covered per pass of the 256 threads
    constexpr int PASSES = TR / RPP;
    constexpr int OC4 = TR / 4;           // float4 columns per output tile row
    constexpr int ORPP = 256 / OC4;
    constexpr int OPASSES = TC / ORPP;
    extern __shared__ __attribute__((aligned(16))) float tile[];
    typedef v4f v4f_u __attribute__((aligned(4)));
    size_t off_in, off_out;
    plane_offsets(pb, blockIdx.z, off_in, off_out);
    const float *src = in + off_in;
    float *dst = out + off_out;
    const size_t ipitch = pb.in_pitch, opitch = pb.out_pitch;
    // diagonal tile order: workgroups that run at the same time (consecutive blockIdx.x) read
    // neighbouring column blocks AND write different column offsets of the output, instead of all
    // writing segments a power-of-two stride apart (HBM channel camping on 8192 x 8192 etc.)
    // (the grid is linear in x: a 10^7 x 3 matrix has more tile rows than gridDim.y allows)
    // order 1 (round 6; rows that are not whole lines share a line with the neighbouring tile at each end of a segment): the same
    // walk dealt to the XCDs in eight contiguous runs, as in transpose_walign_kernel below
    unsigned unit = blockIdx.x;
    if (order) {
        const unsigned W = tiles_x * tiles_y, q = W / 8, r = W % 8, xcd = unit % 8, idx = unit / 8;
        unit = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const unsigned bx = unit % tiles_x;
    const unsigned by = (unit / tiles_x + bx) % tiles_y;
    const unsigned r0 = by * TR, c0 = bx * TC;
    const unsigned tx4 = threadIdx.x % C4, ty = threadIdx.x / C4;

    // load: rows of the input tile; all PASSES loads are issued before the first LDS store
    v4f v[PASSES];
#pragma unroll
    for (int j = 0; j < PASSES; ++j) {
        const unsigned r = r0 + ty + RPP * j, c = c0 + 4 * tx4;
        v[j] = v4f{0, 0, 0, 0};
        if constexpr (VEC) {
            if (r < rows && c < cols) v[j] = __builtin_nontemporal_load((const v4f *)(src + (size_t)r * ipitch + c));
        } else if (r < rows && c + 3 < cols) {
            v[j] = *(const v4f_u *)(src + (size_t)r * ipitch + c);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (r < rows && c + k < cols) v[j][k] = src[(size_t)r * ipitch + c + k];
        }
    }
#pragma unroll
    for (int j = 0; j < PASSES; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) tile[(4 * tx4 + k) * LDT + ty + RPP * j] = v[j][k];
    __syncthreads();
    // store: rows of the output tile (= columns of the input tile)
    const unsigned otx4 = threadIdx.x % OC4, oty = threadIdx.x / OC4;
#pragma unroll
    for (int j = 0; j < OPASSES; ++j) {
        const unsigned oc = oty + ORPP * j;   // output row inside the tile (input column)
        const unsigned orow = c0 + oc, ocol = r0 + 4 * otx4;
        v4f w;
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = tile[oc * LDT + 4 * otx4 + k];
        if constexpr (VEC) {
            if (orow < cols && ocol < rows) __builtin_nontemporal_store(w, (v4f *)(dst + (size_t)orow * opitch + ocol));
        } else if (orow < cols && ocol + 3 < rows) {
            *(v4f_u *)(dst + (size_t)orow * opitch + ocol) = w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (orow < cols && ocol + k < rows) dst[(size_t)orow * opitch + ocol + k] = w[k];
        }
    }
}

// The same transpose for OUTPUT rows that do not start on the 128-byte line grid (out_pitch % 32 != 0: 8196 x 8192 ran 4.9
// TB/s where 8192^2 runs 6.3 — it is the write side that pays for misalignment, profiles/r04/transpose_alignment_probe.log:
// every 512-byte segment shares its first and last line with the neighbouring tile, two partial-line writes per segment).
// Here the tile's window in output row o is shifted left by s(o) = (o * rows) mod 32 floats, so that it starts ON a line:
// windows of consecutive tiles still tile the row, every store is a whole aligned float4 of whole lines (non-temporal),
// only an output row's two ends are partial.  In input terms the workgroup needs, for input column c, the input rows
// by * TR - s(c) ... + TR: it reads TR + 32 rows of its column block (the 32 extra ones are the neighbouring tile's, mostly
// cache hits) and files each element under its column's shift.  `out` must be 128-byte aligned (pool blocks are).
template <int TR, int TC>
__global__ __launch_bounds__(256) void transpose_walign_kernel(const float *__restrict__ in, float *__restrict__ out, unsigned rows,
                                                               unsigned cols, unsigned tiles_x, unsigned tiles_y, PlaneBatch pb,
                                                               unsigned order) {
    constexpr int C4 = TC / 4, RPP = 256 / C4, PASSES = (TR + 32) / RPP, HALF = PASSES / 2;
    constexpr int OC4 = TR / 4, ORPP = 256 / OC4, OPASSES = TC / ORPP;
    extern __shared__ __attribute__((aligned(16))) float tile[];
    typedef v4f v4f_u __attribute__((aligned(4)));
    size_t off_in, off_out;
    plane_offsets(pb, blockIdx.z, off_in, off_out);
    const float *src = in + off_in;
    float *dst = out + off_out;
    const size_t ipitch = pb.in_pitch, opitch = pb.out_pitch;
    // Which tile: order 0 = the diagonal walk of transpose_tile_kernel in launch order.  A tile reads 32 rows of its neighbour
    // above and shares partial lines with its neighbours left and right; in launch order those run on OTHER XCDs (workgroup b on
    // XCD b % 8) and every shared line is fetched into two L2s.  order 1 / 2 deal the tiles to the XCDs in eight contiguous runs
    // (np_sgemm.hip's tile_coords bijection) — 1: of the diagonal walk, 2: column strip by column strip (by fastest).
    unsigned unit = blockIdx.x;
    if (order) {
        const unsigned W = tiles_x * tiles_y, q = W / 8, r = W % 8, xcd = unit % 8, idx = unit / 8;
        unit = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const unsigned bx = order == 2 ? unit / tiles_y : unit % tiles_x;
    const unsigned by = order == 2 ? unit % tiles_y : (unit / tiles_x + bx) % tiles_y;
    const int r0 = (int)(by * TR);
    const unsigned c0 = bx * TC;
    const unsigned tx4 = threadIdx.x % C4, ty = threadIdx.x / C4;
    const unsigned rm = (unsigned)(opitch & 31u);   // (o * opitch) mod 32 = ((o mod 32) * (opitch mod 32)) mod 32
    // LDS row pitch: column cc's elements sit at cc * LDT + (position + shift(cc)), and shift(cc) advances by rm per column —
    // the banks a wave's lanes hit are cc * (LDT + rm) mod 64, so LDT + rm must be odd (with the usual TR + 1 and
    // rm = 31, 8191 x 8193, every lane of a wave hit the SAME bank: 3.4 TB/s)
    const int LDT = TR + ((rm & 1u) ? 2 : 1);
    const unsigned c = c0 + 4 * tx4;
    int shift[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) shift[k] = (int)(((c + k) * rm) & 31u);
    const bool vec_in = (ipitch & 3u) == 0 && (((uintptr_t)src) & 15u) == 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        v4f v[HALF];
#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            const int R = r0 - 32 + (int)ty + RPP * (h * HALF + j);   // input row
            v[j] = v4f{0, 0, 0, 0};
            if (R >= 0 && R < (int)rows) {
                const float *p = src + (size_t)R * ipitch + c;
                if (c + 3 < cols) v[j] = vec_in ? __builtin_nontemporal_load((const v4f *)p) : *(const v4f_u *)p;
                else
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (c + k < cols) v[j][k] = p[k];
            }
        }
#pragma unroll
        for (int j = 0; j < HALF; ++j) {
            const int rel = -32 + (int)ty + RPP * (h * HALF + j);   // R - r0
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int rr = rel + shift[k];
                if (rr >= 0 && rr < TR) tile[(4 * tx4 + k) * LDT + rr] = v[j][k];
            }
        }
    }
    __syncthreads();
    const unsigned otx4 = threadIdx.x % OC4, oty = threadIdx.x / OC4;
#pragma unroll
    for (int j = 0; j < OPASSES; ++j) {
        const unsigned oc = oty + ORPP * j;
        const unsigned o = c0 + oc;                              // output row
        if (o >= cols) continue;
        const int ocol = r0 - (int)((o * rm) & 31u) + 4 * (int)otx4;   // first of this lane's four output columns
        v4f w;
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = tile[oc * LDT + 4 * otx4 + k];
        float *q = dst + (size_t)o * opitch;
        if (ocol >= 0 && ocol + 3 < (int)rows) {
            __builtin_nontemporal_store(w, (v4f *)(q + ocol));   // 16-byte aligned by construction
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (ocol + k >= 0 && ocol + k < (int)rows) q[ocol + k] = w[k];
        }
    }
}

// Skinny matrices (one side <= 16): a 64 x 64 tile would be >= 75 % padding.  One thread per LONG-side
// index walks the short side: TALL (rows long): reads row r's `cols` floats, writes out[c][r]
// (coalesced per c); !TALL (cols long): reads in[r][c] (coalesced per r), writes the `rows` floats of
// out[c][.].  The strided side touches each cache line from consecutive instructions of the same
// wave, so it is served from the vector L1.
template <bool TALL>
__global__ __launch_bounds__(256) void transpose_skinny_kernel(const float *__restrict__ in,
                                                               float *__restrict__ out, size_t rows,
                                                               size_t cols) {
    const size_t plane = rows * cols;
    const float *src = in + (size_t)blockIdx.y * plane;
    float *dst = out + (size_t)blockIdx.y * plane;
    const size_t n_long = TALL ? rows : cols, n_short = TALL ? cols : rows;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_long; i += (size_t)gridDim.x * blockDim.x) {
        for (size_t k = 0; k < n_short; ++k) {
            if (TALL)
                dst[k * rows + i] = src[i * cols + k];
            else
                dst[i * rows + k] = src[k * cols + i];
        }
    }
}

// Two to four long rows (NCHW <-> NHWC with C = 2 .. 4, an (N, 3) point list <-> three coordinate arrays): neither side of
// transpose_skinny_kernel is a float4 stream — the strided side moves 8-16 bytes per lane.  Here a workgroup takes 2048
// positions of the R planes: float4 loads on the side it reads, a pass through LDS (lds[k][pos]: 128-bit accesses on the plane
// side, scalar ones on the interleaved side), float4 stores on the side it writes — both sides contiguous and non-temporal, all
// loads of the tile in flight before the first LDS store.  n (positions per plane) % 4 == 0, both pointers 16-byte aligned.
// TO_INTERLEAVED: in = [batch][R][n] -> out = [batch][n][R]; else the reverse.
template <int R, bool TO_INTERLEAVED>
__global__ __launch_bounds__(256) void interleave_kernel(const float *__restrict__ in, float *__restrict__ out, unsigned n) {
    constexpr unsigned P = 2048, PITCH = P + 4;
    __shared__ __attribute__((aligned(16))) float lds[R * PITCH];
    const unsigned t = threadIdx.x;
    const unsigned p0 = blockIdx.x * P;
    const unsigned np_ = n - p0 < P ? n - p0 : P;                 // positions of this tile (a multiple of 4)
    const unsigned nseg4 = np_ * R / 4;                           // float4s of its interleaved segment
    const size_t batch = blockIdx.y;
    const float *planes_in = in + batch * R * (size_t)n + p0;     // TO_INTERLEAVED: plane k at + k * n
    float *planes_out = out + batch * R * (size_t)n + p0;
    const float *seg_in = in + (batch * (size_t)n + p0) * R;
    float *seg_out = out + (batch * (size_t)n + p0) * R;
    if constexpr (TO_INTERLEAVED) {
        v4f v[R][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const unsigned pos = (t + 256u * u) * 4;
#pragma unroll
            for (int k = 0; k < R; ++k)
                if (pos < np_) v[k][u] = __builtin_nontemporal_load((const v4f *)(planes_in + (size_t)k * n + pos));
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const unsigned pos = (t + 256u * u) * 4;
#pragma unroll
            for (int k = 0; k < R; ++k)
                if (pos < np_) *(v4f *)&lds[k * PITCH + pos] = v[k][u];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2 * R; ++j) {
            const unsigned f = t + 256u * j;
            if (f < nseg4) {
                v4f w;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned e = 4 * f + i;
                    w[i] = lds[(e % R) * PITCH + e / R];
                }
                __builtin_nontemporal_store(w, (v4f *)(seg_out + (size_t)f * 4));
            }
        }
    } else {
        v4f v[2 * R];
#pragma unroll
        for (int j = 0; j < 2 * R; ++j) {
            const unsigned f = t + 256u * j;
            if (f < nseg4) v[j] = __builtin_nontemporal_load((const v4f *)(seg_in + (size_t)f * 4));
        }
#pragma unroll
        for (int j = 0; j < 2 * R; ++j) {
            const unsigned f = t + 256u * j;
            if (f < nseg4) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned e = 4 * f + i;
                    lds[(e % R) * PITCH + e / R] = v[j][i];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const unsigned pos = (t + 256u * u) * 4;
#pragma unroll
            for (int k = 0; k < R; ++k)
                if (pos < np_) __builtin_nontemporal_store(*(const v4f *)&lds[k * PITCH + pos], (v4f *)(planes_out + (size_t)k * n + pos));
        }
    }
}

template <bool TO_INTERLEAVED>
int launch_interleave(const float *in, float *out, size_t batch, unsigned R, size_t n) {
    const dim3 grid((unsigned)((n + 2047) / 2048), (unsigned)batch);
    hipStream_t s = np::stream();
    switch (R) {
        case 2: interleave_kernel<2, TO_INTERLEAVED><<<grid, 256, 0, s>>>(in, out, (unsigned)n); break;
        case 3: interleave_kernel<3, TO_INTERLEAVED><<<grid, 256, 0, s>>>(in, out, (unsigned)n); break;
        default: interleave_kernel<4, TO_INTERLEAVED><<<grid, 256, 0, s>>>(in, out, (unsigned)n); break;
    }
    NP_LAUNCH_CHECK("interleave_kernel");
    return NP_OK;
}

struct PermuteArgs {
    unsigned ndim;
    unsigned out_shape[MAX_ND];
    size_t in_stride[MAX_ND];   // element stride of the input axis that feeds output axis i
};

// W = floats moved per "element": 4 when the innermost axis is kept, contiguous and a multiple of 4
// long (shape / strides are then in float4 units; dword-aligned float4 accesses).
template <typename I, int W>
__global__ __launch_bounds__(256) void permute_gather_kernel(const float *__restrict__ in,
                                                             float *__restrict__ out, I n,
                                                             PermuteArgs a) {
    typedef v4f v4f_u __attribute__((aligned(4)));
    const I stride = (I)gridDim.x * blockDim.x;
    // (four elements per trip with all loads ahead of the stores — what helped permute_plane_kernel — was measured here and
    // lost 3-13 %: (256, 512, 512) (1, 0, 2) 5.05 -> 4.85 TB/s, every-2nd-column copy 3.6 -> 3.1; one element per trip stays)
    for (I i = (I)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        I rem = i;
        size_t off = 0;
#pragma unroll
        for (int d = MAX_ND - 1; d >= 0; --d) {
            if (d < (int)a.ndim) {
                const I q = rem / a.out_shape[d];
                off += (size_t)(rem - q * a.out_shape[d]) * a.in_stride[d];
                rem = q;
            }
        }
        if constexpr (W == 4)
            *(v4f_u *)(out + (size_t)i * 4) = *(const v4f_u *)(in + off * 4);
        else
            out[i] = in[off];
    }
}

// shared launcher: switches to float4 elements when the innermost output axis reads contiguous input
int launch_gather(const float *in, float *out, size_t n, PermuteArgs a) {
    const int last = (int)a.ndim - 1;
    // (short rows stay scalar: with 8-float rows the float4 form measured 2.7 vs 3.4 TB/s)
    bool vec4 = last >= 0 && a.in_stride[last] == 1 && a.out_shape[last] % 4 == 0 && a.out_shape[last] >= 32;
    for (int i = 0; vec4 && i < last; ++i) vec4 = a.in_stride[i] % 4 == 0;   // (negative strides wrap: still multiples of 4)
    if (vec4) {
        a.out_shape[last] /= 4;
        for (int i = 0; i < last; ++i) a.in_stride[i] = (size_t)((long long)a.in_stride[i] / 4);
        n /= 4;
    }
    size_t blocks = (n + 255) / 256;
    const size_t cap = (size_t)np::num_cus() * 16;
    if (blocks > cap) blocks = cap;
    hipStream_t s = np::stream();
    if (n < (size_t(1) << 31)) {
        if (vec4)
            permute_gather_kernel<uint32_t, 4><<<(unsigned)blocks, 256, 0, s>>>(in, out, (uint32_t)n, a);
        else
            permute_gather_kernel<uint32_t, 1><<<(unsigned)blocks, 256, 0, s>>>(in, out, (uint32_t)n, a);
    } else {
        if (vec4)
            permute_gather_kernel<uint64_t, 4><<<(unsigned)blocks, 256, 0, s>>>(in, out, (uint64_t)n, a);
        else
            permute_gather_kernel<uint64_t, 1><<<(unsigned)blocks, 256, 0, s>>>(in, out, (uint64_t)n, a);
    }
    NP_LAUNCH_CHECK("permute_gather_kernel");
    return NP_OK;
}

// General permutation whose output-fastest axis is NOT the input-fastest axis (the default transpose() of a 3-D array
// reverses all axes), or whose innermost axis stays but is only a few floats long (NHWC-like layouts: (64, 128, 1024, 8)
// with axes 1 and 2 swapped): the gather kernel above would read with a large stride per lane (0.4-0.5 TB/s) or in 32-byte
// runs (3.4 TB/s).  Here the plane spanned by the input axis that becomes output-fastest (A) and the input-fastest axis
// (B) goes through an LDS tile — reads run along B, writes along A, both in contiguous runs — an element of the plane being
// a run of E floats that is contiguous on both sides (E = 1, or the short kept inner axis), and every other axis a batch
// index.  The tile is ta x tb elements, ta * tb * E <= 4096 floats, and it is walked by LINEAR index in both phases
// (index -> (a, b, e) by multiply-high): planes whose extents are not multiples of 64 (100 x 100: round 3's fixed 64 x 64
// tiles ran 61 % of their lanes) cut into balanced tiles (50 x 50) keep every lane busy, and the runs stay 200+ bytes.
struct PlanePermuteArgs {
    unsigned A, B, E;               // plane extents (elements), floats per element
    size_t a_in, b_out;             // input stride of A, output stride of B (floats); B's input / A's output stride is E
    unsigned ta, tb, tiles_a, tiles_b;
    unsigned pitch;                 // LDS floats per a
    unsigned m_tbE, m_taE, m_E;     // floor(2^32 / d) + 1 for d = tb * E, ta * E, E (exact quotients below 2^12 ... 2^13)
    unsigned xcd_runs;              // 1: tiles dealt to the XCDs in eight contiguous runs
    PlaneBatch pb;
};
// floats per tile.  16384-float tiles (a 100 x 100 plane as ONE tile, 67 KB of LDS, two workgroups per CU) were measured in
// round 4 — before and after the loads were batched — and dropped: runs of 8 floats 5.5 -> 3.2 TB/s, runs of 4 4.7 -> 3.2,
// single floats +-0 except (200, 33, 77, 41) 2.7 -> 3.3 (profiles/r04/layout_plane_tile_ab.log)
constexpr unsigned kPlaneCap = 4096, kPlanePad = 512;

// W = 4: the element is a whole number of float4s (E % 4 == 0: an NHWC-like layout with 8 channels) — everything in `p` is then
// in float4 units and every access, global and LDS, is 128 bits wide.
template <int W>
__global__ __launch_bounds__(256) void permute_plane_kernel(const float *__restrict__ in_f, float *__restrict__ out_f, PlanePermuteArgs p) {
    typedef typename std::conditional<W == 4, v4f, float>::type T;
    extern __shared__ __attribute__((aligned(16))) float tile_f[];
    T *tile = (T *)tile_f;
    const T *in = (const T *)in_f;
    T *out = (T *)out_f;
    unsigned id = blockIdx.x;
    if (p.xcd_runs) {   // neighbouring tiles (they share the lines at the ends of their runs) on ONE XCD: np_sgemm.hip's tile_coords bijection
        const unsigned units = gridDim.x, q = units / 8, r = units % 8, xcd = id % 8, idx = id / 8;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const unsigned tb_i = id % p.tiles_b;
    id /= p.tiles_b;
    const unsigned ta_i = id % p.tiles_a;
    size_t off_in, off_out;
    plane_offsets(p.pb, id / p.tiles_a, off_in, off_out);
    const unsigned a0 = ta_i * p.ta, b0 = tb_i * p.tb;
    const unsigned na = p.A - a0 < p.ta ? p.A - a0 : p.ta, nb = p.B - b0 < p.tb ? p.B - b0 : p.tb;
    const unsigned tbE = p.tb * p.E, taE = p.ta * p.E, total = p.ta * tbE;
    // phase 1: (a, b, e) with (b, e) fastest — runs of nb * E contiguous floats per a.  Eight loads in flight per lane before
    // the first LDS store (one load -> store round trip per iteration left the kernel waiting on memory latency)
    constexpr int UNR = 8;
    for (unsigned base = threadIdx.x; base < total; base += 256 * UNR) {
        T v[UNR];
        unsigned at[UNR];
        bool ok[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const unsigned idx = base + 256u * u;
            const unsigned a = __umulhi(idx, p.m_tbE), rem = idx - a * tbE;
            const unsigned b = p.E == 1 ? rem : __umulhi(rem, p.m_E);
            ok[u] = idx < total && a < na && b < nb;
            at[u] = a * p.pitch + rem;
            v[u] = T{};
            if (ok[u]) v[u] = __builtin_nontemporal_load(in + off_in + (size_t)(a0 + a) * p.a_in + (size_t)b0 * p.E + rem);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (ok[u]) tile[at[u]] = v[u];
    }
    __syncthreads();
    // phase 2: (b, a, e) with (a, e) fastest — runs of na * E contiguous floats per b
    for (unsigned base = threadIdx.x; base < total; base += 256 * UNR) {
        T v[UNR];
        size_t to[UNR];
        bool ok[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const unsigned idx = base + 256u * u;
            const unsigned b = __umulhi(idx, p.m_taE), rem = idx - b * taE;
            const unsigned a = p.E == 1 ? rem : __umulhi(rem, p.m_E), e = rem - a * p.E;
            ok[u] = idx < total && a < na && b < nb;
            to[u] = off_out + (size_t)(b0 + b) * p.b_out + (size_t)a0 * p.E + rem;
            v[u] = ok[u] ? tile[a * p.pitch + b * p.E + e] : T{};
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u)
            if (ok[u]) __builtin_nontemporal_store(v[u], out + to[u]);
    }
}

// n x n identity: zeros with ones on the diagonal, one pass (NDArray_Identity, initializers.c:479-510,
// is NDArray_Zeros + a strided store loop).
__global__ __launch_bounds__(256) void identity_kernel(float *__restrict__ out, size_t n) {
    const size_t total = n * n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / n;
        out[i] = (i - r * n == r) ? 1.0f : 0.0f;
    }
}

// arange segments (see np_arange): element i of segment s is (float)(base + (i - first) * inc),
// evaluated in double — exact, every such value is representable (it lies on the float lattice of the
// segment's binade).
struct ArangeSeg {
    unsigned long long first;   // index of the segment's first element
    double base, inc;
};

__global__ __launch_bounds__(256) void arange_kernel(float *__restrict__ out, const ArangeSeg *__restrict__ segs,
                                                     unsigned nsegs, size_t n, size_t per_block) {
    // each workgroup owns a contiguous range: the segment is looked up once (binary search) and then
    // only walked forward
    const size_t b0 = (size_t)blockIdx.x * per_block;
    size_t b1 = b0 + per_block;
    if (b1 > n) b1 = n;
    size_t i = b0 + threadIdx.x;
    if (i >= b1) return;
    unsigned lo = 0, hi = nsegs;             // last segment with first <= i
    while (hi - lo > 1) {
        const unsigned mid = (lo + hi) / 2;
        if (segs[mid].first <= i) lo = mid; else hi = mid;
    }
    unsigned long long first = segs[lo].first, next_first = lo + 1 < nsegs ? segs[lo + 1].first : ~0ull;
    double base = segs[lo].base, inc = segs[lo].inc;
    for (; i < b1; i += 256) {
        while (i >= next_first) {
            ++lo;
            first = next_first;
            base = segs[lo].base;
            inc = segs[lo].inc;
            next_first = lo + 1 < nsegs ? segs[lo + 1].first : ~0ull;
        }
        const unsigned long long k = i - first;
        const double kd = k < 0x100000000ull ? (double)(unsigned)k : (double)k;   // one v_cvt for the common case
        out[i] = (float)__fma_rn(kd, inc, base);   // exact: the value lies on the float lattice
    }
}

inline bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

int g_tile = 0;   // 0 = default, else 64 / 128 (np_layout_set_variant)
int g_plane_order = -1;  // permute_plane_kernel: -1 = default, 0 = launch order, 1 = XCD runs (np_layout_set_variant(17020 + k), 17023 = default)
int g_tile_order = -1;   // transpose_tile_kernel's tile order: -1 = by alignment, 0 = launch order, 1 = XCD runs (np_layout_set_variant(17010 + k), 17013 = by alignment)
// Write-aligned transposes with fewer 128 x 128 tiles than this take 64 x 64 tiles (0 = never; np_layout_set_variant(7000 + N) sets it): at two
// workgroups of 67 KB LDS per CU, 1089 tiles (4099^2) are 2.1 resident rounds and a third of the last one idles; 4225 small tiles at eight
// per CU end more evenly — 4099^2 4.80 -> 5.07 TB/s, 5000 x 4099 (1320 tiles) +1 %; from ~2000 tiles up the small tiles LOSE (8191 x 8193 -2 %,
// 6001 x 6003 -14 %, 12345 x 6789 -19 %: they read 96 rows for every 64 they write) — profiles/r05/walign_tile_ab.log.
size_t g_walign64_below_tiles = 1200;
// transpose_walign_kernel's tile order: -1 = by size (below), 0 / 1 / 2 forced (np_layout_set_variant(17000 + k), 17003 = by size).
// Dealing the diagonal walk to the XCDs in eight contiguous runs lets a tile find its neighbour's 32 halo rows and the lines it shares
// left and right in its OWN XCD's L2: 4099^2 5.08 -> 6.00 TB/s, 6001 x 6003 4.88 -> 5.71, 12345 x 6789 5.22 -> 5.75 — but 8191 x 8193
// 5.74 -> 5.58 and 8193 x 8191 5.75 -> 5.18 (the eight XCDs then work 1024 rows = 2^25 bytes apart; profiles/r06/walign_order_ab.log).
// Taken below 6 * 10^7 elements per plane (4000 tiles of 128 x 128), where every shape of the sweep gains.
int g_walign_order = -1;
static inline unsigned walign_order(size_t rows, size_t cols) {
    return g_walign_order >= 0 ? (unsigned)g_walign_order : (rows * cols < (size_t)60'000'000 ? 1u : 0u);
}

template <int TR, int TC>
int launch_transpose(const float *in, float *out, size_t batch, size_t rows, size_t cols, bool vec, const PlaneBatch &pb) {
    const size_t tiles_x = (cols + TC - 1) / TC, tiles_y = (rows + TR - 1) / TR;
    if (tiles_x * tiles_y > 0x7fffffffu) return np::fail(NP_ERR_INVALID, "np_transpose2d: too many tiles");
    const dim3 grid((unsigned)(tiles_x * tiles_y), 1, (unsigned)batch);
    constexpr size_t lds = (size_t)TC * (TR + 1) * sizeof(float);
    if (lds > 64 * 1024) {   // above the 64 KB default limit of dynamic LDS
        static bool attr_set[64] = {false};   // per device: the attribute belongs to the device's copy of the kernel
        int dev = 0;
        NP_HIP_CHECK(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            NP_HIP_CHECK(hipFuncSetAttribute((const void *)transpose_tile_kernel<TR, TC, true>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            NP_HIP_CHECK(hipFuncSetAttribute((const void *)transpose_tile_kernel<TR, TC, false>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
    }
    // (the XCD runs that help the write-aligned kernel do NOT help here — input rows off the line grid, output rows on it: 8000 x 8001
    //  4.24 -> 4.35 TB/s, 4096 x 4099 5.12 -> 4.49, 8192 x 10001 / 16000 x 5001 / 3200 x 30001 -1 %, profiles/r06/tile_order_ab.log —
    //  so the launch order stays; np_layout_set_variant(17011) forces the runs for A/B)
    const unsigned order = g_tile_order > 0 ? 1u : 0u;
    if (vec)
        transpose_tile_kernel<TR, TC, true><<<grid, 256, lds, np::stream()>>>(in, out, (unsigned)rows, (unsigned)cols,
                                                                             (unsigned)tiles_x, (unsigned)tiles_y, pb, order);
    else
        transpose_tile_kernel<TR, TC, false><<<grid, 256, lds, np::stream()>>>(in, out, (unsigned)rows, (unsigned)cols,
                                                                              (unsigned)tiles_x, (unsigned)tiles_y, pb, order);
    NP_LAUNCH_CHECK("transpose_tile_kernel");
    return NP_OK;
}

// Pitched 2-D copy: `rows` rows of `width` floats, row starts dst_pitch / src_pitch floats apart — what
// concatenation along an inner axis is (each input is a slab of the result's rows).  Rows are walked
// in float4 units when the width allows (dword-aligned accesses: any pitch, any base), linearised
// over (row, unit) so narrow and wide rows fill the machine alike.
template <int V>
__global__ __launch_bounds__(256) void copy2d_kernel(float *__restrict__ dst, size_t dst_pitch, const float *__restrict__ src,
                                                     size_t src_pitch, unsigned per_row, size_t units) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t u = (size_t)blockIdx.x * 256 + threadIdx.x; u < units; u += stride) {
        const size_t r = u / per_row;
        const size_t c = (u - r * per_row) * V;
        if constexpr (V == 4) {
            const U4 x = *(const U4 *)(src + r * src_pitch + c);
            *(U4 *)(dst + r * dst_pitch + c) = x;
        } else {
            dst[r * dst_pitch + c] = src[r * src_pitch + c];
        }
    }
}

// rows x cols planes (rows `pb.in_pitch` apart) -> cols x rows planes (rows `pb.out_pitch` apart), `batch` of them
int transpose_planes(const float *in, float *out, size_t batch, size_t rows, size_t cols, const PlaneBatch &pb) {
    const bool vec = rows % 4 == 0 && cols % 4 == 0 && aligned16(in) && aligned16(out) && pb.in_pitch % 4 == 0 && pb.out_pitch % 4 == 0;
    bool vec_batch = vec;
    for (unsigned d = 0; d < pb.nbatch; ++d) vec_batch = vec_batch && pb.bin[d] % 4 == 0 && pb.bout[d] % 4 == 0;
    // 128 x 128 tiles (512-byte row segments) for large matrices, 64 x 64 when that would leave
    // CUs without work
    int tile = (g_tile == 64 || g_tile == 128) ? g_tile : 0;   // (variant 1: default tiles, but never the write-aligned form; 3, 4: np_permute's A/B switches)
    if (tile == 0)
        tile = (((rows + 127) / 128) * ((cols + 127) / 128) * batch >= (size_t)np::num_cus() * 4) ? 128 : 64;
    // Rectangular tiles (the kernel takes any TR x TC) were measured in round 2 — 256x64, 64x256, 128x64, 64x128, 256x32,
    // 32x256 (profiles/r02/transpose_rect_ab.log): none beats 128 x 128.  What counts is the length of the READ
    // segments (64x256: 1 KiB reads, 256 B writes = 128x128's 5.74 TB/s at 65536 x 4096; 256x64: 256 B reads, 1 KiB
    // writes = 5.16), so only the two square tiles are instantiated.  A probe with the LDS round trip taken out (same
    // loads and stores, wrong values) runs at the same 5.65 TB/s: the LDS transpose is hidden, the rate is what 512-byte
    // segments at two tiles per CU get from HBM — and with no LDS allocated (more tiles in flight) it drops to 5.35.
    // Round 4 tried a persistent form that issues the NEXT tile's loads before it pushes the current one through LDS (more
    // bytes in flight): 65536 x 4096 5.69 -> 5.33 TB/s, 16384^2 5.31 -> 5.06, 8192^2 equal (profiles/r04/transpose_pipelined_ab.log)
    // — memory-level parallelism is not what this kernel lacks.
    // output rows off the 128-byte line grid: the write-aligned form (variant 1: off)
    bool walign = g_tile != 1 && g_tile != 64 && g_tile != 128 && tile == 128 && pb.out_pitch % 32 != 0 && ((uintptr_t)out & 127u) == 0 && rows >= 256 && cols >= 64 &&
                  rows + 31 < 0x7fffffffu;
    for (unsigned d = 0; d < pb.nbatch; ++d) walign = walign && (pb.bshape[d] == 1 || pb.bout[d] % 32 == 0);
    // 64 x 64 write-aligned tiles: variant 7 = always; by default for the smallest matrices that get here (g_walign64_below_tiles)
    if (walign && (g_tile == 7 || (g_walign64_below_tiles && ((cols + 127) / 128) * ((rows + 31 + 127) / 128) * batch < g_walign64_below_tiles))) {
        const size_t tiles_x = (cols + 63) / 64, tiles_y = (rows + 31 + 63) / 64;
        if (tiles_x * tiles_y <= 0x7fffffffu) {
            constexpr size_t lds = (size_t)64 * 66 * sizeof(float);
            transpose_walign_kernel<64, 64><<<dim3((unsigned)(tiles_x * tiles_y), 1, (unsigned)batch), 256, lds, np::stream()>>>(
                in, out, (unsigned)rows, (unsigned)cols, (unsigned)tiles_x, (unsigned)tiles_y, pb, walign_order(rows, cols));
            NP_LAUNCH_CHECK("transpose_walign_kernel");
            return NP_OK;
        }
    }
    if (walign) {
        const size_t tiles_x = (cols + 127) / 128, tiles_y = (rows + 31 + 127) / 128;
        if (tiles_x * tiles_y <= 0x7fffffffu) {
            constexpr size_t lds = (size_t)128 * 130 * sizeof(float);
            static bool attr_set[64] = {false};   // per device, as in launch_transpose
            int dev = 0;
            NP_HIP_CHECK(hipGetDevice(&dev));
            if (dev < 0 || dev >= 64 || !attr_set[dev]) {
                NP_HIP_CHECK(hipFuncSetAttribute((const void *)transpose_walign_kernel<128, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                if (dev >= 0 && dev < 64) attr_set[dev] = true;
            }
            transpose_walign_kernel<128, 128><<<dim3((unsigned)(tiles_x * tiles_y), 1, (unsigned)batch), 256, lds, np::stream()>>>(
                in, out, (unsigned)rows, (unsigned)cols, (unsigned)tiles_x, (unsigned)tiles_y, pb, walign_order(rows, cols));
            NP_LAUNCH_CHECK("transpose_walign_kernel");
            return NP_OK;
        }
    }
    if (tile == 128) return launch_transpose<128, 128>(in, out, batch, rows, cols, vec_batch, pb);
    return launch_transpose<64, 64>(in, out, batch, rows, cols, vec_batch, pb);
}

}  // namespace

extern "C" {

int np_layout_set_variant(int variant) {
    if (variant >= 7000 && variant < 17000) {
        g_walign64_below_tiles = (size_t)(variant - 7000);
        return NP_OK;
    }
    if (variant >= 17020 && variant <= 17023) {
        g_plane_order = variant == 17023 ? -1 : variant - 17020;
        return NP_OK;
    }
    if (variant >= 17010 && variant <= 17013) {
        g_tile_order = variant == 17013 ? -1 : variant - 17010;
        return NP_OK;
    }
    if (variant >= 17000 && variant <= 17003) {   // write-aligned transposes: 0 = diagonal walk in launch order, 1 / 2 = dealt to the XCDs in runs, 3 = by size
        g_walign_order = variant == 17003 ? -1 : variant - 17000;
        return NP_OK;
    }
    g_tile = variant;
    return NP_OK;
}

int np_transpose2d(const float *in, float *out, size_t batch, size_t rows, size_t cols) {
    if (batch == 0 || rows == 0 || cols == 0) return NP_OK;
    if (!in || !out) return np::fail(NP_ERR_INVALID, "np_transpose2d: null pointer");
    if (in == out) return np::fail(NP_ERR_INVALID, "np_transpose2d: in-place transpose is not supported");
    if (rows > 0x7fffffffu || cols > 0x7fffffffu || batch > 65535)
        return np::fail(NP_ERR_INVALID, "np_transpose2d: dimension too large");
    if (int rc = np::ensure_init()) return rc;
    // two to four long rows or columns: the float4 interleave (variant 5: off, the skinny kernel below as before round 5)
    if (g_tile == 0 && aligned16(in) && aligned16(out)) {
        if (rows >= 2 && rows <= 4 && cols % 4 == 0 && cols >= 1024) return launch_interleave<true>(in, out, batch, (unsigned)rows, cols);
        if (cols >= 2 && cols <= 4 && rows % 4 == 0 && rows >= 1024) return launch_interleave<false>(in, out, batch, (unsigned)cols, rows);
    }
    if ((rows <= 16 || cols <= 16) && (g_tile == 0 || g_tile == 5)) {
        const bool tall = cols <= rows;
        const size_t n_long = tall ? rows : cols;
        size_t blocks = (n_long + 255) / 256;
        const size_t cap = (size_t)np::num_cus() * 32;
        if (blocks > cap) blocks = cap;
        const dim3 grid((unsigned)blocks, (unsigned)batch);
        if (tall)
            transpose_skinny_kernel<true><<<grid, 256, 0, np::stream()>>>(in, out, rows, cols);
        else
            transpose_skinny_kernel<false><<<grid, 256, 0, np::stream()>>>(in, out, rows, cols);
        NP_LAUNCH_CHECK("transpose_skinny_kernel");
        return NP_OK;
    }
    PlaneBatch pb{};
    pb.in_pitch = cols;
    pb.out_pitch = rows;
    pb.nbatch = 1;
    for (int d = 0; d < MAX_ND; ++d) {
        pb.bshape[d] = 1;
        pb.bin[d] = pb.bout[d] = 0;
    }
    pb.bshape[0] = (unsigned)batch;
    pb.bin[0] = pb.bout[0] = rows * cols;
    return transpose_planes(in, out, batch, rows, cols, pb);
}

int np_permute(const float *in, float *out, int ndim, const int *host_shape, const int *host_perm) {
    if (ndim < 0 || ndim > MAX_ND) return np::fail(NP_ERR_INVALID, "np_permute: ndim %d not in 0..%d", ndim, MAX_ND);
    if (ndim > 0 && (!host_shape || !host_perm)) return np::fail(NP_ERR_INVALID, "np_permute: null shape/perm");
    size_t n = 1;
    bool seen[MAX_ND] = {false};
    for (int i = 0; i < ndim; ++i) {
        if (host_shape[i] < 0) return np::fail(NP_ERR_INVALID, "np_permute: negative dimension");
        n *= (size_t)host_shape[i];
        const int p = host_perm[i];
        if (p < 0 || p >= ndim) return np::fail(NP_ERR_INVALID, "axes don't match array");
        if (seen[p]) return np::fail(NP_ERR_INVALID, "repeated axis in transpose");
        seen[p] = true;
    }
    if (n == 0) return NP_OK;
    if (!in || !out) return np::fail(NP_ERR_INVALID, "np_permute: null pointer");
    if (int rc = np::ensure_init()) return rc;

    // Simplify first: drop axes of extent 1 and fuse input axes that stay adjacent and in order in the
    // output.  NCHW -> NHWC, (N, C, H, W) perm (0, 2, 3, 1), becomes (N, C, H*W) perm (0, 2, 1): a
    // batched 2-D transpose of C x HW planes instead of a 4-D gather.
    int fshape[MAX_ND], fperm[MAX_ND];
    {
        int pos[MAX_ND];                       // output position of each input axis
        for (int o = 0; o < ndim; ++o) pos[host_perm[o]] = o;
        int group_of[MAX_ND], ngroups = 0, first_pos[MAX_ND];
        int prev = -1;                         // previous kept input axis
        for (int i = 0; i < ndim; ++i) {
            if (host_shape[i] == 1) {
                group_of[i] = -1;
                continue;
            }
            // same group as the previous kept axis iff every output slot between them holds an extent-1 axis
            bool fuse = prev >= 0 && pos[i] > pos[prev];
            if (fuse)
                for (int o = pos[prev] + 1; o < pos[i]; ++o) fuse = fuse && host_shape[host_perm[o]] == 1;
            if (fuse) {
                group_of[i] = ngroups - 1;
                fshape[ngroups - 1] *= host_shape[i];
            } else {
                group_of[i] = ngroups;
                fshape[ngroups] = host_shape[i];
                first_pos[ngroups] = pos[i];
                ++ngroups;
            }
            prev = i;
        }
        // order the groups by output position
        int order[MAX_ND];
        for (int g = 0; g < ngroups; ++g) order[g] = g;
        for (int a = 0; a < ngroups; ++a)
            for (int b = a + 1; b < ngroups; ++b)
                if (first_pos[order[b]] < first_pos[order[a]]) {
                    const int t = order[a];
                    order[a] = order[b];
                    order[b] = t;
                }
        for (int o = 0; o < ngroups; ++o) fperm[o] = order[o];
        ndim = ngroups;
        host_shape = fshape;
        host_perm = fperm;
    }

    bool identity = true;
    for (int i = 0; i < ndim; ++i) identity = identity && host_perm[i] == i;
    if (identity) return np_memcpy_d2d(out, in, n * sizeof(float));
    // leading axes untouched, last two swapped -> batched tile transpose
    bool last_two = ndim >= 2 && host_perm[ndim - 1] == ndim - 2 && host_perm[ndim - 2] == ndim - 1;
    for (int i = 0; i + 2 < ndim; ++i) last_two = last_two && host_perm[i] == i;
    if (last_two) {
        size_t batch = 1;
        for (int i = 0; i + 2 < ndim; ++i) batch *= (size_t)host_shape[i];
        if (batch <= 65535 && (size_t)host_shape[ndim - 2] <= (size_t)65535 * 64)
            return np_transpose2d(in, out, batch, (size_t)host_shape[ndim - 2], (size_t)host_shape[ndim - 1]);
    }
    size_t in_strides[MAX_ND];
    size_t s = 1;
    for (int i = ndim - 1; i >= 0; --i) {
        in_strides[i] = s;
        s *= (size_t)host_shape[i];
    }
    // The plane path (permute_plane_kernel / the float4 tile transpose): the output-fastest axis is not the input-fastest
    // one, or the innermost axis is kept but short (E floats, < 32) and the axes in front of it are reordered — then the
    // "elements" of the plane are runs of E floats.
    {
        int nd = ndim;
        unsigned E = 1;
        // (runs shorter than 32 floats; from 32 up the float4 gather has 128-byte runs of its own.  (64, 128, 1024, 8) with axes
        // 1, 2 swapped: gather 3.2 TB/s, plane kernel 5.5; (128, 128, 128, 16) (2, 1, 0, 3): 3.9 -> 5.5; profiles/r04/layout_sweep.log)
        if (nd >= 3 && host_perm[nd - 1] == nd - 1 && host_shape[nd - 1] < 32 && (g_tile == 0 || g_tile == 6)) {
            E = (unsigned)host_shape[nd - 1];
            --nd;   // the remaining axes permute elements of E floats; host_perm[0 .. nd) is a permutation of 0 .. nd - 1
        }
        if (nd >= 2 && host_perm[nd - 1] != nd - 1 && (g_tile == 0 || g_tile == 6)) {
            const int ax_a = host_perm[nd - 1], ax_b = nd - 1;   // input axes: A becomes output-fastest, B is input-fastest
            size_t out_strides[MAX_ND];   // output stride of each OUTPUT axis (floats)
            size_t t = 1;
            for (int i = ndim - 1; i >= 0; --i) {
                out_strides[i] = t;
                t *= (size_t)host_shape[host_perm[i]];
            }
            PlanePermuteArgs p{};
            p.A = (unsigned)host_shape[ax_a];
            p.B = (unsigned)host_shape[ax_b];
            p.E = E;
            p.a_in = in_strides[ax_a];
            p.b_out = 0;
            PlaneBatch &pb = p.pb;
            pb.nbatch = 0;
            size_t batch = 1;
            for (int i = 0; i < MAX_ND; ++i) {
                pb.bshape[i] = 1;
                pb.bin[i] = pb.bout[i] = 0;
            }
            for (int o = 0; o < nd; ++o) {          // o: output axis, fed by input axis host_perm[o]
                const int ia = host_perm[o];
                if (ia == ax_b) {
                    p.b_out = out_strides[o];
                } else if (ia != ax_a) {
                    pb.bshape[pb.nbatch] = (unsigned)host_shape[ia];
                    pb.bin[pb.nbatch] = in_strides[ia];
                    pb.bout[pb.nbatch] = out_strides[o];
                    ++pb.nbatch;
                    batch *= (size_t)host_shape[ia];
                }
            }
            pb.in_pitch = p.a_in;
            pb.out_pitch = p.b_out;
            // large planes of single floats: the float4 tile transpose of np_transpose2d, pitched (512-byte row segments)
            if (E == 1 && p.A >= 64 && p.B >= 64 && batch <= 65535)
                return transpose_planes(in, out, batch, p.A, p.B, pb);
            // elements that are whole float4s move as float4s (variant 6: off): every stride below is then a multiple of 4 floats
            const bool w4 = E % 4 == 0 && aligned16(in) && aligned16(out) && g_tile != 6;
            if (w4) {
                E /= 4;
                p.E = E;
                p.a_in /= 4;
                p.b_out /= 4;
                for (unsigned d = 0; d < pb.nbatch; ++d) {
                    pb.bin[d] /= 4;
                    pb.bout[d] /= 4;
                }
                pb.in_pitch = p.a_in;
                pb.out_pitch = p.b_out;
            }
            // balanced tiles of at most 4096 floats, as square as the extents allow, runs of >= 64 floats where they can be
            const unsigned cap = w4 ? kPlaneCap / 4 : kPlaneCap, pad = w4 ? kPlanePad / 4 : kPlanePad;
            unsigned side = 64;
            while (side > 1 && (size_t)side * side * E > cap) --side;
            unsigned tb_max = p.B < side ? p.B : side;
            if (p.A < side) {   // a short A leaves room for a longer B run
                const size_t room = cap / ((size_t)p.A * E);
                tb_max = (unsigned)(room > 256 ? 256 : room);
                if (tb_max > p.B) tb_max = p.B;
                if (tb_max < 1) tb_max = 1;
            }
            p.tiles_b = (p.B + tb_max - 1) / tb_max;
            p.tb = (p.B + p.tiles_b - 1) / p.tiles_b;
            // LDS row pitch: consecutive a (phase 2's lanes) must not share banks — odd for single floats, else
            // the run length past a multiple of 64 floats
            const unsigned row = p.tb * E;
            p.pitch = (E == 1 || w4) ? (row | 1u) : ((row + 63) / 64 * 64 + E);   // (float4 units: an odd pitch spreads 8 lanes over all banks)
            unsigned ta_max = cap / row < (cap + pad) / p.pitch ? cap / row : (cap + pad) / p.pitch;
            if (ta_max > 256) ta_max = 256;
            if (ta_max > p.A) ta_max = p.A;
            if (ta_max < 1) ta_max = 1;
            p.tiles_a = (p.A + ta_max - 1) / ta_max;
            p.ta = (p.A + p.tiles_a - 1) / p.tiles_a;
            auto magic = [](unsigned d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull / d) + 1); };
            p.m_tbE = magic(p.tb * E);
            p.m_taE = magic(p.ta * E);
            p.m_E = magic(E);
            const size_t blocks = (size_t)p.tiles_a * p.tiles_b * batch;
            if (blocks <= 0x7fffffffu && (size_t)p.ta * p.pitch <= cap + pad && p.tb * E > 1 && p.ta * E > 1) {
                const size_t lds = ((size_t)p.ta * p.pitch + 3) / 4 * 4 * sizeof(float) * (w4 ? 4 : 1);
                // tiles dealt to the XCDs in runs (round 6) where elements are single floats or odd runs of them — neighbouring tiles then
                // share the lines at the ends of their runs in ONE L2: (200, 33, 77, 41) -> (3, 1, 2, 0) 2.69 -> 3.40 TB/s, (1000, 1000, 5, 20)
                // reversed 2.79 -> 2.91, 100^4 and (50, 60, 70, 80) unchanged; float4 elements (runs that ARE whole lines) lose 5-20 % that
                // way and keep the launch order (profiles/r06/plane_order_ab.log)
                p.xcd_runs = g_plane_order >= 0 ? (unsigned)g_plane_order : (w4 ? 0u : 1u);
                if (w4)
                    permute_plane_kernel<4><<<(unsigned)blocks, 256, lds, np::stream()>>>(in, out, p);
                else
                    permute_plane_kernel<1><<<(unsigned)blocks, 256, lds, np::stream()>>>(in, out, p);
                NP_LAUNCH_CHECK("permute_plane_kernel");
                return NP_OK;
            }
        }
    }
    // general gather
    PermuteArgs a;
    a.ndim = (unsigned)ndim;
    for (int i = 0; i < MAX_ND; ++i) {
        a.out_shape[i] = 1;
        a.in_stride[i] = 0;
    }
    for (int i = 0; i < ndim; ++i) {
        a.out_shape[i] = (unsigned)host_shape[host_perm[i]];
        a.in_stride[i] = in_strides[host_perm[i]];
    }
    return launch_gather(in, out, n, a);
}

int np_identity(float *out, size_t n) {
    if (n == 0) return NP_OK;
    if (!out) return np::fail(NP_ERR_INVALID, "np_identity: null pointer");
    if (int rc = np::ensure_init()) return rc;
    size_t blocks = (n * n + 255) / 256;
    const size_t cap = (size_t)np::num_cus() * 32;
    if (blocks > cap) blocks = cap;
    identity_kernel<<<(unsigned)blocks, 256, 0, np::stream()>>>(out, n);
    NP_LAUNCH_CHECK("identity_kernel");
    return NP_OK;
}

// NDArray_Arange (initializers.c:818-841) is a recurrence: x[0] = (float)start,
// x[i] = (float)((double)x[i-1] + step) — every element carries the rounding of all earlier ones, and
// past 2^24 the sequence of arange(n) simply stops moving.  It is still piecewise arithmetic: while
// x stays inside one binade [2^k, 2^(k+1)) the float lattice has a fixed spacing u, (double)x + step
// rounds the same way for every x on that lattice, so the increment is a constant multiple of u
// (after at most one step when step sits exactly between two lattice points: ties-to-even then
// depends on the parity x starts with).  The host walks the recurrence only across binade borders —
// a few scalar steps each — and describes everything in between as (first index, base, increment)
// segments; the kernel evaluates them in double, which is exact.  Result: bit-identical to the
// reference loop at 4 B/elem of HBM writes instead of a 4-cycle dependent add per element.
int np_arange(float *out, double start, double step, size_t n) {
    if (n == 0) return NP_OK;
    if (!out) return np::fail(NP_ERR_INVALID, "np_arange: null pointer");
    if (int rc = np::ensure_init()) return rc;
    auto next = [step](float x) { return (float)((double)x + step); };
    auto binade = [](float x) {   // identifies sign + exponent; zero and non-finite values get their own
        int e = 0;
        if (x == 0.0f || !(x - x == 0.0f)) return 1 << 20;
        (void)frexpf(x, &e);
        return x < 0.0f ? -(e + 1024) : (e + 1024);
    };
    std::vector<ArangeSeg> segs;
    size_t i = 0;
    float x = (float)start;
    while (i < n) {
        if (!(x - x == 0.0f)) {   // inf / NaN: stays inf or turns NaN; check one step, then it is stuck
            const float y = next(x);
            segs.push_back(ArangeSeg{(unsigned long long)i, (double)x, 0.0});
            if (memcmp(&x, &y, sizeof(float)) == 0 || (x != x && y != y)) break;   // covers the rest
            ++i;
            x = y;
            continue;
        }
        // try to open a run at x: two equal consecutive increments with all three points in one binade
        const float y = next(x), z = next(y);
        const double d1 = (double)y - (double)x, d2 = (double)z - (double)y;
        const bool same = binade(x) == binade(y) && binade(y) == binade(z) && binade(x) != (1 << 20);
        if (d1 == d2 && (d1 == 0.0 || same)) {
            size_t count;
            if (d1 == 0.0) {
                count = n - i;   // the sequence is stuck: x + step rounds back to x
            } else {
                // how many further results stay inside the binade of x — in integer units of the
                // lattice spacing u, so that no rounding can put a point on the wrong side of a border
                int e = 0;
                (void)frexpf(x, &e);                                   // |x| in [2^(e-1), 2^e)
                const int ue = e - 24 > -149 ? e - 24 : -149;
                const double u = ldexp(1.0, ue);
                const long long A = llround(fabs((double)x) / u), D = llround(fabs(d1) / u);
                const long long H = llround(ldexp(1.0, e) / u), L = llround(ldexp(1.0, e - 1) / u);
                const bool growing = (x > 0) == (d1 > 0);
                long long steps = growing ? (H - 1 - A) / D : (A - L) / D;
                if (steps < 2) steps = 2;                              // y and z were verified above
                count = (size_t)steps + 1;
                if (count > n - i) count = n - i;
            }
            segs.push_back(ArangeSeg{(unsigned long long)i, (double)x, d1});
            x = (float)((double)x + (double)(count - 1) * d1);          // last element of the run (exact)
            i += count;
            if (i < n) x = next(x);                                    // step out of the run sequentially
        } else {
            segs.push_back(ArangeSeg{(unsigned long long)i, (double)x, 0.0});   // a single element
            ++i;
            x = y;
        }
        if (segs.size() > (size_t(1) << 20)) return np::fail(NP_ERR_INVALID, "np_arange: degenerate step");
    }
    np::Scratch table;
    if (int rc = table.alloc(segs.size() * sizeof(ArangeSeg))) return rc;
    if (int rc = np_memcpy_h2d(table.ptr, segs.data(), segs.size() * sizeof(ArangeSeg))) return rc;
    size_t blocks = (n + 255) / 256;
    const size_t cap = (size_t)np::num_cus() * 32;
    if (blocks > cap) blocks = cap;
    const size_t per_block = ((n + blocks - 1) / blocks + 255) / 256 * 256;
    blocks = (n + per_block - 1) / per_block;
    arange_kernel<<<(unsigned)blocks, 256, 0, np::stream()>>>(out, (const ArangeSeg *)table.ptr, (unsigned)segs.size(), n,
                                                             per_block);
    NP_LAUNCH_CHECK("arange_kernel");
    return NP_OK;
}

int np_copy2d(float *dst, size_t dst_pitch, const float *src, size_t src_pitch, size_t width, size_t rows) {
    if (width == 0 || rows == 0) return NP_OK;
    if (!dst || !src) return np::fail(NP_ERR_INVALID, "np_copy2d: null pointer");
    if (dst_pitch < width || src_pitch < width) return np::fail(NP_ERR_INVALID, "np_copy2d: pitch smaller than the row width");
    if (int rc = np::ensure_init()) return rc;
    if (dst_pitch == width && src_pitch == width) return np_memcpy_d2d(dst, src, rows * width * sizeof(float));
    hipStream_t s = np::stream();
    const bool vec = width % 4 == 0;
    const size_t per_row = vec ? width / 4 : width;
    const size_t units = per_row * rows;
    size_t blocks = (units + 255) / 256;
    const size_t cap = (size_t)np::num_cus() * 32;
    if (blocks > cap) blocks = cap;
    if (units >> 32) return np::fail(NP_ERR_INVALID, "np_copy2d: more than 2^32 units");
    if (vec) copy2d_kernel<4><<<(unsigned)blocks, 256, 0, s>>>(dst, dst_pitch, src, src_pitch, (unsigned)per_row, units);
    else copy2d_kernel<1><<<(unsigned)blocks, 256, 0, s>>>(dst, dst_pitch, src, src_pitch, (unsigned)per_row, units);
    NP_LAUNCH_CHECK("copy2d_kernel");
    return NP_OK;
}

int np_strided_copy(const float *in, float *out, int ndim, const int *host_shape, const long long *host_strides) {
    if (ndim < 0 || ndim > MAX_ND) return np::fail(NP_ERR_INVALID, "np_strided_copy: ndim %d not in 0..%d", ndim, MAX_ND);
    if (ndim > 0 && (!host_shape || !host_strides)) return np::fail(NP_ERR_INVALID, "np_strided_copy: null shape/strides");
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) {
        if (host_shape[i] < 0) return np::fail(NP_ERR_INVALID, "np_strided_copy: negative dimension");
        n *= (size_t)host_shape[i];
    }
    if (n == 0) return NP_OK;
    if (!in || !out) return np::fail(NP_ERR_INVALID, "np_strided_copy: null pointer");
    if (int rc = np::ensure_init()) return rc;
    bool contiguous = true;
    size_t expect = 1;
    for (int i = ndim - 1; i >= 0; --i) {
        if (host_shape[i] != 1 && host_strides[i] != (long long)expect) contiguous = false;
        expect *= (size_t)host_shape[i];
    }
    if (contiguous) return np_memcpy_d2d(out, in, n * sizeof(float));
    PermuteArgs a;
    a.ndim = (unsigned)ndim;
    for (int i = 0; i < MAX_ND; ++i) {
        a.out_shape[i] = 1;
        a.in_stride[i] = 0;
    }
    for (int i = 0; i < ndim; ++i) {
        a.out_shape[i] = (unsigned)host_shape[i];
        a.in_stride[i] = (size_t)host_strides[i];   // negative strides wrap modulo 2^64, as pointer arithmetic does
    }
    return launch_gather(in, out, n, a);
}

}  // extern "C"
