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
#include "np_internal.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

// in: [batch][rows][cols] -> out: [batch][cols][rows].  TILE x TILE floats per workgroup; the LDS
// tile is padded to TILE+1 floats per row.  Global reads and writes are float4 per lane along
// rows: TILE*4 contiguous bytes per row segment on both sides (256 B at TILE 64, 512 B at 128).
template <int TILE, bool VEC>
__global__ __launch_bounds__(256) void transpose_tile_kernel(const float *__restrict__ in,
                                                             float *__restrict__ out, unsigned rows,
                                                             unsigned cols) {
    constexpr int LDT = TILE + 1;
    constexpr int C4 = TILE / 4;     // float4 columns per tile row
    constexpr int RPP = 256 / C4;    // tile rows covered per pass of the 256 threads
    constexpr int PASSES = TILE / RPP;
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const size_t plane = (size_t)rows * cols;
    const float *src = in + (size_t)blockIdx.z * plane;
    float *dst = out + (size_t)blockIdx.z * plane;
    // diagonal tile order: workgroups that run at the same time (consecutive blockIdx.x) read
    // neighbouring column blocks AND write different column offsets of the output, instead of all
    // writing segments a power-of-two stride apart (HBM channel camping on 8192 x 8192 etc.)
    const unsigned by = (blockIdx.y + blockIdx.x) % gridDim.y;
    const unsigned r0 = by * TILE, c0 = blockIdx.x * TILE;
    const unsigned tx4 = threadIdx.x % C4, ty = threadIdx.x / C4;

    // load: rows of the input tile; all PASSES loads are issued before the first LDS store
    v4f v[PASSES];
#pragma unroll
    for (int j = 0; j < PASSES; ++j) {
        const unsigned r = r0 + ty + RPP * j, c = c0 + 4 * tx4;
        v[j] = v4f{0, 0, 0, 0};
        if constexpr (VEC) {
            if (r < rows && c < cols) v[j] = __builtin_nontemporal_load((const v4f *)(src + (size_t)r * cols + c));
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (r < rows && c + k < cols) v[j][k] = src[(size_t)r * cols + c + k];
        }
    }
#pragma unroll
    for (int j = 0; j < PASSES; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) tile[(4 * tx4 + k) * LDT + ty + RPP * j] = v[j][k];
    __syncthreads();
    // store: rows of the output tile (= columns of the input tile)
#pragma unroll
    for (int j = 0; j < PASSES; ++j) {
        const unsigned oc = ty + RPP * j;   // output row inside the tile (input column)
        const unsigned orow = c0 + oc, ocol = r0 + 4 * tx4;
        v4f w;
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = tile[oc * LDT + 4 * tx4 + k];
        if constexpr (VEC) {
            if (orow < cols && ocol < rows) __builtin_nontemporal_store(w, (v4f *)(dst + (size_t)orow * rows + ocol));
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (orow < cols && ocol + k < rows) dst[(size_t)orow * rows + ocol + k] = w[k];
        }
    }
}

constexpr int MAX_ND = 8;
struct PermuteArgs {
    unsigned ndim;
    unsigned out_shape[MAX_ND];
    size_t in_stride[MAX_ND];   // element stride of the input axis that feeds output axis i
};

template <typename I>
__global__ __launch_bounds__(256) void permute_gather_kernel(const float *__restrict__ in,
                                                             float *__restrict__ out, I n,
                                                             PermuteArgs a) {
    const I stride = (I)gridDim.x * blockDim.x;
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
        out[i] = in[off];
    }
}

inline bool aligned16(const void *p) { return ((uintptr_t)p & 15u) == 0; }

int g_tile = 0;   // 0 = default, else 64 / 128 (np_layout_set_variant)

template <int TILE>
int launch_transpose(const float *in, float *out, size_t batch, size_t rows, size_t cols, bool vec) {
    const dim3 grid((unsigned)((cols + TILE - 1) / TILE), (unsigned)((rows + TILE - 1) / TILE), (unsigned)batch);
    if (grid.y > 65535) return np::fail(NP_ERR_INVALID, "np_transpose2d: too many row tiles");
    const size_t lds = (size_t)TILE * (TILE + 1) * sizeof(float);
    if (vec)
        transpose_tile_kernel<TILE, true><<<grid, 256, lds, np::stream()>>>(in, out, (unsigned)rows, (unsigned)cols);
    else
        transpose_tile_kernel<TILE, false><<<grid, 256, lds, np::stream()>>>(in, out, (unsigned)rows, (unsigned)cols);
    NP_LAUNCH_CHECK("transpose_tile_kernel");
    return NP_OK;
}

}  // namespace

extern "C" {

int np_layout_set_variant(int variant) {
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
    const bool vec = rows % 4 == 0 && cols % 4 == 0 && aligned16(in) && aligned16(out);
    // 128 x 128 tiles (512-byte row segments) for large matrices, 64 x 64 when that would leave
    // CUs without work
    int tile = g_tile;
    if (tile == 0)
        tile = (((rows + 127) / 128) * ((cols + 127) / 128) * batch >= (size_t)np::num_cus() * 4) ? 128 : 64;
    if (tile == 128) {
        static bool attr_set = false;
        if (!attr_set) {   // 66 KB of dynamic LDS is above the 64 KB default limit
            NP_HIP_CHECK(hipFuncSetAttribute((const void *)transpose_tile_kernel<128, true>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 129 * 4));
            NP_HIP_CHECK(hipFuncSetAttribute((const void *)transpose_tile_kernel<128, false>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 129 * 4));
            attr_set = true;
        }
        return launch_transpose<128>(in, out, batch, rows, cols, vec);
    }
    return launch_transpose<64>(in, out, batch, rows, cols, vec);
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
    // general gather
    size_t in_strides[MAX_ND];
    size_t s = 1;
    for (int i = ndim - 1; i >= 0; --i) {
        in_strides[i] = s;
        s *= (size_t)host_shape[i];
    }
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
    size_t blocks = (n + 255) / 256;
    const size_t cap = (size_t)np::num_cus() * 16;
    if (blocks > cap) blocks = cap;
    if (n < (size_t(1) << 31))
        permute_gather_kernel<uint32_t><<<(unsigned)blocks, 256, 0, np::stream()>>>(in, out, (uint32_t)n, a);
    else
        permute_gather_kernel<uint64_t><<<(unsigned)blocks, 256, 0, np::stream()>>>(in, out, (uint64_t)n, a);
    NP_LAUNCH_CHECK("permute_gather_kernel");
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
    size_t blocks = (n + 255) / 256;
    const size_t cap = (size_t)np::num_cus() * 16;
    if (blocks > cap) blocks = cap;
    if (n < (size_t(1) << 31))
        permute_gather_kernel<uint32_t><<<(unsigned)blocks, 256, 0, np::stream()>>>(in, out, (uint32_t)n, a);
    else
        permute_gather_kernel<uint64_t><<<(unsigned)blocks, 256, 0, np::stream()>>>(in, out, (uint64_t)n, a);
    NP_LAUNCH_CHECK("permute_gather_kernel");
    return NP_OK;
}

}  // extern "C"
