// Exploration harness (dev tool): does global_load_lds_dwordx4 take global addresses that are only 4-byte aligned,
// and what does it cost?  Every lane DMAs 16 bytes from src + off floats (off = 0..3) into its LDS slot; the workgroup
// then copies the slots out.  Checks the values and times a streaming pass per offset.
//   hipcc --offload-arch=gfx950 -O3 -o dma_unaligned dma_unaligned.hip && ./dma_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// ROWLEN floats per "row" handled by 4 lanes (64 B), rows `pitch` floats apart: the access pattern of the GEMM's A tile
__global__ __launch_bounds__(256) void k(const float *src, float *out, unsigned pitch, unsigned rows, unsigned off) {
    __shared__ __attribute__((aligned(16))) float lds[256 * 4];
    const unsigned tid = threadIdx.x, wave = tid >> 6;
    for (unsigned r0 = blockIdx.x * 64; r0 < rows; r0 += gridDim.x * 64) {
        const unsigned row = r0 + (tid >> 2);
        const float *p = src + (size_t)(row < rows ? row : rows - 1) * pitch + off + (tid & 3) * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)p,
                                         (__attribute__((address_space(3))) void *)(lds + wave * 256), 16, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const v4f v = *(const v4f *)(lds + tid * 4);
        if (row < rows) *(v4f *)(out + (size_t)row * 16 + (tid & 3) * 4) = v;
        __syncthreads();
    }
}

int main() {
    const unsigned rows = 1u << 22;
    for (unsigned pitch : {16u, 17u, 4097u}) {
        const unsigned use_rows = pitch > 64 ? rows / 64 : rows;
        const size_t n = (size_t)use_rows * pitch + 64;
        std::vector<float> h(n);
        for (size_t i = 0; i < n; ++i) h[i] = (float)(i % 1000003);
        float *d, *o;
        CK(hipMalloc(&d, n * 4));
        CK(hipMalloc(&o, (size_t)use_rows * 16 * 4));
        CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
        std::vector<float> got((size_t)use_rows * 16);
        for (unsigned off = 0; off < 4; ++off) {
            CK(hipMemset(o, 0, got.size() * 4));
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            k<<<2048, 256>>>(d, o, pitch, use_rows, off);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int it = 0; it < 10; ++it) k<<<2048, 256>>>(d, o, pitch, use_rows, off);
            CK(hipEventRecord(e1));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(got.data(), o, got.size() * 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (unsigned r = 0; r < use_rows; ++r)
                for (unsigned c = 0; c < 16; ++c)
                    if (got[(size_t)r * 16 + c] != h[(size_t)r * pitch + off + c]) ++bad;
            printf("pitch %5u off %u: %zu wrong of %zu, %.3f ms per pass (%.0f GB/s read)\n", pitch, off, bad, got.size(), ms / 10,
                   (double)use_rows * 64 / (ms / 10) / 1e6);
        }
        CK(hipFree(d)); CK(hipFree(o));
    }
    return 0;
}
