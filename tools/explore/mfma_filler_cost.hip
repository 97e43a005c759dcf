// Microbenchmark: what does ONE memory instruction cost a wave that is issuing v_mfma_f32_16x16x4_f32 back to back?
// One wave per SIMD (256 threads, one workgroup per CU), a loop of 36 independent-ish MFMAs (9 accumulators x 4) per trip plus
// 6 fillers of one kind per trip: none / global_load_lds_dwordx4 / global_load_dwordx4 (VGPR) / ds_write_b128 / ds_read_b128 /
// global_load_dwordx4 + ds_write_b128 of the previous trip's data (register staging).  Reports cycles per trip (s_memtime).
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_filler_cost tools/explore/mfma_filler_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const float *buf, float *sink, unsigned long long *cycles, unsigned trips) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 8 * 256];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *my = lds + wave * 8 * 256;
    const float *src = buf + ((size_t)blockIdx.x * 4 + wave) * 6 * 256 + lane * 4;
    v4f acc[9];
    for (int i = 0; i < 9; ++i) acc[i] = v4f{0, 0, 0, 0};
    v4f stage[6];
    for (int i = 0; i < 6; ++i) stage[i] = v4f{1, 1, 1, 1};
    float a = (float)lane, b = 1.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (unsigned t = 0; t < trips; ++t) {
        v4f got[6];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                const int f = s * 9 + i;   // filler slots 3, 9, 15, 21, 27, 33
                if (f % 6 == 3) {
                    const int c = f / 6;
                    if (MODE == 1)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + c * 256),
                                                         (__attribute__((address_space(3))) void *)(my + c * 256), 16, 0, 0);
                    if (MODE == 2 || MODE == 5) got[c] = *(const v4f *)(src + c * 256);
                    if (MODE == 3 || MODE == 5) *(v4f *)(my + c * 256 + lane * 4) = stage[c];
                    if (MODE == 4) stage[c] = *(const v4f *)(my + c * 256 + lane * 4);
                }
            }
        }
        if (MODE == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE == 2 || MODE == 5)
            for (int c = 0; c < 6; ++c) stage[c] = got[c];
        if (MODE == 4) a += stage[0][0] * 1e-30f;
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float r = 0;
    for (int i = 0; i < 9; ++i) r += acc[i][0] + acc[i][1];
    for (int c = 0; c < 6; ++c) r += stage[c][0];
    if (r == 123.456f) sink[threadIdx.x] = r + my[lane];
    if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int MODE>
void run(const char *name, const float *buf, float *sink, unsigned long long *cyc) {
    const unsigned trips = 2000;
    k<MODE><<<256, 256>>>(buf, sink, cyc, trips);
    k<MODE><<<256, 256>>>(buf, sink, cyc, trips);
    hipDeviceSynchronize();
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-52s %8.1f cycles per trip of 36 MFMAs (%.1f per MFMA; 36 x 32 = 1152)\n", name, (double)c / trips, (double)c / trips / 36);
}

int main() {
    float *buf, *sink;
    unsigned long long *cyc;
    hipMalloc(&buf, 256ull * 4 * 6 * 256 * 4 + 4096);
    hipMalloc(&sink, 4096);
    hipMalloc(&cyc, 8);
    hipMemset(buf, 0, 256ull * 4 * 6 * 256 * 4);
    run<0>("36 MFMAs alone", buf, sink, cyc);
    run<1>("+ 6 global_load_lds_dwordx4", buf, sink, cyc);
    run<2>("+ 6 global_load_dwordx4 -> VGPR", buf, sink, cyc);
    run<3>("+ 6 ds_write_b128", buf, sink, cyc);
    run<4>("+ 6 ds_read_b128", buf, sink, cyc);
    run<5>("+ 6 global_load_dwordx4 + 6 ds_write_b128 (staged)", buf, sink, cyc);
    return 0;
}
