// Microbenchmark: how fast can ONE CU pull L2-resident data, (a) with global_load_lds_dwordx4 (LDS-DMA), (b) with
// global_load_dwordx4 into VGPRs, (c) as (b) + ds_write_b128 — all 256 CUs at once, every workgroup walking its own
// 24 KiB window of a 6 MiB buffer (L2-resident after the first pass) with 4 waves, `inflight` KiB per wave outstanding.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_fill_bw tools/explore/l2_fill_bw.hip ; run: /tmp/l2_fill_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE, int DEPTH>
__global__ __launch_bounds__(256, 1) void fill_kernel(const float *buf, float *sink, unsigned window_floats, unsigned iters) {
    __shared__ __attribute__((aligned(16))) float lds[4 * DEPTH * 256];
    const unsigned lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *base = buf + (size_t)blockIdx.x * window_floats;
    const unsigned chunks = window_floats / 256;   // 1 KiB pieces in the window
    v4f acc{0, 0, 0, 0};
    for (unsigned it = 0; it < iters; ++it) {
        for (unsigned c = wave * DEPTH; c + DEPTH <= chunks; c += 4 * DEPTH) {
            if (MODE == 0) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + (c + d) * 256 + lane * 4),
                                                     (__attribute__((address_space(3))) void *)(lds + (wave * DEPTH + d) * 256), 16, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                v4f r[DEPTH];
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) r[d] = *(const v4f *)(base + (c + d) * 256 + lane * 4);
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    if (MODE == 2) *(v4f *)(lds + (wave * DEPTH + d) * 256 + lane * 4) = r[d];
                    else acc += r[d];
                }
            }
        }
    }
    if (MODE == 0 || MODE == 2) acc += *(v4f *)(lds + threadIdx.x * 4);
    if (acc[0] == 123.456f) sink[threadIdx.x] = acc[1] + acc[2] + acc[3];
}

template <int MODE, int DEPTH>
void run(const char *name, const float *buf, float *sink, unsigned window_kib, unsigned iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const unsigned wf = window_kib * 256;
    for (int rep = 0; rep < 3; ++rep) fill_kernel<MODE, DEPTH><<<256, 256>>>(buf, sink, wf, iters);
    hipEventRecord(a);
    for (int rep = 0; rep < 5; ++rep) fill_kernel<MODE, DEPTH><<<256, 256>>>(buf, sink, wf, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    const double bytes = 256.0 * window_kib * 1024 * iters;
    printf("%-44s window %3u KiB depth %d: %7.1f us  %6.2f TB/s  %5.1f GB/s per CU\n", name, window_kib, DEPTH, ms * 1e3, bytes / ms / 1e9, bytes / 256 / ms / 1e6);
}

int main() {
    float *buf, *sink;
    const size_t n = 256ull * 96 * 256;   // 256 windows of up to 96 KiB
    hipMalloc(&buf, n * 4); hipMalloc(&sink, 4096);
    hipMemset(buf, 0, n * 4);
    for (unsigned w : {24u, 96u}) {
        run<0, 2>("global_load_lds_dwordx4", buf, sink, w, 200);
        run<0, 6>("global_load_lds_dwordx4", buf, sink, w, 200);
        run<1, 2>("global_load_dwordx4 -> VGPR", buf, sink, w, 200);
        run<1, 6>("global_load_dwordx4 -> VGPR", buf, sink, w, 200);
        run<2, 6>("global_load_dwordx4 -> VGPR -> ds_write_b128", buf, sink, w, 200);
    }
    return 0;
}
