// mfma_chain.hip -- issue rate of v_mfma_f32_32x32x2_f32 / 16x16x4 as a function of the number of INDEPENDENT accumulator chains per wave and of
// the waves per SIMD: does a burst of back-to-back dependent MFMAs (attention: 32 products into one S tile) keep the matrix pipe busy?
// (dev microbenchmark; hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int WPS, bool BIG>
__global__ __launch_bounds__(256, WPS) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC]; f32x4 acs[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) acc[i][r] = 0.f; acs[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    float a[8], b[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) { a[s] = a0 + threadIdx.x * 1e-6f + s; b[s] = b0 + s; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                if (BIG) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b[s], acc[i], 0, 0, 0);
                else     acs[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[s], acs[i], 0, 0, 0);
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += BIG ? acc[i][0] + acc[i][5] : acs[i][0] + acs[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int WPS, bool BIG>
void run() {
    float* out; hipMalloc(&out, 256 * 1024 * 4 * 8);
    const int iters = 2000, grid = 256 * WPS;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, WPS, BIG>), dim3(grid), dim3(256), 0, 0, out, 50, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, WPS, BIG>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 * iters * 8.0 * NACC * (BIG ? 4096.0 : 2048.0);
    printf("%s chains/wave %d waves/SIMD %d : %7.3f ms  %6.1f TFLOP/s\n", BIG ? "32x32x2" : "16x16x4", NACC, WPS, ms, flops / ms / 1e9);
    hipFree(out);
}
int main() {
    run<1, 1, true>(); run<1, 2, true>(); run<1, 3, true>(); run<1, 4, true>();
    run<2, 1, true>(); run<2, 2, true>(); run<2, 3, true>(); run<2, 4, true>();
    run<4, 1, true>(); run<4, 2, true>(); run<4, 4, true>();
    run<1, 1, false>(); run<1, 2, false>(); run<1, 4, false>();
    run<2, 1, false>(); run<2, 2, false>(); run<2, 4, false>();
    run<4, 1, false>(); run<4, 2, false>(); run<4, 4, false>();
    return 0;
}
