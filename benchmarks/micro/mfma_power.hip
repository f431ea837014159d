// mfma_power.hip -- the matrix pipe's sustained rate as a function of the operand VALUES (DVFS: the chip clocks to its power budget).
// 16 independent 16x16x4 accumulators per wave, no memory operations in the loop; A / B operands per k-step come from registers filled with
//   mode 0: zeros   mode 1: one constant   mode 2: uniform random fp32 in [-1, 1) (what GEMM operands look like)   mode 3: random, 8 register sets
// rotated per k-step (more toggling on the operand buses).  hipcc --offload-arch=gfx950 -O3 mfma_power.hip -o mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ src, float* out, int iters) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a[8][4], b[8][4];
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[s][i] = src[(s * 8 + i) * 256 + threadIdx.x]; b[s][i] = src[(s * 8 + 4 + i) * 256 + threadIdx.x]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][i], b[s][j], acc[i * 4 + j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    const int n = 64 * 256;
    float* h = (float*)malloc(n * 4); float *src, *out;
    hipMalloc(&src, n * 4); hipMalloc(&out, 256 * 1024 * 4);
    const char* names[4] = {"zeros", "one constant (0.37)", "uniform random [-1,1), same set every k-step", "uniform random, 8 operand sets"};
    for (int mode = 0; mode < 4; ++mode) {
        srand(1);
        for (int i = 0; i < n; ++i) {
            const float r = 2.f * (rand() / (float)RAND_MAX) - 1.f;
            h[i] = mode == 0 ? 0.f : mode == 1 ? 0.37f : (mode == 2 ? h[i % (8 * 256)] * 0.f + (i < 8 * 256 ? r : h[i % (8 * 256)]) : r);
        }
        hipMemcpy(src, h, n * 4, hipMemcpyHostToDevice);
        const int iters = 6000, grid = 512;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, src, out, 200);
        hipDeviceSynchronize();
        float best = 1e9f, last = 0.f;
        for (int rep = 0; rep < 6; ++rep) {                 // ~25 ms per launch: long enough for the clock governor to settle
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, src, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms; last = ms;
        }
        const double flops = (double)grid * 4 * iters * 128.0 * 2048.0;
        printf("%-52s best %7.2f ms %6.1f TFLOP/s   last (warm) %7.2f ms %6.1f TFLOP/s\n", names[mode], best, flops / best / 1e9, last, flops / last / 1e9);
    }
    return 0;
}
