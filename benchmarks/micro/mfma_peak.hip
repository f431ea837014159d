// mfma_peak.hip -- how fast can a wave issue v_mfma_f32_16x16x4_f32 with 16 independent accumulators, and does it matter whether
// the accumulators live in ArchVGPRs or AccVGPRs?  (dev microbenchmark; hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void k(float* out, int iters, float a0, float b0) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                else if (MODE == 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int WPS>
void run(const char* name, int blocks_per_cu) {
    float* out; hipMalloc(&out, 256 * 1024 * 4 * 8);
    const int iters = 4000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, WPS>), dim3(grid), dim3(256), 0, 0, out, 100, 1.f, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, WPS>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 /*waves*/ * iters * 64.0 * 2048.0;
    printf("%-34s WGs/CU %d : %7.2f ms  %6.1f TFLOP/s\n", name, blocks_per_cu, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int bpc = 1; bpc <= 4; ++bpc) {
        if (bpc == 1) { run<0, 1>("builtin (compiler-allocated)", 1); run<1, 1>("asm, ArchVGPR accumulators", 1); run<2, 1>("asm, AccVGPR accumulators", 1); }
        if (bpc == 2) { run<0, 2>("builtin (compiler-allocated)", 2); run<1, 2>("asm, ArchVGPR accumulators", 2); run<2, 2>("asm, AccVGPR accumulators", 2); }
        if (bpc == 3) { run<0, 3>("builtin (compiler-allocated)", 3); run<1, 3>("asm, ArchVGPR accumulators", 3); run<2, 3>("asm, AccVGPR accumulators", 3); }
        if (bpc == 4) { run<0, 4>("builtin (compiler-allocated)", 4); run<1, 4>("asm, ArchVGPR accumulators", 4); run<2, 4>("asm, AccVGPR accumulators", 4); }
    }
    return 0;
}
