// mfma_valu_overlap.hip -- do VALU phases of one wave overlap the MFMA bursts of the OTHER waves of its SIMD?  (attention forward: time = skeleton + MFMA,
// benchmarks/attn_fwd_diag.py.)  Every wave alternates a burst of NM dependent v_mfma_f32_32x32x2_f32 and a burst of NV v_fma_f32 (16 independent chains);
// variants: MFMA only, VALU only, both in program order, both with the phase order flipped for every second workgroup, and wave-specialised
// (even workgroups only MFMA, odd only VALU: pure co-issue test).  (dev microbenchmark; hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o mvo)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NM, int NV, int MODE, int WPS>
__global__ __launch_bounds__(256, WPS) void k(float* out, int iters, float a0, float c0) {
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = a0 + i + threadIdx.x * 1e-6f;
    const float a = a0 + threadIdx.x * 1e-6f, b = c0;
    // parity of the wave's SLOT on its SIMD (HW_ID.WAVE_ID), not of blockIdx: consecutive workgroups go to different XCDs, so a blockIdx parity splits the
    // chip into MFMA-only and VALU-only XCDs instead of mixing the two kinds on every SIMD (first version of this file)
    unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const bool odd = (hwid & 1u) != 0;
    const bool flip = (MODE == 3) && odd;
    const bool only_m = MODE == 0 || (MODE == 4 && !odd), only_v = MODE == 1 || (MODE == 4 && odd);
    if (MODE == 5 && blockIdx.x == 0 && (threadIdx.x & 63) == 0) out[256 * 1024 * 8 - 1 - (threadIdx.x >> 6)] = (float)(hwid & 0xffffff);
    auto mf = [&]() {
#pragma unroll
        for (int s = 0; s < NM; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    };
    auto va = [&]() {
#pragma unroll
        for (int s = 0; s < NV / 16; ++s)
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], c0, a0);
    };
    if (flip) { va(); __builtin_amdgcn_sched_barrier(0); }
    for (int it = 0; it < iters; ++it) {
        if (!only_v) mf();
        __builtin_amdgcn_sched_barrier(0);
        if (!only_m) va();
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = acc[0] + acc[7];
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NM, int NV, int MODE, int WPS>
float run() {
    float* out; hipMalloc(&out, 256 * 1024 * 4 * 8);
    const int iters = 400, grid = 256 * WPS;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NM, NV, MODE, WPS>), dim3(grid), dim3(256), 0, 0, out, 20, 1.f, 0.999f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NM, NV, MODE, WPS>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 0.999f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms * 1000.f;
}
template <int NM, int NV, int WPS>
void sweep() {
    const float m = run<NM, NV, 0, WPS>(), v = run<NM, NV, 1, WPS>(), both = run<NM, NV, 2, WPS>(), fl = run<NM, NV, 3, WPS>(), sp = run<NM, NV, 4, WPS>();
    printf("NM %3d NV %4d waves/SIMD %d : MFMA only %8.1f us  VALU only %8.1f us  program order %8.1f us  flipped per WG %8.1f us  (sum %8.1f, max %8.1f)  specialised WGs %8.1f us\n",
           NM, NV, WPS, m, v, both, fl, m + v, m > v ? m : v, sp);
}
int main() {
    sweep<32, 128, 1>(); sweep<32, 128, 2>(); sweep<32, 128, 3>(); sweep<32, 128, 4>();
    sweep<32, 256, 2>(); sweep<32, 256, 3>(); sweep<32, 256, 4>();
    sweep<32, 512, 2>(); sweep<32, 512, 3>(); sweep<32, 512, 4>();
    sweep<8, 64, 3>(); sweep<8, 64, 4>();
    return 0;
}
