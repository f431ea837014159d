// mfma_valu_kinds.hip -- which VALU instruction kinds steal matrix-pipe time from v_mfma_f32_32x32x2_f32 on gfx950?  Wave-specialised: on every SIMD the waves in
// even slots (HW_ID.WAVE_ID) run only dependent MFMA bursts, the waves in odd slots only the instruction kind under test; reported: each kind alone, both
// together, and the sum / max they would give without / with overlap.  Companion of mfma_valu_overlap.hip (f32 FMA: no overlap at all).
// (dev microbenchmark; hipcc --offload-arch=gfx950 -O3 mfma_valu_kinds.hip -o mvk)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;

enum { OP_FMA, OP_INT, OP_EXP, OP_PKFMA, OP_LDS, OP_MAX, OP_MOV, OP_NONE };
static const char* op_name[] = {"v_fma_f32", "v_add_u32/v_xor", "v_exp_f32", "v_pk_fma_f32", "ds_read_b128", "v_max_f32", "v_cndmask/v_mov"};

// WHO: 0 = even slots MFMA / odd slots OP, 1 = everybody MFMA-role only (odd slots idle), 2 = everybody OP-role only (even slots idle)
template <int OP, int WHO, bool BF16>
__global__ __launch_bounds__(256, 4) void k(float* out, int iters, float a0, float c0, int z) {
    __shared__ float lds[4096];
    unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const bool odd = (hwid & 1u) != 0;
    f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float x[16]; unsigned u[16]; f32x2 pk[8];
    for (int i = 0; i < 16; ++i) { x[i] = a0 + i + threadIdx.x * 1e-6f; u[i] = threadIdx.x + i; }
    for (int i = 0; i < 8; ++i) pk[i] = f32x2{a0 + i, a0 - i};
    lds[threadIdx.x] = a0; lds[threadIdx.x + 256] = c0;
    __syncthreads();
    const float a = a0 + threadIdx.x * 1e-6f, b = c0;
    bf16x8 ha, hb; for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(a0 + i); hb[i] = (__bf16)c0; }
    if (!odd) {
        if (WHO != 2)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int s = 0; s < 32; ++s) {
                    if (BF16) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, acc, 0, 0, 0);
                    else      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                }
            }
    } else if (WHO != 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (OP == OP_FMA) x[i] = __builtin_fmaf(x[i], c0, a0);
                    if (OP == OP_INT) u[i] = (u[i] + (unsigned)z) ^ (unsigned)it;
                    if (OP == OP_EXP) x[i] = __builtin_amdgcn_exp2f(x[i]);
                    if (OP == OP_PKFMA && i < 8) { pk[i] = __builtin_elementwise_fma(pk[i], f32x2{c0, c0}, f32x2{a0, a0}); }
                    if (OP == OP_LDS && i < 4) { const float4 t = *reinterpret_cast<const float4*>(&lds[((threadIdx.x & 63) * 4 + i * 256 + (int)x[15] * 0) & 4092]); x[i] += t.x; x[i + 4] += t.y; }
                    if (OP == OP_MAX) x[i] = fmaxf(x[i], x[(i + 1) & 15] * 0.f + c0);
                    if (OP == OP_MOV) u[i] = (u[(i + 1) & 15] > (unsigned)z) ? u[i] : (unsigned)it;
                }
            }
        }
    }
    float s = acc[0] + acc[7];
    for (int i = 0; i < 16; ++i) s += x[i] + (float)u[i];
    for (int i = 0; i < 8; ++i) s += pk[i][0] + pk[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP, int WHO, bool BF16>
float run() {
    float* out; hipMalloc(&out, 256 * 1024 * 4 * 8);
    const int iters = 400, grid = 256 * 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<OP, WHO, BF16>), dim3(grid), dim3(256), 0, 0, out, 20, 1.f, 0.999f, 3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<OP, WHO, BF16>), dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 0.999f, 3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms * 1000.f;
}
template <int OP, bool BF16>
void row() {
    const float m = run<OP, 1, BF16>(), v = run<OP, 2, BF16>(), both = run<OP, 0, BF16>();
    printf("%-18s vs %s : MFMA waves alone %8.1f us   op waves alone %8.1f us   together %8.1f us   (sum %8.1f, max %8.1f)  overlap %4.0f %%\n",
           op_name[OP], BF16 ? "mfma_f32_32x32x16_bf16" : "mfma_f32_32x32x2_f32  ", m, v, both, m + v, m > v ? m : v, 100.0 * (m + v - both) / (m < v ? m : v));
}
int main() {
    row<OP_FMA, false>(); row<OP_INT, false>(); row<OP_EXP, false>(); row<OP_PKFMA, false>(); row<OP_LDS, false>(); row<OP_MAX, false>(); row<OP_MOV, false>();
    row<OP_FMA, true>(); row<OP_INT, true>(); row<OP_EXP, true>(); row<OP_PKFMA, true>(); row<OP_LDS, true>();
    return 0;
}
