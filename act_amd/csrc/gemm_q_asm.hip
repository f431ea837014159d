// gemm_q_asm.hip -- launchers (= instantiations) of the NN / TN kernels with the hand-scheduled main loop (gemm_q_asm_kernel.h)
#include "gemm_q_asm_kernel.h"

// tile: 0 = 128x128 (NN, TN), 1 = 64x128, 2 = 64x64, 3 = 128x64 (1..3 NN only).  N % BN == 0, every K range % 32 == 0, 16-byte aligned operands,
// 32-bit lane offsets inside a tile (host-checked).  false: no such kernel.
bool launch_sgemm_q_asm(const GemmParams& p, int tile, int a_kmajor, dim3 grid, hipStream_t s) {
#define QA_ACT(BM_, BN_, AK_) \
    switch (p.epi.act) { \
        case ACT_EPI_NONE:          hipLaunchKernelGGL((sgemm_q_asm_kernel<BM_, BN_, AK_, false, false, ACT_EPI_NONE>), grid, dim3(256), 0, s, p); break; \
        case ACT_EPI_MUL_GELU_GRAD: hipLaunchKernelGGL((sgemm_q_asm_kernel<BM_, BN_, AK_, false, false, ACT_EPI_MUL_GELU_GRAD>), grid, dim3(256), 0, s, p); break; \
        case ACT_EPI_MUL_RELU_MASK: hipLaunchKernelGGL((sgemm_q_asm_kernel<BM_, BN_, AK_, false, false, ACT_EPI_MUL_RELU_MASK>), grid, dim3(256), 0, s, p); break; \
        default:                    hipLaunchKernelGGL((sgemm_q_asm_kernel<BM_, BN_, AK_>), grid, dim3(256), 0, s, p); break; \
    }
    if (!a_kmajor) {
        if (tile != 0 || p.M % 128 != 0) return false;
        QA_ACT(128, 128, false)
        return true;
    }
    const int bm = (tile == 1 || tile == 2) ? 64 : 128;
    if (p.M % bm != 0) {
        if (tile == 0)      hipLaunchKernelGGL((sgemm_q_asm_kernel<128, 128, true, true>), grid, dim3(256), 0, s, p);
        else if (tile == 1) hipLaunchKernelGGL((sgemm_q_asm_kernel<64, 128, true, true>), grid, dim3(256), 0, s, p);
        else if (tile == 2) hipLaunchKernelGGL((sgemm_q_asm_kernel<64, 64, true, true>), grid, dim3(256), 0, s, p);
        else                hipLaunchKernelGGL((sgemm_q_asm_kernel<128, 64, true, true>), grid, dim3(256), 0, s, p);
        return true;
    }
    if (tile == 0)      { QA_ACT(128, 128, true) }
    else if (tile == 1) { QA_ACT(64, 128, true) }
    else if (tile == 2) { QA_ACT(64, 64, true) }
    else                { QA_ACT(128, 64, true) }
    return true;
#undef QA_ACT
}

// fused max-pool backward: only the epilogue-side term (FX_SCATTER_EPI) exists on the hand-scheduled loop -- the on-load terms (FX_SCATTER_A, FX_AFFINE_B)
// stay on sgemm_q16_kernel.  Same contract as launch_sgemm_q16_fx plus K % 32 == 0; bit-identical.  false = no such kernel.
bool launch_sgemm_q_asm_fx(const GemmParams& p, int a_kmajor, int fx_mask, dim3 grid, hipStream_t s) {
    if (p.epi.act != ACT_EPI_NONE || !a_kmajor || fx_mask != FX_SCATTER_EPI || (p.K & 31) || p.k_per_split != p.K) return false;
    if ((long long)128 * p.lda * 4 >= (1ll << 31) || (long long)32 * p.ldb * 4 >= (1ll << 31)) return false;
    hipLaunchKernelGGL((sgemm_q_asm_kernel<128, 128, true, false, true, ACT_EPI_NONE>), grid, dim3(256), 0, s, p);
    return true;
}
