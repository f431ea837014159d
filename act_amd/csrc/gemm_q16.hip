// gemm_q16.hip -- launchers (= instantiations) of the plain quad-fragment NN / TN kernels
#include "gemm_q16_kernel.h"

bool launch_sgemm_q16(const GemmParams& p, int tile, int a_kmajor, int b_kmajor, dim3 grid, hipStream_t s) {
    // kernels instantiated per activation (see epilogue_apply): none (dW, plain dX), gelu' (dX through fc2 -> GELU), relu mask (MLP heads)
#define Q16_ACT(BM_, BN_, AK_) \
    switch (p.epi.act) { \
        case ACT_EPI_NONE:          hipLaunchKernelGGL((sgemm_q16_kernel<BM_, BN_, AK_, false, false, false, false, false, ACT_EPI_NONE>), grid, dim3(256), 0, s, p); break; \
        case ACT_EPI_MUL_GELU_GRAD: hipLaunchKernelGGL((sgemm_q16_kernel<BM_, BN_, AK_, false, false, false, false, false, ACT_EPI_MUL_GELU_GRAD>), grid, dim3(256), 0, s, p); break; \
        case ACT_EPI_MUL_RELU_MASK: hipLaunchKernelGGL((sgemm_q16_kernel<BM_, BN_, AK_, false, false, false, false, false, ACT_EPI_MUL_RELU_MASK>), grid, dim3(256), 0, s, p); break; \
        default:                    hipLaunchKernelGGL((sgemm_q16_kernel<BM_, BN_, AK_, false>), grid, dim3(256), 0, s, p); break; \
    }
    if (a_kmajor && b_kmajor) return false;
    if (b_kmajor) return false;                                       // (A [K][M], B [N][K]) never occurs on this path
    if (a_kmajor && (tile == 2 || tile == 3)) {
        const int bm = tile == 2 ? 64 : 128;
        if (p.M % bm != 0) {
            if (tile == 2) hipLaunchKernelGGL((sgemm_q16_kernel<64, 64, true, false, true>), grid, dim3(256), 0, s, p);
            else           hipLaunchKernelGGL((sgemm_q16_kernel<128, 64, true, false, true>), grid, dim3(256), 0, s, p);
        } else {
            if (tile == 2) { Q16_ACT(64, 64, true) }
            else           { Q16_ACT(128, 64, true) }
        }
        return true;
    }
    if (a_kmajor) {                                                   // NN: dX = dY . W
        const int bm = tile == 1 ? 64 : 128;
        if (p.M % bm != 0) {
            if (tile == 1) hipLaunchKernelGGL((sgemm_q16_kernel<64, 128, true, false, true>), grid, dim3(256), 0, s, p);
            else           hipLaunchKernelGGL((sgemm_q16_kernel<128, 128, true, false, true>), grid, dim3(256), 0, s, p);
        } else {
            if (tile == 1) { Q16_ACT(64, 128, true) }
            else           { Q16_ACT(128, 128, true) }
        }
        return true;
    }
    if (tile != 0) return false;                                      // TN: dW = dY^T . X, 128x128 only
    Q16_ACT(128, 128, false)
    return true;
#undef Q16_ACT
}

