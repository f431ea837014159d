// gemm_q16_fx.hip -- launchers (= instantiations) of the quad-fragment kernels with fused passes (mini-PointNet backward)
#include "gemm_q16_kernel.h"

bool launch_sgemm_q16_fx(const GemmParams& p, int a_kmajor, int fx_mask, dim3 grid, hipStream_t s) {
    if (p.epi.act != ACT_EPI_NONE) return false;                    // fused variants: no activation (see launch_sgemm_nt16_fx)
#define QL(AK, FB, FA, FE) hipLaunchKernelGGL((sgemm_q16_kernel<128, 128, AK, false, false, FB, FA, FE, ACT_EPI_NONE>), grid, dim3(256), 0, s, p); return true
    if (!a_kmajor) {                                                  // TN weight gradients
        if (fx_mask == FX_AFFINE_B) { QL(false, true, false, false); }
        if (fx_mask == (FX_AFFINE_B | FX_SCATTER_A)) { QL(false, true, true, false); }
        if (fx_mask == FX_SCATTER_A) { QL(false, false, true, false); }
    } else {                                                          // NN input gradients
        if (fx_mask == FX_SCATTER_A) { QL(true, false, true, false); }
        if (fx_mask == FX_SCATTER_EPI) { QL(true, false, false, true); }
        if (fx_mask == (FX_SCATTER_A | FX_SCATTER_EPI)) { QL(true, false, true, true); }
    }
#undef QL
    return false;
}

