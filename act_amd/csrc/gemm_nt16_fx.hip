// gemm_nt16_fx.hip -- launchers (= instantiations) of the NT b128 kernels with fused producer / consumer passes (mini-PointNet)
#include "gemm_nt16_kernel.h"

bool launch_sgemm_nt16_fx(const GemmParams& p, int tile, int fx, dim3 grid, hipStream_t s) {
    if (p.epi.act != ACT_EPI_NONE) return false;                    // the fused variants are instantiated without activation only (the mini-PointNet applies BatchNorm + ReLU on the NEXT layer's load)
#define FXL(BN_, MASK) hipLaunchKernelGGL((sgemm_nt16_kernel<128, BN_, false, MASK, false, ACT_EPI_NONE>), grid, dim3(256), 0, s, p); return true
    if (tile == 0) {
        if (fx == FX_COLSTATS) { FXL(128, FX_COLSTATS); }
        if (fx == (FX_AFFINE_A | FX_GROUPMAX)) { FXL(128, FX_AFFINE_A | FX_GROUPMAX); }
        if (fx == (FX_AFFINE_A | FX_GROUPMAX | FX_NOSTORE)) { FXL(128, FX_AFFINE_A | FX_GROUPMAX | FX_NOSTORE); }
    } else if (tile == 1) {
        if (fx == FX_COLSTATS) { FXL(64, FX_COLSTATS); }
        if (fx == (FX_AFFINE_A | FX_GROUPMAX)) { FXL(64, FX_AFFINE_A | FX_GROUPMAX); }
        if (fx == (FX_AFFINE_A | FX_GROUPMAX | FX_NOSTORE)) { FXL(64, FX_AFFINE_A | FX_GROUPMAX | FX_NOSTORE); }
    }
#undef FXL
    return false;
}


