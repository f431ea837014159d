// gemm_nt_asm_fx.hip -- launchers (= instantiations) of the hand-scheduled NT kernels with the fused mini-PointNet passes (gemm_nt_asm_kernel.h)
#include "gemm_nt_asm_kernel.h"

// same contract as launch_sgemm_nt16_fx (gemm_nt16_fx.hip) plus K % 32 == 0; bit-identical results.  false = no such kernel.
bool launch_sgemm_nt_asm_fx(const GemmParams& p, int tile, int fx, dim3 grid, hipStream_t s) {
    if (p.epi.act != ACT_EPI_NONE || (p.K & 31) || p.k_per_split != p.K) return false;
    if ((long long)128 * p.lda * 4 >= (1ll << 31) || (long long)128 * p.ldb * 4 >= (1ll << 31)) return false;
#define FXL(BN_, MASK) hipLaunchKernelGGL((sgemm_nt_asm_kernel<128, BN_, false, MASK, ACT_EPI_NONE>), grid, dim3(256), 0, s, p); return true
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
