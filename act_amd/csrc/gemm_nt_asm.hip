// gemm_nt_asm.hip -- launchers (= instantiations) of the NT kernels with the hand-scheduled main loop (gemm_nt_asm_kernel.h)
#include "gemm_nt_asm_kernel.h"

// tile: 0 = 128x128, 1 = 128x64, 2 = 64x64.  N % BN == 0, every K range % 32 == 0, 16-byte aligned operands (host-checked); an M tail is clamped on load
// and guarded on store.  The activation is a template parameter of the hot instantiations (see epilogue_rows).
void launch_sgemm_nt_asm(const GemmParams& p, int tile, dim3 grid, hipStream_t s) {
    const int bm = tile == 2 ? 64 : 128;
    if (p.M % bm != 0) {
        if (tile == 0)      hipLaunchKernelGGL((sgemm_nt_asm_kernel<128, 128, true>), grid, dim3(256), 0, s, p);
        else if (tile == 1) hipLaunchKernelGGL((sgemm_nt_asm_kernel<128, 64, true>), grid, dim3(256), 0, s, p);
        else                hipLaunchKernelGGL((sgemm_nt_asm_kernel<64, 64, true>), grid, dim3(256), 0, s, p);
        return;
    }
#define NTA_ACT(BM_, BN_) \
    switch (p.epi.act) { \
        case ACT_EPI_NONE: hipLaunchKernelGGL((sgemm_nt_asm_kernel<BM_, BN_, false, 0, ACT_EPI_NONE>), grid, dim3(256), 0, s, p); break; \
        case ACT_EPI_GELU: hipLaunchKernelGGL((sgemm_nt_asm_kernel<BM_, BN_, false, 0, ACT_EPI_GELU>), grid, dim3(256), 0, s, p); break; \
        case ACT_EPI_RELU: hipLaunchKernelGGL((sgemm_nt_asm_kernel<BM_, BN_, false, 0, ACT_EPI_RELU>), grid, dim3(256), 0, s, p); break; \
        default:           hipLaunchKernelGGL((sgemm_nt_asm_kernel<BM_, BN_>), grid, dim3(256), 0, s, p); break; \
    }
    if (tile == 0)      { NTA_ACT(128, 128) }
    else if (tile == 1) { NTA_ACT(128, 64) }
    else                { NTA_ACT(64, 64) }
#undef NTA_ACT
}
