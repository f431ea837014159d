// gemm_q_asm_kernel.h -- NN / TN fp32 GEMM kernels (B stored [K][N]: input gradients dX = dY . W, weight gradients dW = dY^T . X) around the
// hand-scheduled main loop of gemm_nt_asm_loop.h.  Operand fetch scheme of the quad-fragment kernels (gemm_q16_kernel.h): a row-contiguous
// operand keeps its [k][rows] memory layout in LDS (no transpose anywhere) and one ds_read_b128 along the rows feeds the four 16-wide blocks of a
// wave's 64 rows for one MFMA k-step; a K-contiguous A keeps the swizzled [row][16 k] image of the NT kernels.  The loop is the NT one with
// other ds_read offsets and another fragment-register -> MFMA mapping (gen_nt_asm.py, a_row / b_row): 32-deep K tiles, two fragment sets, one
// barrier per 8 k-steps, every memory instruction in an MFMA gap.  Same products in the same order as sgemm_q16_kernel: BIT-IDENTICAL to tiles
// 13 / 14 / 15 / 16 for the same split-K; shared epilogue_rows (a lane owns four consecutive columns -> float4 stores; ROWQ row order for TN).
#pragma once
#include "gemm_common.h"
#include "gemm_nt_asm_loop.h"

template <int N> struct QAsmVec;
template <> struct QAsmVec<2> { typedef u32x2 type; };
template <> struct QAsmVec<4> { typedef u32x4 type; };

// A_K: A stored [M][K] (NN; BM 64 | 128, M tail allowed) or [K][M] (TN; BM = 128).  BN = 128: 2 x 2 waves; BN = 64: 4 x 1 waves (NN only).
// FXE: C[r][c] += ep_arg[r / group][c] == r % group ? ep_src[r / group][c] : 0 in the epilogue (max-pool backward, see gemm_q16_kernel.h).
template <int BM, int BN, bool A_K, bool MG = false, bool FXE = false, int ACT = -1>
__global__ __launch_bounds__(256, BM * BN > 128 * 64 ? 2 : (BM * BN > 64 * 64 ? 3 : 4)) void sgemm_q_asm_kernel(const GemmParams p) {
    static_assert(A_K || BM == 128, "row-contiguous A needs BM = 128");
    static_assert(BN == 128 || (BN == 64 && A_K), "BN = 128 (2 x 2 waves) or 64 (4 x 1 waves, NN)");
    static_assert(!MG || A_K, "M tail: K-contiguous A only");
    constexpr int WN = (BN == 64) ? 1 : 2, WM = 4 / WN;
    constexpr int TM = BM / WM / 16, TN = 4;
    constexpr int NA = BM / 32, NB = BN / 32;
    constexpr int A_KG = BM * 64, B_KG = BN * 64, B_BASE = 2 * A_KG, STAGE = 2 * A_KG + 2 * B_KG;        // bytes
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
    const int wg = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    int tile_m, tile_n;
    tile_coords(p, wg, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = blockIdx.z * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int ntiles = (kend - kbeg) / 32;
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    // ---- staging.  K-contiguous A: as the NT kernels (row 2 (tid >> 4) + ((tid >> 2) & 1) + 32 i, sub-tile (tid >> 3) & 1, chunk tid & 3, swizzled).
    // Row-contiguous operand: thread owns float4 #(tid + 256 i) of the [32 k][rows] tile, stored at the same index (two [16][rows] sub-tiles back to back).
    unsigned offa[NA], offb[NB], wbase_a, wbase_b, step_a, step_b;
    const float *pa, *pb;
    if constexpr (A_K) {
        const int cc = tid & 3, kg = (tid >> 3) & 1, row = 2 * (tid >> 4) + ((tid >> 2) & 1);
        wbase_a = lbase + kg * A_KG + (unsigned)(row * 64 + 16 * (cc ^ ((4 - ((row >> 2) & 3)) & 3)));
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int r = row + 32 * i;
            offa[i] = (unsigned)(((MG ? min(m0 + r, p.M - 1) - m0 : r) * p.lda + kg * 16 + cc * 4) * 4);
        }
        pa = p.A + (size_t)m0 * p.lda + kbeg; step_a = 128u;
    } else {
        wbase_a = lbase + tid * 16;
#pragma unroll
        for (int i = 0; i < NA; ++i) { const int v = tid + 256 * i; offa[i] = (unsigned)(((v / (BM / 4)) * p.lda + (v % (BM / 4)) * 4) * 4); }
        pa = p.A + (size_t)kbeg * p.lda + m0; step_a = (unsigned)(32 * p.lda * 4);
    }
    wbase_b = lbase + tid * 16;
#pragma unroll
    for (int j = 0; j < NB; ++j) { const int v = tid + 256 * j; offb[j] = (unsigned)(((v / (BN / 4)) * p.ldb + (v % (BN / 4)) * 4) * 4); }
    pb = p.B + (size_t)kbeg * p.ldb + n0; step_b = (unsigned)(32 * p.ldb * 4);

    const int kl = lane >> 4, ml = lane & 15;
    f32x4 acc[TM][TN];
    if (ntiles > 0) {
#pragma unroll
        for (int i = 0; i < NA; ++i)                                 // K-tile 0 -> stage 0
            *reinterpret_cast<float4*>(lds + (wbase_a - lbase) + i * (A_K ? 2048 : 4096)) = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(pa) + offa[i]);
#pragma unroll
        for (int j = 0; j < NB; ++j)
            *reinterpret_cast<float4*>(lds + B_BASE + (wbase_b - lbase) + j * 4096) = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(pb) + offb[j]);
        __syncthreads();
        const int hsw = (4 - ((ml >> 2) & 3)) & 3;
        const unsigned ra = lbase + (A_K ? (unsigned)((wm * (BM / WM) + ml) * 64 + 16 * (kl ^ hsw)) : (unsigned)(((4 * kl) * BM + wm * 64 + 4 * ml) * 4));
        const unsigned rb = lbase + (unsigned)(((4 * kl) * BN + wn * 64 + 4 * ml) * 4);
        typename QAsmVec<NA>::type va; typename QAsmVec<NB>::type vb;
#pragma unroll
        for (int i = 0; i < NA; ++i) va[i] = offa[i];
#pragma unroll
        for (int j = 0; j < NB; ++j) vb[j] = offb[j];
        if constexpr (!A_K) tn_asm_loop_4x4(acc, pa, pb, ntiles, step_a, step_b, va, vb, wbase_a, wbase_b, ra, rb);
        else if constexpr (BM == 128 && BN == 128) nn_asm_loop_4x4(acc, pa, pb, ntiles, step_a, step_b, va, vb, wbase_a, wbase_b, ra, rb);
        else if constexpr (BM == 64 && BN == 128) nn_asm_loop_2x4(acc, pa, pb, ntiles, step_a, step_b, va, vb, wbase_a, wbase_b, ra, rb);
        else if constexpr (BM == 128 && BN == 64) nn_asm_loop_2x4_w4x1(acc, pa, pb, ntiles, step_a, step_b, va, vb, wbase_a, wbase_b, ra, rb);
        else nn_asm_loop_1x4_w4x1(acc, pa, pb, ntiles, step_a, step_b, va, vb, wbase_a, wbase_b, ra, rb);
    } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int wu = __builtin_amdgcn_readfirstlane(wave);
    const int wmu = WN == 2 ? wu >> 1 : wu, wnu = WN == 2 ? wu & 1 : 0;
    epilogue_rows<ACT, TM, TN, MG, false, true, !A_K, FXE>(p, acc, m0 + (A_K ? wmu * (BM / WM) : wmu * 64), n0 + wnu * 64, ml, kl);
}
