// gemm_common.h -- parameter block and fused epilogues shared by the fp32 MFMA GEMM kernels (gemm.hip, gemm16.hip, gemm_nt16*.hip, gemm_q16*.hip)
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GEMM_BK_DEFAULT 16
#ifndef GEMM_MIN_WAVES
#define GEMM_MIN_WAVES 4          // waves per SIMD the register allocator must leave room for (4 workgroups / CU)
#endif

// optional fusions of the producer / consumer passes around a GEMM (mini-PointNet: BatchNorm statistics, BatchNorm apply + ReLU, max-pool)
struct GemmFx {
    const float* a_scale; const float* a_shift;    // NT: A'[r,k] = max(0, A[r,k] * a_scale[k] + a_shift[k]) applied while staging A (K <= 1024)
    const float* b_scale; const float* b_shift;    // TN: B'[k,n] = max(0, B[k,n] * b_scale[n] + b_shift[n]) applied while staging B
    float* tile_stats;                             // NT: [tiles_m][2][N] per-tile column mean and sum of squared deviations of the stored values
    float* gmax; int32_t* garg; int group;         // NT: max (+ first arg-max) over every `group` (32 | 64) consecutive rows -> [M/group][N]
    int store_c;                                   // 0: C is not written (only its group max is wanted)
    const float* sa_src; const int32_t* sa_arg;    // NN / TN: virtual A[r][c] = sa_arg[r/group][c] == r % group ? sa_src[r/group][c] : 0 (max-pool backward on load)
    const float* ep_src; const int32_t* ep_arg;    // NN: C[r][c] += ep_arg[r/group][c] == r % group ? ep_src[r/group][c] : 0 in the epilogue
    const int32_t* row_groups;                     // NT + group max, no C store: A row m = row_groups[m/group]*group + m%group, pooled row m/group -> row_groups[m/group]
};
enum { FX_AFFINE_A = 1, FX_COLSTATS = 2, FX_GROUPMAX = 4, FX_NOSTORE = 8, FX_AFFINE_B = 16, FX_SCATTER_A = 32, FX_SCATTER_EPI = 64 };

struct GemmParams {
    GemmFx fx;
    const float* A; const float* B; float* C;
    int M, N, K, lda, ldb, ldc;
    int k_per_split;                 // K range per blockIdx.z (== K when no split)
    float* partial;                  // split-K workspace [splits][M][N] (C untouched) or null
    act_gemm_epilogue_t epi;
    int tiles_m, tiles_n;
    int group_m;                     // tile rasterisation: GROUP_M tile rows are swept column by column (1 = plain row-major)
    int epi_vec;                     // quad epilogues: C / bias / residual / aux are 16-byte aligned with leading dimensions % 4 == 0 (float4 accesses)
    int xcd_rows;                    // 0: every XCD owns a contiguous band of the rasterised tile order; r (1, 2, 4): the 8 XCDs form an r x 8/r grid of tile blocks
    // round 6: the divisions of the tile rasterisation as multiply-high by host-computed reciprocals (gemm_set_tiling).  All operands are wave-uniform, so this is
    // scalar-ALU work; the compiler expands a run-time integer division into ~20 VECTOR instructions (v_cvt / v_rcp_iflag_f32 / v_mul_hi ...) even for uniform
    // operands, and on gfx950 vector-ALU time is matrix-pipe time (profiles/r06_mfma_valu_kinds.txt).  div_ok = 0: plain divisions (>= 2^24 tiles, or not set).
    unsigned mg_tn, mg_pg, mg_gm, mg_tail;     // ceil(2^sh / d) for d = tiles_n, group_m * tiles_n, group_m, tiles_m % group_m
    int sh_tn, sh_pg, sh_gm, sh_tail;
    int div_ok;
};
// n / d for n < 2^24 with m = ceil(2^sh / d), sh = 24 + ceil(log2 d): n m / 2^sh = n / d + n e / (d 2^sh), e = m d - 2^sh < d <= 2^(sh - 24), so the excess is < 1 / d: exact
__device__ __forceinline__ int gemm_fastdiv(int n, unsigned m, int sh) { return (int)(((unsigned long long)(unsigned)n * m) >> sh); }
inline void gemm_set_tiling(GemmParams& p) {
    p.div_ok = 0;
    const long long nwg = (long long)p.tiles_m * p.tiles_n;
    if (nwg <= 0 || nwg >= (1ll << 24) || p.tiles_n <= 0) return;
    auto mk = [](unsigned d, unsigned& m, int& sh) { int s = 0; while ((1u << s) < d) ++s; sh = 24 + s; m = (unsigned)(((1ull << sh) + d - 1) / d); };
    mk((unsigned)p.tiles_n, p.mg_tn, p.sh_tn);
    if (p.group_m > 1) {
        if ((long long)p.group_m * p.tiles_n >= (1ll << 24)) return;
        mk((unsigned)(p.group_m * p.tiles_n), p.mg_pg, p.sh_pg);
        mk((unsigned)p.group_m, p.mg_gm, p.sh_gm);
        const int tail = p.tiles_m % p.group_m;
        mk((unsigned)(tail > 0 ? tail : 1), p.mg_tail, p.sh_tail);
    }
    p.div_ok = 1;
}

// linear workgroup index (after the XCD remap) -> tile coordinates.  Grouped order: the tiles a (band of) CUs works on at the same
// time form a compact 2-D block, so both the A row-panels and the B column-panels they touch are re-used out of the XCD's L2
// instead of one of them streaming from fabric for every tile row.
__device__ __forceinline__ void tile_coords(const GemmParams& p, int wg, int& tile_m, int& tile_n) {
    if (p.div_ok) {                                                   // same mapping as below, divisions by reciprocal multiplication (scalar ALU)
        if (p.group_m <= 1) { tile_m = gemm_fastdiv(wg, p.mg_tn, p.sh_tn); tile_n = wg - tile_m * p.tiles_n; return; }
        const int per_group = p.group_m * p.tiles_n;
        const int group = gemm_fastdiv(wg, p.mg_pg, p.sh_pg), first_m = group * p.group_m;
        const int gsz = min(p.tiles_m - first_m, p.group_m);          // group_m, or tiles_m % group_m in the last group
        const int in_group = wg - group * per_group;
        tile_n = gsz == p.group_m ? gemm_fastdiv(in_group, p.mg_gm, p.sh_gm) : gemm_fastdiv(in_group, p.mg_tail, p.sh_tail);
        tile_m = first_m + in_group - tile_n * gsz;
        return;
    }
    if (p.group_m <= 1) { tile_m = wg / p.tiles_n; tile_n = wg % p.tiles_n; return; }
    const int per_group = p.group_m * p.tiles_n;
    const int group = wg / per_group, first_m = group * p.group_m;
    const int gsz = min(p.tiles_m - first_m, p.group_m);
    const int in_group = wg - group * per_group;
    tile_m = first_m + in_group % gsz;
    tile_n = in_group / gsz;
}

// GELU (exact, erf form: nn.GELU() of the reference, utils/transformer_layers.py:130-139) in 14 VALU instructions instead of the ~28 of 0.5 x (1 + erff(x / sqrt 2)) on the
// device library's two-branch erff.  On gfx950 no VALU instruction issues beside an f32 MFMA of the same SIMD (profiles/r06_mfma_valu_kinds.txt), so an epilogue's VALU count is
// matrix-pipe time: the erff form cost 5 % of the teacher's fc1 launches and 12-13 % of the student's (profiles/r06_gelu_epilogue_ab.txt).
//   e2(x) = 0.5 erfc(|x| / sqrt 2) = exp2(-(t h(t) + 1)),  t = |x| / sqrt 2,  h(t) = -log2(erfc(t)) / t  as ONE degree-8 polynomial on [0, 4] (weighted minimax fit,
//   benchmarks/fit_gelu_poly.py; beyond t = 4 the polynomial argument is clamped and the exponent keeps falling linearly: e2 < 8e-9 there);
//   gelu(x) = max(x, 0) - |x| e2(x);   gelu'(x) = Phi(x) + x phi(x),  Phi = x >= 0 ? 1 - e2 : e2  (no 1 + erf cancellation on the negative side).
// Against the float64 function over [-12, 12]: max abs error 2.4e-7 (gelu) / 1.5e-7 (gelu'), rms 4e-8 / 2e-8 -- the fp32 erf form itself (torch CPU) has 1.2e-6 / 2.9e-7, rms 1.1e-7 / 4.5e-8.
#ifndef ACT_GELU_FAST
#define ACT_GELU_FAST 1                  // 0: the erff form of rounds 1-5 (A/B builds: ACT_HIPCC_EXTRA=-DACT_GELU_FAST=0)
#endif
__device__ __forceinline__ float gelu_half_erfc(float x) {           // 0.5 erfc(|x| / sqrt 2)
    const float t = __builtin_fabsf(x) * 0.70710678118654752440f;
    const float tc = __builtin_fminf(t, 4.0f);
    float h = -5.621429409075063e-06f;
    h = __builtin_fmaf(h, tc, 9.077531285583973e-05f);
    h = __builtin_fmaf(h, tc, -0.0005887305014766753f);
    h = __builtin_fmaf(h, tc, 0.0017150973435491323f);
    h = __builtin_fmaf(h, tc, 0.0005855816416442394f);
    h = __builtin_fmaf(h, tc, -0.028170492500066757f);
    h = __builtin_fmaf(h, tc, 0.14846348762512207f);
    h = __builtin_fmaf(h, tc, 0.918418288230896f);
    h = __builtin_fmaf(h, tc, 1.62790846824646f);
    return __builtin_amdgcn_exp2f(__builtin_fmaf(-h, t, -1.0f));
}
#if ACT_GELU_FAST
__device__ __forceinline__ float gelu_f(float x) { return __builtin_fmaf(-__builtin_fabsf(x), gelu_half_erfc(x), __builtin_fmaxf(x, 0.0f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float e2 = gelu_half_erfc(x);
    const float Phi = x >= 0.0f ? 1.0f - e2 : e2;
    return __builtin_fmaf(x * 0.39894228040143267794f, __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f), Phi);
}
#else
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}
#endif
// ACT: compile-time value of e.act (ACT_EPI_*), or -1 = decide at run time.  The run-time form inlines erff / expf for EVERY accumulator element
// (64 per lane in a 128 x 128 tile): ~60 KB of code per kernel that the launches without an activation only jump over -- more than the
// instruction cache two CUs share -- so the vector epilogues below take ONE uniform branch to a body specialised for the activation, and the launchers of
// the hot kernels instantiate per activation (launch_sgemm_nt16, launch_sgemm_q16).  This per-element form is left for the ragged-edge kernels.
template <int ACT = -1>
__device__ __forceinline__ float epilogue_apply(const act_gemm_epilogue_t& e, float v, int row, int col) {
    v *= e.alpha;
    if (e.bias) v += e.bias[col];
    const int act = ACT >= 0 ? ACT : e.act;
    switch (act) {
        case ACT_EPI_GELU:      if (e.aux) e.aux[(size_t)row * e.ldaux + col] = v; v = gelu_f(v); break;
        case ACT_EPI_RELU:      v = fmaxf(v, 0.f); break;
        case ACT_EPI_MUL_GELU_GRAD: v *= gelu_grad_f(e.aux[(size_t)row * e.ldaux + col]); break;
        case ACT_EPI_MUL_RELU_MASK: v = e.aux[(size_t)row * e.ldaux + col] > 0.f ? v : 0.f; break;
        default: break;
    }
    if (e.rowscale) v *= e.rowscale[row / e.rows_per_scale];
    if (e.res) v += e.res[(size_t)(e.res_row_div > 1 ? row / e.res_row_div : row) * e.ldr + col];
    return v;
}


// four consecutive columns of one row at once (col % 4 == 0; bias / aux / res 16-byte aligned with leading dimensions % 4 == 0): the elementwise
// kernels that apply the epilogue after a split-K reduction.  Same arithmetic per element as epilogue_apply, in the same order.
template <int ACT>
__device__ __forceinline__ float4 epilogue_apply4_act(const act_gemm_epilogue_t& e, float4 v, int row, int col) {
    v.x *= e.alpha; v.y *= e.alpha; v.z *= e.alpha; v.w *= e.alpha;
    if (e.bias) { const float4 b = *reinterpret_cast<const float4*>(e.bias + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
    if constexpr (ACT == ACT_EPI_GELU) {
        if (e.aux) *reinterpret_cast<float4*>(e.aux + (size_t)row * e.ldaux + col) = v;
        v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w);
    } else if constexpr (ACT == ACT_EPI_RELU) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    } else if constexpr (ACT == ACT_EPI_MUL_GELU_GRAD) {
        const float4 x = *reinterpret_cast<const float4*>(e.aux + (size_t)row * e.ldaux + col);
        v.x *= gelu_grad_f(x.x); v.y *= gelu_grad_f(x.y); v.z *= gelu_grad_f(x.z); v.w *= gelu_grad_f(x.w);
    } else if constexpr (ACT == ACT_EPI_MUL_RELU_MASK) {
        const float4 x = *reinterpret_cast<const float4*>(e.aux + (size_t)row * e.ldaux + col);
        v.x = x.x > 0.f ? v.x : 0.f; v.y = x.y > 0.f ? v.y : 0.f; v.z = x.z > 0.f ? v.z : 0.f; v.w = x.w > 0.f ? v.w : 0.f;
    }
    if (e.rowscale) { const float rs = e.rowscale[row / e.rows_per_scale]; v.x *= rs; v.y *= rs; v.z *= rs; v.w *= rs; }
    if (e.res) {
        const float4 q = *reinterpret_cast<const float4*>(e.res + (size_t)(e.res_row_div > 1 ? row / e.res_row_div : row) * e.ldr + col);
        v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    }
    return v;
}
__device__ __forceinline__ float4 epilogue_apply4(const act_gemm_epilogue_t& e, float4 v, int row, int col) {
    switch (e.act) {
        case ACT_EPI_GELU:          return epilogue_apply4_act<ACT_EPI_GELU>(e, v, row, col);
        case ACT_EPI_RELU:          return epilogue_apply4_act<ACT_EPI_RELU>(e, v, row, col);
        case ACT_EPI_MUL_GELU_GRAD: return epilogue_apply4_act<ACT_EPI_MUL_GELU_GRAD>(e, v, row, col);
        case ACT_EPI_MUL_RELU_MASK: return epilogue_apply4_act<ACT_EPI_MUL_RELU_MASK>(e, v, row, col);
        default:                    return epilogue_apply4_act<ACT_EPI_NONE>(e, v, row, col);
    }
}

// ---- vector epilogue of the NT b128 kernels.  Their B operand is staged with PERMUTED rows (LDS row j*16 + m of a wave's 16*TN-row block holds
// global column TN*m + j), so after the MFMAs lane (ml = lane & 15, kl = lane >> 4) owns, for every row i*16 + 4*kl + r it holds, the TN CONSECUTIVE
// columns TN*ml ... TN*ml + TN-1 (acc[i][0..TN-1][r]): one float4 (float2) per row, and the 16 lanes of a quarter-wave cover 256 (128) contiguous
// bytes of that row.  The per-element epilogue this replaces issued, per lane and 128 x 128 tile, 64 scalar residual loads, 64 scalar stores and
// ~20 VALU instructions of 64-bit address arithmetic per element: launches with bias + residual ran 8 % below the bare product at K = 768 and
// 40 % below at K = 256 (benchmarks/epi_spec_bench.py).  Here every access is a vector, every address is (wave-uniform 64-bit base in SGPRs) +
// (one 32-bit lane offset computed once), the bias is one vector per lane and tile.  Same arithmetic per element in the same order: bit-identical.
// `vec` (wave-uniform): all pointers 16-byte aligned and all leading dimensions % 4 == 0; otherwise scalar accesses.
inline int epilogue_is_vec(const float* C, int ldc, const act_gemm_epilogue_t& e) {
    uintptr_t a = reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(e.bias) | reinterpret_cast<uintptr_t>(e.res) | reinterpret_cast<uintptr_t>(e.aux);
    int ld = ldc | (e.res ? e.ldr : 0) | (e.aux ? e.ldaux : 0);
    return (a & 15) == 0 && (ld & 3) == 0;
}
template <int W> struct VecOf;
template <> struct VecOf<4> { typedef float4 type; };
template <> struct VecOf<2> { typedef float2 type; };
template <int W>
__device__ __forceinline__ void ldw(float (&v)[W], const float* base, unsigned lane_bytes, bool vec) {
    const float* q = reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + lane_bytes);
    if (vec) {
        const typename VecOf<W>::type t = *reinterpret_cast<const typename VecOf<W>::type*>(q);
        const float* tp = reinterpret_cast<const float*>(&t);
#pragma unroll
        for (int k = 0; k < W; ++k) v[k] = tp[k];
    } else {
#pragma unroll
        for (int k = 0; k < W; ++k) v[k] = q[k];
    }
}
template <int W>
__device__ __forceinline__ void stw(float* base, unsigned lane_bytes, bool vec, const float (&v)[W]) {
    float* q = reinterpret_cast<float*>(reinterpret_cast<char*>(base) + lane_bytes);
    if (vec) {
        typename VecOf<W>::type t;
        float* tp = reinterpret_cast<float*>(&t);
#pragma unroll
        for (int k = 0; k < W; ++k) tp[k] = v[k];
        *reinterpret_cast<typename VecOf<W>::type*>(q) = t;
    } else {
#pragma unroll
        for (int k = 0; k < W; ++k) q[k] = v[k];
    }
}

// row0 / col0: first row / column of the wave's tile (wave-uniform).  acc[i][j][r] = C[row0 + i*16 + 4*kl + r][col0 + TN*ml + j].
// KEEP: the values as stored (before an `accumulate` read-modify-write) are written back to acc for the fused reductions; STORE: write C.
// ROWQ: the quad-fragment kernels' row layout when A is row-contiguous -- acc[i][..][r] is row 16*kl + 4*r + i instead of i*16 + 4*kl + r.
// FXE (TN == 4): C[r][c] += ep_arg[r / group][c] == r % group ? ep_src[r / group][c] : 0 (max-pool backward of the mini-PointNet) after the residual.
template <int ACT, int TM, int TN, bool MG, bool KEEP, bool STORE, bool ROWQ = false, bool FXE = false, typename Acc>
__device__ __forceinline__ void epilogue_rows(const GemmParams& p, Acc& acc, const int row0, const int col0, const int ml, const int kl) {
    const act_gemm_epilogue_t& e = p.epi;
    constexpr int RL = ROWQ ? 16 : 4;                                    // rows between the four lane groups kl
    if (p.partial) {                                                     // split-K: raw partial sums, [split][M][N] (workspace: 16-byte aligned, N % 4 == 0)
        float* base = p.partial + ((size_t)blockIdx.z * p.M + row0) * p.N + col0;
        const unsigned lp = (unsigned)(RL * kl * p.N + TN * ml) * 4u;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ru = ROWQ ? 4 * r + i : i * 16 + r;
                if (MG && row0 + ru + RL * kl >= p.M) continue;
                float v[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) v[j] = acc[i][j][r];
                stw<TN>(base + (size_t)ru * p.N, lp, true, v);
            }
        return;
    }
    if constexpr (ACT < 0) {                                             // run-time activation: ONE uniform branch to a body specialised for it
        switch (e.act) {                                                 // (not a switch per element: the executed path stays contiguous in the instruction cache)
            case ACT_EPI_GELU:          epilogue_rows<ACT_EPI_GELU, TM, TN, MG, KEEP, STORE, ROWQ, FXE>(p, acc, row0, col0, ml, kl); return;
            case ACT_EPI_RELU:          epilogue_rows<ACT_EPI_RELU, TM, TN, MG, KEEP, STORE, ROWQ, FXE>(p, acc, row0, col0, ml, kl); return;
            case ACT_EPI_MUL_GELU_GRAD: epilogue_rows<ACT_EPI_MUL_GELU_GRAD, TM, TN, MG, KEEP, STORE, ROWQ, FXE>(p, acc, row0, col0, ml, kl); return;
            case ACT_EPI_MUL_RELU_MASK: epilogue_rows<ACT_EPI_MUL_RELU_MASK, TM, TN, MG, KEEP, STORE, ROWQ, FXE>(p, acc, row0, col0, ml, kl); return;
            default:                    epilogue_rows<ACT_EPI_NONE, TM, TN, MG, KEEP, STORE, ROWQ, FXE>(p, acc, row0, col0, ml, kl); return;
        }
    }
    const bool vec = p.epi_vec != 0;
    constexpr int act = ACT;
    const unsigned lc = (unsigned)(RL * kl * p.ldc + TN * ml) * 4u, lx = (unsigned)(RL * kl * e.ldaux + TN * ml) * 4u;
    const unsigned lr = (unsigned)(RL * kl * e.ldr + TN * ml) * 4u, lb = (unsigned)(TN * ml) * 4u;
    float b[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) b[j] = 0.f;
    if (e.bias) ldw<TN>(b, e.bias + col0, lb, vec);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rowu = row0 + (ROWQ ? 4 * r + i : i * 16 + r), row = rowu + RL * kl;     // rowu: wave-uniform part
            if (MG && row >= p.M) continue;
            float v[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) { v[j] = acc[i][j][r] * e.alpha; if (e.bias) v[j] += b[j]; }
            switch (act) {
                case ACT_EPI_GELU:
                    if (e.aux) stw<TN>(e.aux + (size_t)rowu * e.ldaux + col0, lx, vec, v);
#pragma unroll
                    for (int j = 0; j < TN; ++j) v[j] = gelu_f(v[j]);
                    break;
                case ACT_EPI_RELU:
#pragma unroll
                    for (int j = 0; j < TN; ++j) v[j] = fmaxf(v[j], 0.f);
                    break;
                case ACT_EPI_MUL_GELU_GRAD: {
                    float x[TN]; ldw<TN>(x, e.aux + (size_t)rowu * e.ldaux + col0, lx, vec);
#pragma unroll
                    for (int j = 0; j < TN; ++j) v[j] *= gelu_grad_f(x[j]);
                    break; }
                case ACT_EPI_MUL_RELU_MASK: {
                    float x[TN]; ldw<TN>(x, e.aux + (size_t)rowu * e.ldaux + col0, lx, vec);
#pragma unroll
                    for (int j = 0; j < TN; ++j) v[j] = x[j] > 0.f ? v[j] : 0.f;
                    break; }
                default: break;
            }
            if (e.rowscale) {
                const float rs = e.rowscale[row / e.rows_per_scale];
#pragma unroll
                for (int j = 0; j < TN; ++j) v[j] *= rs;
            }
            if (e.res) {
                float q[TN];
                if (e.res_row_div > 1) ldw<TN>(q, e.res + (size_t)(row / e.res_row_div) * e.ldr + col0, lb, vec);
                else                   ldw<TN>(q, e.res + (size_t)rowu * e.ldr + col0, lr, vec);
#pragma unroll
                for (int j = 0; j < TN; ++j) v[j] += q[j];
            }
            if constexpr (FXE) {
                static_assert(!FXE || TN == 4, "scatter epilogue: four columns per lane");
                const int gsh = p.fx.group == 64 ? 6 : 5;
                const size_t o = (size_t)(row >> gsh) * p.N + col0 + TN * ml; const int pos = row & ((1 << gsh) - 1);
                const int4 ea = *reinterpret_cast<const int4*>(p.fx.ep_arg + o);
                const float4 ev = *reinterpret_cast<const float4*>(p.fx.ep_src + o);
                v[0] += ea.x == pos ? ev.x : 0.f; v[1] += ea.y == pos ? ev.y : 0.f; v[2] += ea.z == pos ? ev.z : 0.f; v[3] += ea.w == pos ? ev.w : 0.f;
            }
            if (KEEP) {
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j][r] = v[j];
            }
            if (STORE) {
                float* cb = p.C + (size_t)rowu * p.ldc + col0;
                if (e.accumulate) {
                    float q[TN]; ldw<TN>(q, cb, lc, vec);
#pragma unroll
                    for (int j = 0; j < TN; ++j) v[j] += q[j];
                }
                stw<TN>(cb, lc, vec, v);
            }
        }
}

// Scalar form for the kernels whose lanes own single columns (sgemm16_kernel: acc[i][j][r] = C[row0 + i*16 + 4*kl + r][col0 + j*16 + ml]): same
// address split (wave-uniform base + one lane offset) and the same hoisted activation branch; accesses stay 4 bytes per lane (16 lanes = 64 B).
template <int ACT, int TM, int TN, bool MG, typename Acc>
__device__ __forceinline__ void epilogue_elems(const GemmParams& p, Acc& acc, const int row0, const int col0, const int ml, const int kl) {
    const act_gemm_epilogue_t& e = p.epi;
    if (p.partial) {
        float* base = p.partial + ((size_t)blockIdx.z * p.M + row0) * p.N + col0;
        const unsigned lp = (unsigned)(4 * kl * p.N + ml) * 4u;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (MG && row0 + i * 16 + r + 4 * kl >= p.M) continue;
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    *reinterpret_cast<float*>(reinterpret_cast<char*>(base + (size_t)(i * 16 + r) * p.N + j * 16) + lp) = acc[i][j][r];
            }
        return;
    }
    if constexpr (ACT < 0) {
        switch (e.act) {
            case ACT_EPI_GELU:          epilogue_elems<ACT_EPI_GELU, TM, TN, MG>(p, acc, row0, col0, ml, kl); return;
            case ACT_EPI_RELU:          epilogue_elems<ACT_EPI_RELU, TM, TN, MG>(p, acc, row0, col0, ml, kl); return;
            case ACT_EPI_MUL_GELU_GRAD: epilogue_elems<ACT_EPI_MUL_GELU_GRAD, TM, TN, MG>(p, acc, row0, col0, ml, kl); return;
            case ACT_EPI_MUL_RELU_MASK: epilogue_elems<ACT_EPI_MUL_RELU_MASK, TM, TN, MG>(p, acc, row0, col0, ml, kl); return;
            default:                    epilogue_elems<ACT_EPI_NONE, TM, TN, MG>(p, acc, row0, col0, ml, kl); return;
        }
    }
    const unsigned lc = (unsigned)(4 * kl * p.ldc + ml) * 4u, lx = (unsigned)(4 * kl * e.ldaux + ml) * 4u, lr = (unsigned)(4 * kl * e.ldr + ml) * 4u;
    const unsigned lb = (unsigned)ml * 4u;
    auto at = [](const float* base, unsigned lane_bytes) { return reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + lane_bytes); };
    float b[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) b[j] = e.bias ? *at(e.bias + col0 + j * 16, lb) : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int rowu = row0 + i * 16 + r, row = rowu + 4 * kl;
            if (MG && row >= p.M) continue;
            float rs = 1.f;
            if (e.rowscale) rs = e.rowscale[row / e.rows_per_scale];
            const float* res_row = (e.res && e.res_row_div > 1) ? e.res + (size_t)(row / e.res_row_div) * e.ldr + col0 : nullptr;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                float v = acc[i][j][r] * e.alpha;
                if (e.bias) v += b[j];
                if constexpr (ACT == ACT_EPI_GELU) {
                    if (e.aux) *const_cast<float*>(at(e.aux + (size_t)rowu * e.ldaux + col0 + j * 16, lx)) = v;
                    v = gelu_f(v);
                } else if constexpr (ACT == ACT_EPI_RELU) {
                    v = fmaxf(v, 0.f);
                } else if constexpr (ACT == ACT_EPI_MUL_GELU_GRAD) {
                    v *= gelu_grad_f(*at(e.aux + (size_t)rowu * e.ldaux + col0 + j * 16, lx));
                } else if constexpr (ACT == ACT_EPI_MUL_RELU_MASK) {
                    v = *at(e.aux + (size_t)rowu * e.ldaux + col0 + j * 16, lx) > 0.f ? v : 0.f;
                }
                if (e.rowscale) v *= rs;
                if (e.res) v += res_row ? *at(res_row + j * 16, lb) : *at(e.res + (size_t)rowu * e.ldr + col0 + j * 16, lr);
                float* c = const_cast<float*>(at(p.C + (size_t)rowu * p.ldc + col0 + j * 16, lc));
                if (e.accumulate) v += *c;
                *c = v;
            }
        }
}

// XCD-aware remap: workgroup b is dispatched to XCD b%8; give each XCD a contiguous band of tiles (own L2)
__device__ __forceinline__ int xcd_remap(int wg, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, loc = wg >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// workgroup -> tile with the XCD placement folded in.  Default (xcd_rows == 0): XCD x (= blockIdx.x % 8) owns the x-th eighth of the
// rasterised tile order, i.e. a band of tile rows: it reads its A rows once and ALL of B.  xcd_rows = r: the XCDs form an r x (8 / r) grid of
// tile blocks, XCD (i, j) owns tile rows [i * tm / r, ...) x tile columns [j * tn / (8 / r), ...): A is fetched 8 / r times, B r times --
// less fabric traffic when B (the weight) is not much smaller than A.  Needs tiles_m % r == 0, tiles_n % (8 / r) == 0 (host-checked).
__device__ __forceinline__ void tile_of_workgroup(const GemmParams& p, int bid, int& tile_m, int& tile_n) {
    if (p.xcd_rows == 0) { tile_coords(p, xcd_remap(bid, p.tiles_m * p.tiles_n), tile_m, tile_n); return; }
    const int xcd = bid & 7, loc = bid >> 3;
    const int r = p.xcd_rows, c = 8 / r;
    const int bm = p.tiles_m / r, bn = p.tiles_n / c;                   // tile block of one XCD
    const int xi = xcd / c, xj = xcd - xi * c;
    int lm, ln;                                                          // grouped rasterisation inside the block
    const int gm = p.group_m > 1 ? p.group_m : 1;
    const int per_group = gm * bn, grp = loc / per_group, first = grp * gm;
    const int gsz = min(bm - first, gm), in_group = loc - grp * per_group;
    lm = first + in_group % gsz; ln = in_group / gsz;
    tile_m = xi * bm + lm; tile_n = xj * bn + ln;
}

// launcher of the 16x16x4-MFMA kernels (gemm16.hip); tile: 0 = 128x128, 1 = 128x64, 2 = 64x64.  FULL shapes only.
void launch_sgemm16(const GemmParams& p, int tile, int a_kmajor, int b_kmajor, dim3 grid, hipStream_t s);
// NT-only kernels with K-contiguous swizzled LDS image and ds_read_b128 operand fetch (gemm_nt16.hip)
void launch_sgemm_nt16(const GemmParams& p, int tile, dim3 grid, hipStream_t s);
// NT kernels with 32-deep K tiles (gemm_nt16.hip): tile 0 = 128x128, 1 = 128x64; full tiles only
void launch_sgemm_nt32(const GemmParams& p, int tile, dim3 grid, hipStream_t s);
// NT kernels with the hand-scheduled (asm) main loop, 32-deep K tiles (gemm_nt_asm.hip): tile 0 = 128x128, 1 = 128x64, 2 = 64x64; M tail allowed
void launch_sgemm_nt_asm(const GemmParams& p, int tile, dim3 grid, hipStream_t s);
// NN / TN kernels with the hand-scheduled main loop (gemm_q_asm.hip): tile 0 = 128x128, 1 = 64x128, 2 = 64x64, 3 = 128x64 (1..3 NN only); false = no such kernel
bool launch_sgemm_q_asm(const GemmParams& p, int tile, int a_kmajor, dim3 grid, hipStream_t s);
bool launch_sgemm_q_asm_fx(const GemmParams& p, int a_kmajor, int fx_mask, dim3 grid, hipStream_t s);    // ... with the epilogue-side max-pool backward term (NN, FX_SCATTER_EPI only)
// quad-fragment kernels for the NN / TN layouts (gemm_q16.hip): tile 0 = 128x128, 1 = 64x128 (NN only); false = no such kernel
bool launch_sgemm_q16(const GemmParams& p, int tile, int a_kmajor, int b_kmajor, dim3 grid, hipStream_t s);
// NT kernels with fused producer / consumer passes (gemm_nt16_fx.hip); fx_mask = FX_* bits; tile 0 = 128x128, 1 = 128x64.  false = no such kernel
bool launch_sgemm_nt16_fx(const GemmParams& p, int tile, int fx_mask, dim3 grid, hipStream_t s);
// ... the same on the hand-scheduled main loop (gemm_nt_asm_fx.hip; additionally K % 32 == 0): bit-identical to launch_sgemm_nt16_fx
bool launch_sgemm_nt_asm_fx(const GemmParams& p, int tile, int fx_mask, dim3 grid, hipStream_t s);
// quad-fragment 128x128 kernels with fused passes (gemm_q16_fx.hip): TN with FX_AFFINE_B (+ FX_SCATTER_A), NN with FX_SCATTER_A and / or FX_SCATTER_EPI
bool launch_sgemm_q16_fx(const GemmParams& p, int a_kmajor, int fx_mask, dim3 grid, hipStream_t s);
