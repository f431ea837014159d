// gemm_nt16_kernel.h -- the NT b128 kernel template (instantiated by gemm_nt16.hip: plain + pipelined + M-tail variants, and by
// gemm_nt16_fx.hip: the fused mini-PointNet variants; two translation units so they compile in parallel)
#pragma once
#include "gemm_common.h"
#include <type_traits>

// ---------------------------------------------------------------------------------------------------------------------------
// NT specialisation (both operands K-contiguous in memory: every forward Linear / Conv1d(k=1)): the LDS image keeps the
// global layout, [row][16 k] with NO padding, so global float4 -> ds_write_b128 needs no transpose, and one ds_read_b128
// per operand row-block feeds FOUR MFMA k-steps: lane (m = lane&15, g = lane>>4) reads k = 4g..4g+3 of its row and MFMA
// step s consumes element s of both operands (the reduction order over k is a permutation, identical for A and B).
// Bank conflicts of the b128 reads (serviced in the 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...) are removed
// by XOR-swizzling the 16-byte chunk index with H[(row>>2)&3], H = {0,3,2,1}.  The main loop then contains no VALU address
// arithmetic at all (row-block strides are ds_read immediates) and 4x fewer LDS instructions than the [k][row] kernels.
// FX (mini-PointNet fusions, BM = 128 only): FX_AFFINE_A applies the producer's BatchNorm + ReLU to A while it is staged (the activated
// tensor never exists in HBM); FX_COLSTATS leaves per-tile column (mean, sum of squared deviations) of the stored values for the
// following BatchNorm (no statistics pass over the output); FX_GROUPMAX reduces every `group` consecutive rows to their max / first
// arg-max (the max-pool over the points of a group) in the epilogue; FX_NOSTORE drops the C store when only that max is wanted.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// ---- consumer passes of the fused mini-PointNet variants, after the epilogue (acc holds the values as stored: epilogue_rows<KEEP>): shared by the
// compiler-scheduled kernel below and the hand-scheduled one (gemm_nt_asm_kernel.h).  `red`: >= 2 * BN floats of LDS no wave still reads.
template <int BM, int BN, int FX, typename Acc>
__device__ __forceinline__ void nt_fx_tail(const GemmParams& p, Acc& acc, float* red, const int tile_m, const int m0, const int n0,
                                           const int wm, const int wn, const int ml, const int kl) {
    constexpr int TM = BM / 32, TN = BN / 32;
    const int cw = wn * (BN / 2) + TN * ml;                           // + j: this lane's columns inside the tile (B rows are staged permuted)

    if constexpr ((FX & FX_GROUPMAX) != 0) {
        // max + first arg-max over every `group` consecutive rows (torch.max(feature, dim=2) over the points of a group,
        // models/dvae.py:211,214).  A wave owns 64 rows = two groups of 32 or one of 64: no cross-wave step.
        const int group = p.fx.group;                                 // 32 or 64
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + cw + j;
            float hb[2]; int hi[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {                             // half of the wave's rows: blocks 2h, 2h+1
                float best = acc[2 * h][j][0]; int bi = 4 * kl;       // local row within the half
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = acc[2 * h + ii][j][r]; const int idx = ii * 16 + 4 * kl + r;
                        if (v > best) { best = v; bi = idx; }         // ascending idx in-lane: strict '>' keeps the first maximum
                    }
#pragma unroll
                for (int off = 16; off <= 32; off <<= 1) {            // across the four 16-lane rows (kl): lowest index wins ties
                    const float ov = __shfl_xor(best, off); const int oi = __shfl_xor(bi, off);
                    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
                }
                hb[h] = best; hi[h] = bi;
            }
            if (kl == 0) {
                if (group == 32) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        size_t g = (size_t)(m0 + wm * 64 + h * 32) / 32;
                        if (p.fx.row_groups) g = (size_t)p.fx.row_groups[g];
                        p.fx.gmax[g * p.N + col] = hb[h];
                        if (p.fx.garg) p.fx.garg[g * p.N + col] = hi[h];
                    }
                } else {                                              // one group of 64 rows: the first half wins ties
                    const bool second = hb[1] > hb[0];
                    size_t g = (size_t)(m0 + wm * 64) / 64;
                    if (p.fx.row_groups) g = (size_t)p.fx.row_groups[g];
                    p.fx.gmax[g * p.N + col] = second ? hb[1] : hb[0];
                    if (p.fx.garg) p.fx.garg[g * p.N + col] = second ? hi[1] + 32 : hi[0];
                }
            }
        }
    }
    if constexpr ((FX & FX_COLSTATS) != 0) {
        // per-tile column mean and sum of squared deviations over the tile's 128 rows (two passes over the accumulators, so no
        // E[x^2] - E[x]^2 cancellation); bn_tiles_finalize merges the tiles_m partials of a column in a fixed order
        __syncthreads();                                              // the staging LDS is free now: [2 wm][BN] exchange buffer
        float csum[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i) s += (acc[i][j][0] + acc[i][j][1]) + (acc[i][j][2] + acc[i][j][3]);
            s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
            csum[j] = s;
            if (kl == 0) red[wm * BN + cw + j] = s;
        }
        __syncthreads();
        float mean[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int c = cw + j;
            mean[j] = (red[c] + red[BN + c]) * (1.0f / BM);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = acc[i][j][r] - mean[j]; q += d * d; }
            q += __shfl_xor(q, 16); q += __shfl_xor(q, 32);
            if (kl == 0) red[wm * BN + cw + j] = q;
        }
        __syncthreads();
        if (wm == 0 && kl == 0) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int c = cw + j;
                float* ts = p.fx.tile_stats + (size_t)tile_m * 2 * p.N + n0 + c;
                ts[0] = mean[j];
                ts[p.N] = red[c] + red[BN + c];
            }
        }
        (void)csum;
    }
}

#ifndef ACT_NT16_BUFFER_LOADS
#define ACT_NT16_BUFFER_LOADS 1     // 0: per-thread 64-bit pointers in the K loop as in rounds 1-5 (A/B builds: ACT_HIPCC_EXTRA=-DACT_NT16_BUFFER_LOADS=0)
#endif
#ifndef NT16_OCC_SMALL
#define NT16_OCC_SMALL 3
#endif
// PIPE: software-pipelined main loop -- the fragments of K-tile t+1 are read from LDS while the MFMAs of tile t run (two register sets),
// tile t+2 is already in flight from global memory, and the barrier waits for LDS traffic only (bare s_barrier: the global loads stay in
// flight across it).  Same products in the same order: results are bit-identical to the plain loop.  Needs an even number of K-tiles
// (K per split % 32 == 0); loads past the end are clamped to the last tile and land in an LDS buffer nobody reads.  Pays on long K
// (+3-4 % at K = 3,072, benchmarks/micro/nt_pipe.hip) and on launches with few workgroups per CU.
template <int BM, int BN, bool MG = false, int FX = 0, bool PIPE = false, int ACT = -1>
__global__ __launch_bounds__(256, (BM * BN <= 128 * 64 && FX == 0) ? NT16_OCC_SMALL : 3) void sgemm_nt16_kernel(const GemmParams p) {
    static_assert(!PIPE || (FX == 0 && !MG), "pipelined loop: plain full tiles");
    constexpr int BK = 16;
    constexpr int TM = BM / 32, TN = BN / 32;
    constexpr int NA = BM * BK / 1024, NB = BN * BK / 1024;
    static_assert(FX == 0 || (BM == 128 && !MG), "fused variants: 128-row tiles, no M tail");
    __shared__ __attribute__((aligned(16))) float As[2][BM * BK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * BK];
    __shared__ __attribute__((aligned(16))) float Sx[(FX & FX_AFFINE_A) ? 2048 : 4];      // scale[K] | shift[K] of the A-side affine map (K <= 1024)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int tile_m, tile_n;
    tile_of_workgroup(p, blockIdx.x, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = blockIdx.z * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int ntiles = (kend - kbeg) / BK;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // global -> register staging: thread v owns the float4 (row = v>>2 (+64 per extra load), chunk = v&3) of each operand tile;
    // rows 64 apart share the swizzle, so the extra loads are plain immediates on one pointer / one LDS offset per operand
    const int srow = tid >> 2, sch = tid & 3;
    // (fused variants: the rows may be the listed groups of a larger tensor, GemmFx::row_groups)
    auto phys_row = [&](int m) -> size_t {
        if constexpr (FX != 0) { if (p.fx.row_groups) return (size_t)p.fx.row_groups[m / p.fx.group] * p.fx.group + m % p.fx.group; }
        return (size_t)m;
    };
    const float* ga = p.A + (MG ? (size_t)min(m0 + srow, p.M - 1) : phys_row(m0 + srow)) * p.lda + kbeg + sch * 4;
    // B rows are staged PERMUTED (see epilogue_rows): LDS row j*16 + m of every 16*TN-row block holds global row TN*m + j of that block
    const int srow_b = (srow / (16 * TN)) * (16 * TN) + TN * (srow & 15) + (srow % (16 * TN)) / 16;
    const float* gb = p.B + (size_t)(n0 + srow_b) * p.ldb + kbeg + sch * 4;
    const int s_off = srow * 16 + 4 * (sch ^ ((4 - ((srow >> 2) & 3)) & 3));
    const ptrdiff_t stride_a = MG ? (ptrdiff_t)(min(m0 + srow + 64, p.M - 1) - min(m0 + srow, p.M - 1)) * p.lda
                                  : ((ptrdiff_t)phys_row(m0 + srow + 64) - (ptrdiff_t)phys_row(m0 + srow)) * p.lda;   // (a listed group may lie below)
    const size_t stride_b = (size_t)64 * p.ldb;
    // staging registers as named scalars (NA, NB <= 2): arrays indexed inside the helper lambdas are not promoted to
    // registers by hipcc here and would round-trip through scratch memory in the main loop
    float4 ra0, ra1, rb0, rb1;
    ra0 = ra1 = rb0 = rb1 = make_float4(0.f, 0.f, 0.f, 0.f);
    // round 6: operand addresses as a wave-uniform base (SGPRs, advanced per K tile by scalar adds) + ONE 32-bit lane offset per load, so that the loop holds no
    // 64-bit VALU pointer increments (8 v_lshl_add_u64 per K tile of 64 MFMAs before: on gfx950 every VALU instruction is matrix-pipe time).  The per-thread
    // pointer form stays for the M-tail kernels and for listed row groups, whose rows are not affine in the tile origin.
    const bool lin = !MG && !(FX != 0 && p.fx.row_groups != nullptr) && (size_t)BM * (size_t)p.lda < (1u << 29) && (size_t)BN * (size_t)p.ldb < (1u << 29);
    const float* a_base = p.A + (size_t)m0 * p.lda + kbeg;
    const float* b_base = p.B + (size_t)n0 * p.ldb + kbeg;
    // (byte offsets from the tile's first row / column)
    const unsigned a_of0 = ((unsigned)srow * (unsigned)p.lda + sch * 4) * 4u, a_of1 = a_of0 + 256u * (unsigned)p.lda;
    const unsigned b_of0 = ((unsigned)srow_b * (unsigned)p.ldb + sch * 4) * 4u, b_of1 = b_of0 + 256u * (unsigned)p.ldb;
    const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_base), 0, -1, 0x00020000);     // raw buffer, no bounds (full tiles only)
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b_base), 0, -1, 0x00020000);
    auto load_g = [&](auto lin_c, int t) {
        if constexpr (decltype(lin_c)::value) {                      // buffer_load_dwordx4 v, v_off, s[rsrc], s_off offen: no address arithmetic on the vector ALU
            const int so = t * (BK * 4);
            ra0 = buf_load4(rs_a, a_of0, so);
            if constexpr (NA > 1) ra1 = buf_load4(rs_a, a_of1, so);
            rb0 = buf_load4(rs_b, b_of0, so);
            if constexpr (NB > 1) rb1 = buf_load4(rs_b, b_of1, so);
            return;
        }
        ra0 = *reinterpret_cast<const float4*>(ga + t * BK);
        if constexpr (NA > 1) ra1 = *reinterpret_cast<const float4*>(ga + stride_a + t * BK);
        rb0 = *reinterpret_cast<const float4*>(gb + t * BK);
        if constexpr (NB > 1) rb1 = *reinterpret_cast<const float4*>(gb + stride_b + t * BK);
    };
    if constexpr ((FX & FX_AFFINE_A) != 0) {
        for (int k = tid; k < p.K; k += 256) { Sx[k] = p.fx.a_scale[k]; Sx[1024 + k] = p.fx.a_shift[k]; }
        __syncthreads();
    }
    auto store_lds = [&](int buf, int t) {
        if constexpr ((FX & FX_AFFINE_A) != 0) {                      // A' = relu(A * scale[k] + shift[k]) for this thread's 4 k of tile t
            const float4 sc = *reinterpret_cast<const float4*>(&Sx[kbeg + t * BK + sch * 4]);
            const float4 sh = *reinterpret_cast<const float4*>(&Sx[1024 + kbeg + t * BK + sch * 4]);
            ra0.x = fmaxf(ra0.x * sc.x + sh.x, 0.f); ra0.y = fmaxf(ra0.y * sc.y + sh.y, 0.f);
            ra0.z = fmaxf(ra0.z * sc.z + sh.z, 0.f); ra0.w = fmaxf(ra0.w * sc.w + sh.w, 0.f);
            if constexpr (NA > 1) {
                ra1.x = fmaxf(ra1.x * sc.x + sh.x, 0.f); ra1.y = fmaxf(ra1.y * sc.y + sh.y, 0.f);
                ra1.z = fmaxf(ra1.z * sc.z + sh.z, 0.f); ra1.w = fmaxf(ra1.w * sc.w + sh.w, 0.f);
            }
        }
        *reinterpret_cast<float4*>(&As[buf][s_off]) = ra0;
        if constexpr (NA > 1) *reinterpret_cast<float4*>(&As[buf][s_off + 1024]) = ra1;
        *reinterpret_cast<float4*>(&Bs[buf][s_off]) = rb0;
        if constexpr (NB > 1) *reinterpret_cast<float4*>(&Bs[buf][s_off + 1024]) = rb1;
    };

    const int kl = lane >> 4, ml = lane & 15;
    const int hsw = (4 - ((ml >> 2) & 3)) & 3;                        // row-block bases are multiples of 16: H depends on ml only
    const int a_off = (wm * (BM / 2) + ml) * 16 + 4 * (kl ^ hsw);
    const int b_off = (wn * (BN / 2) + ml) * 16 + 4 * (kl ^ hsw);
    // the K loop, instantiated for both address forms (a run-time select inside ONE loop made the compiler fold them back into per-thread 64-bit pointers)
    auto k_loop = [&](auto lin_c) {
    if (ntiles > 0) {
        load_g(lin_c, 0);
        store_lds(0, 0);
        __syncthreads();
    }
    auto compute = [&](int buf) {
        float4 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(&As[buf][a_off + i * 256]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(&Bs[buf][b_off + j * 256]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
    };
    if constexpr (PIPE) {
        struct Frag { float4 a[TM], b[TN]; };
        auto read_frags = [&](Frag& f, int buf) {
#pragma unroll
            for (int i = 0; i < TM; ++i) f.a[i] = *reinterpret_cast<const float4*>(&As[buf][a_off + i * 256]);
#pragma unroll
            for (int j = 0; j < TN; ++j) f.b[j] = *reinterpret_cast<const float4*>(&Bs[buf][b_off + j * 256]);
        };
        auto mfma_tile = [&](const Frag& f) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[i].x, f.b[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[i].y, f.b[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[i].z, f.b[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[i].w, f.b[j].w, acc[i][j], 0, 0, 0);
        };
        auto lds_barrier = [&]() {
            __builtin_amdgcn_s_waitcnt(0xc07f);             // lgkmcnt(0); vmcnt untouched
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        if (ntiles > 0) {                                   // (tile 0 is in LDS buffer 0 and visible: prologue above)
            Frag F0, F1;
            const int last = ntiles - 1;
            load_g(lin_c, min(1, last));
            read_frags(F0, 0);
            store_lds(1, 0); load_g(lin_c, min(2, last));
            lds_barrier();
            for (int t = 0; t < ntiles; t += 2) {
                read_frags(F1, 1);
                mfma_tile(F0);
                store_lds(0, 0); load_g(lin_c, min(t + 3, last));
                lds_barrier();
                read_frags(F0, 0);
                mfma_tile(F1);
                store_lds(1, 0); load_g(lin_c, min(t + 4, last));
                lds_barrier();
            }
        }
    } else {
    for (int t = 0; t + 1 < ntiles; ++t) {              // steady state: fetch tile t+1 while computing tile t
        load_g(lin_c, t + 1);
        compute(t & 1);
        store_lds((t & 1) ^ 1, t + 1);
        __syncthreads();
    }
    if (ntiles > 0) compute((ntiles - 1) & 1);
    }
    };
    if constexpr (MG || !ACT_NT16_BUFFER_LOADS) k_loop(std::false_type{});
    else { if (lin) k_loop(std::true_type{}); else k_loop(std::false_type{}); }

    {
        const int wu = __builtin_amdgcn_readfirstlane(wave);
        epilogue_rows<ACT, TM, TN, MG, FX != 0, (FX & FX_NOSTORE) == 0>(p, acc, m0 + (wu >> 1) * (BM / 2), n0 + (wu & 1) * (BN / 2), ml, kl);
    }
    if constexpr (FX == 0) return;
    if (p.partial) return;
    nt_fx_tail<BM, BN, FX>(p, acc, &As[0][0], tile_m, m0, n0, wm, wn, ml, kl);
}

