// gemm_q16_kernel.h -- the quad-fragment NN / TN kernel template (instantiated by gemm_q16.hip and gemm_q16_fx.hip)
#pragma once
#include "gemm_common.h"

// ---------------------------------------------------------------------------------------------------------------------------
// "Quad-fragment" kernels for the layouts with a ROW-contiguous operand: input gradients dX = dY . W (B stored [K][N]) and weight
// gradients dW = dY^T . X (A stored [K][M], B stored [K][N]).  No transpose anywhere: the LDS image of a row-contiguous operand
// keeps the global [k][rows] layout (float4 -> ds_write_b128), and ONE ds_read_b128 along the rows feeds the four 16-wide blocks
// of a wave's 64 rows for one MFMA k-step: lane (q = lane&15, g = lane>>4) reads rows 4q..4q+3 of k-row 4g+s, and element i of the
// quad is the operand of block i, whose 16 MFMA rows are therefore the INTERLEAVED rows {4r+i}.  The interleave is undone for free
// in the epilogue (a lane then owns 4 consecutive columns -> one float4 store instead of four scalar stores).  A K-contiguous
// operand keeps the scheme of sgemm_nt16_kernel ([row][16 k], swizzled, one b128 = four k-steps); both use k = 4g + s for lane group
// g at k-step s, so the two fetch schemes combine freely.  Per 16-deep K-tile: 8 ds_read_b128 for 64 MFMAs in every layout.
// Constraint: a row-contiguous operand needs a 64-wide wave extent, i.e. BM = 128 when A is [K][M], BN = 128 when B is [K][N].
// FXB (weight gradients of the mini-PointNet): B'[k,n] = relu(B[k,n] * b_scale[n] + b_shift[n]) while B is staged -- the activated input of
// the layer is recomputed from the stored pre-BatchNorm tensor instead of being kept (a thread's float4 always covers the same 4 columns).
// FXA (max-pool backward on load): the A operand is virtual, A[r][c] = sa_arg[r/group][c] == r % group ? sa_src[r/group][c] : 0 with lda = channels:
// the scattered gradient of torch.max(feature, dim=2) is generated while it is staged instead of being written (and read twice) as an
// [R][C] tensor.  FXE: the same term added in the epilogue, C[r][c] += ep_arg[r/group][c] == r % group ? ep_src[r/group][c] : 0.
template <int BM, int BN, bool A_K, bool B_K, bool MG = false, bool FXB = false, bool FXA = false, bool FXE = false, int ACT = -1>
__global__ __launch_bounds__(256, 3) void sgemm_q16_kernel(const GemmParams p) {
    static_assert(!FXB || !B_K, "FXB: row-contiguous B");
    static_assert(!(FXA || FXE) || (BM == 128 && !MG), "fused max-pool backward: 128-row tiles, no M tail");
    static_assert(!FXE || !B_K, "FXE: float4 epilogue");
    static_assert(A_K || BM == 128, "row-contiguous A needs BM = 128");
    static_assert(B_K || BN == 128 || (BN == 64 && A_K), "row-contiguous B needs a 64-wide wave extent: BN = 128 (2 x 2 waves) or 64 (4 x 1 waves, NN)");
    static_assert(!(A_K && B_K), "NT is sgemm_nt16_kernel");
    static_assert(BN == 128 || !(FXB || FXA || FXE), "fused variants: BN = 128");
    constexpr int BK = 16;
    constexpr int WN = (BN == 64) ? 1 : 2, WM = 4 / WN;                 // wave grid: 2 x 2, or 4 x 1 for the 64-column NN tiles
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    constexpr int NA = BM * BK / 1024, NB = BN * BK / 1024;
    __shared__ __attribute__((aligned(16))) float As[2][BM * BK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * BK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
    const int wg = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    int tile_m, tile_n;
    tile_coords(p, wg, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = blockIdx.z * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int ntiles = (kend - kbeg) / BK;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- global -> register -> LDS staging
    // K-contiguous operand: thread owns float4 (row = tid>>2 (+64), chunk = tid&3), swizzled 16-byte chunks (as sgemm_nt16_kernel)
    // row-contiguous operand: thread owns float4 #(tid + 256 i) of the [16][rows] tile, stored at the same index
    const int srow = tid >> 2, sch = tid & 3;
    const int s_off_k = srow * 16 + 4 * (sch ^ ((4 - ((srow >> 2) & 3)) & 3));
    const float* ga; size_t ga_step, ga_second;
    if (A_K) {
        const int r0 = MG ? min(m0 + srow, p.M - 1) : m0 + srow, r1 = MG ? min(m0 + srow + 64, p.M - 1) : m0 + srow + 64;
        ga = p.A + (size_t)r0 * p.lda + kbeg + sch * 4; ga_step = BK; ga_second = (size_t)(r1 - r0) * p.lda;
    } else {                                            // [K][M]: float4 v -> k = v / (BM/4), m4 = v % (BM/4); BM = 128: second load = +8 k-rows
        ga = p.A + (size_t)(kbeg + tid / (BM / 4)) * p.lda + m0 + (tid % (BM / 4)) * 4; ga_step = (size_t)BK * p.lda; ga_second = (size_t)8 * p.lda;
    }
    const float* gb; size_t gb_step, gb_second;
    if (B_K) {
        gb = p.B + (size_t)(n0 + srow) * p.ldb + kbeg + sch * 4; gb_step = BK; gb_second = (size_t)64 * p.ldb;
    } else {
        gb = p.B + (size_t)(kbeg + tid / (BN / 4)) * p.ldb + n0 + (tid % (BN / 4)) * 4; gb_step = (size_t)BK * p.ldb; gb_second = (size_t)8 * p.ldb;
    }
    float4 ra0, ra1, rb0, rb1;
    ra0 = ra1 = rb0 = rb1 = make_float4(0.f, 0.f, 0.f, 0.f);
    // FXA: the loads fetch (value, arg) of the group row; the select against the row's position in its group happens in store_lds, so the
    // loads stay in flight across compute() like the plain ones
    int4 qa0 = make_int4(0, 0, 0, 0), qa1 = make_int4(0, 0, 0, 0);
    int pos0 = 0, pos1 = 0;                                 // r % group of the rows this thread stages (A_K: fixed; else: per K-tile)
    const int gsh = (FXA || FXE) ? (p.fx.group == 64 ? 6 : 5) : 0, gmask = (1 << gsh) - 1;
    size_t fa0 = 0, fa1 = 0;                                // A_K: offsets of the two group rows (+ k); else recomputed per tile
    if constexpr (FXA && A_K) {
        const int r0 = m0 + srow, r1 = r0 + 64;
        pos0 = r0 & gmask; pos1 = r1 & gmask;
        fa0 = (size_t)(r0 >> gsh) * p.lda + kbeg + sch * 4; fa1 = (size_t)(r1 >> gsh) * p.lda + kbeg + sch * 4;
    }
    auto load_g = [&](int t) {
        if constexpr (FXA) {
            if constexpr (A_K) {
                ra0 = *reinterpret_cast<const float4*>(p.fx.sa_src + fa0 + t * BK); qa0 = *reinterpret_cast<const int4*>(p.fx.sa_arg + fa0 + t * BK);
                ra1 = *reinterpret_cast<const float4*>(p.fx.sa_src + fa1 + t * BK); qa1 = *reinterpret_cast<const int4*>(p.fx.sa_arg + fa1 + t * BK);
            } else {                                        // [K][M]: this thread's k-rows of tile t are r, r + 8
                const int r = kbeg + t * BK + tid / (BM / 4), c = m0 + (tid % (BM / 4)) * 4;
                pos0 = r & gmask; pos1 = (r + 8) & gmask;
                const size_t o0 = (size_t)(r >> gsh) * p.lda + c, o1 = (size_t)((r + 8) >> gsh) * p.lda + c;
                ra0 = *reinterpret_cast<const float4*>(p.fx.sa_src + o0); qa0 = *reinterpret_cast<const int4*>(p.fx.sa_arg + o0);
                ra1 = *reinterpret_cast<const float4*>(p.fx.sa_src + o1); qa1 = *reinterpret_cast<const int4*>(p.fx.sa_arg + o1);
            }
        } else {
            ra0 = *reinterpret_cast<const float4*>(ga + t * ga_step);
            if constexpr (NA > 1) ra1 = *reinterpret_cast<const float4*>(ga + ga_second + t * ga_step);
        }
        rb0 = *reinterpret_cast<const float4*>(gb + t * gb_step);
        if constexpr (NB > 1) rb1 = *reinterpret_cast<const float4*>(gb + gb_second + t * gb_step);
    };
    const int sa_off = A_K ? s_off_k : tid * 4, sb_off = B_K ? s_off_k : tid * 4;
    float4 bsc = make_float4(1.f, 1.f, 1.f, 1.f), bsh = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (FXB) {
        bsc = *reinterpret_cast<const float4*>(p.fx.b_scale + n0 + (tid % (BN / 4)) * 4);
        bsh = *reinterpret_cast<const float4*>(p.fx.b_shift + n0 + (tid % (BN / 4)) * 4);
    }
    auto store_lds = [&](int buf) {
        if constexpr (FXA) {
            ra0.x = qa0.x == pos0 ? ra0.x : 0.f; ra0.y = qa0.y == pos0 ? ra0.y : 0.f; ra0.z = qa0.z == pos0 ? ra0.z : 0.f; ra0.w = qa0.w == pos0 ? ra0.w : 0.f;
            ra1.x = qa1.x == pos1 ? ra1.x : 0.f; ra1.y = qa1.y == pos1 ? ra1.y : 0.f; ra1.z = qa1.z == pos1 ? ra1.z : 0.f; ra1.w = qa1.w == pos1 ? ra1.w : 0.f;
        }
        if constexpr (FXB) {
            rb0.x = fmaxf(rb0.x * bsc.x + bsh.x, 0.f); rb0.y = fmaxf(rb0.y * bsc.y + bsh.y, 0.f);
            rb0.z = fmaxf(rb0.z * bsc.z + bsh.z, 0.f); rb0.w = fmaxf(rb0.w * bsc.w + bsh.w, 0.f);
            if constexpr (NB > 1) {
                rb1.x = fmaxf(rb1.x * bsc.x + bsh.x, 0.f); rb1.y = fmaxf(rb1.y * bsc.y + bsh.y, 0.f);
                rb1.z = fmaxf(rb1.z * bsc.z + bsh.z, 0.f); rb1.w = fmaxf(rb1.w * bsc.w + bsh.w, 0.f);
            }
        }
        *reinterpret_cast<float4*>(&As[buf][sa_off]) = ra0;
        if constexpr (NA > 1) *reinterpret_cast<float4*>(&As[buf][sa_off + 1024]) = ra1;
        *reinterpret_cast<float4*>(&Bs[buf][sb_off]) = rb0;
        if constexpr (NB > 1) *reinterpret_cast<float4*>(&Bs[buf][sb_off + 1024]) = rb1;
    };

    if (ntiles > 0) { load_g(0); store_lds(0); __syncthreads(); }
    const int kl = lane >> 4, ml = lane & 15;
    const int hsw = (4 - ((ml >> 2) & 3)) & 3;
    // K-contiguous: fragment i at +i*256 floats; row-contiguous: k-step s at +s*rows floats
    const int a_off = A_K ? (wm * (BM / WM) + ml) * 16 + 4 * (kl ^ hsw) : (4 * kl) * BM + wm * 64 + 4 * ml;
    const int b_off = B_K ? (wn * (BN / WN) + ml) * 16 + 4 * (kl ^ hsw) : (4 * kl) * BN + wn * 64 + 4 * ml;
    auto compute = [&](int buf) {
        float4 af[4], bf[4];                            // A_K: af[i] = 4 k-steps of block i; else af[s] = 4 blocks of k-step s  (TM, TN <= 4)
#pragma unroll
        for (int i = 0; i < (A_K ? TM : 4); ++i) af[i] = *reinterpret_cast<const float4*>(&As[buf][a_off + (A_K ? i * 256 : i * BM)]);
#pragma unroll
        for (int j = 0; j < (B_K ? TN : 4); ++j) bf[j] = *reinterpret_cast<const float4*>(&Bs[buf][b_off + (B_K ? j * 256 : j * BN)]);
        auto el = [](const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); };
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float a = A_K ? el(af[i], s) : el(af[s], i);
                    const float b = B_K ? el(bf[j], s) : el(bf[s], j);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i][j], 0, 0, 0);
                }
    };
    for (int t = 0; t + 1 < ntiles; ++t) {
        load_g(t + 1);
        compute(t & 1);
        store_lds((t & 1) ^ 1);
        __syncthreads();
    }
    if (ntiles > 0) compute((ntiles - 1) & 1);

    // epilogue.  D layout of one 16x16 block: MFMA col = ml, MFMA row = 4*kl + reg.
    //   actual row of (block i, MFMA row r) = A_K ? wm*BM/2 + 16 i + r : wm*64 + 4 r + i
    //   actual col of (block j, MFMA col c) = B_K ? wn*BN/2 + 16 j + c : wn*64 + 4 c + j     (-> 4 consecutive columns per lane)
    if constexpr (!B_K) {                                              // four consecutive columns per lane: vector epilogue (gemm_common.h)
        const int wu = __builtin_amdgcn_readfirstlane(wave);
        const int wmu = WN == 2 ? wu >> 1 : wu, wnu = WN == 2 ? wu & 1 : 0;
        epilogue_rows<ACT, TM, TN, MG, false, true, !A_K, FXE>(p, acc, m0 + (A_K ? wmu * (BM / WM) : wmu * 64), n0 + wnu * 64, ml, kl);
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int r16 = kl * 4 + r;
            const int row = m0 + (A_K ? wm * (BM / WM) + i * 16 + r16 : wm * 64 + 4 * r16 + i);
            if (MG && row >= p.M) continue;
            {                                                        // (B K-contiguous with A row-contiguous: not instantiated on this path)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = n0 + wn * (BN / WN) + j * 16 + ml;
                    float v = acc[i][j][r];
                    if (p.partial) p.partial[((size_t)blockIdx.z * p.M + row) * p.N + col] = v;
                    else {
                        v = epilogue_apply(p.epi, v, row, col);
                        float* c = p.C + (size_t)row * p.ldc + col;
                        if (p.epi.accumulate) v += *c;
                        *c = v;
                    }
                }
            }
        }
}

