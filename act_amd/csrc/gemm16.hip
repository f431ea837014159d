// gemm16.hip -- fp32 GEMM main loop on v_mfma_f32_16x16x4_f32 (32-cycle issue, 4 accumulator registers per 16x16 tile).
//
// Same contract, operand layouts, split-K and fused epilogue as gemm.hip; FULL shapes only (M % BM == N % BN == 0, every
// K range a multiple of 16, float4-aligned operands) -- the guarded 32x32x2 kernels of gemm.hip cover everything else.
// Why a second MFMA shape: on gfx950 the 64-cycle 32x32x2 stream of gemm.hip saturates at ~0.83 of the f32 peak even with
// all memory operations removed (ablation in DESIGN.md), the finer-grained 16x16x4 stream interleaves better.
//
// Workgroup = 256 threads = 2x2 waves; wave tile (BM/2)x(BN/2) = TM x TN tiles of 16x16; operands staged in LDS as
// [k][row] with row stride = rows + 16 (so the four k-rows of one MFMA operand fetch fall into disjoint bank halves) and a
// per-k-quad XOR swizzle of the row index (col ^ 8*(k>>2)) that makes the transposing ds_write_b32 stores of K-major
// operands conflict-free as well.  Fragment layout: A[m = lane&15][k = lane>>4], B[k = lane>>4][n = lane&15],
// C/D: col = lane&15, row = 4*(lane>>4) + reg.
#include "gemm_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// MG (A_K only): M need not be a multiple of BM -- A rows are clamped on load, C rows guarded on store (token counts such as
// 32 clouds x 65 tokens = 2080 rows keep the fast path instead of the fully guarded 32x32x2 kernels)
template <int BM, int BN, bool A_K, bool B_K, bool MG = false>
__global__ __launch_bounds__(256, 4) void sgemm16_kernel(const GemmParams p) {
    constexpr int BK = 16;
    constexpr int LDA_S = BM + 16, LDB_S = BN + 16;
    constexpr int TM = BM / 32, TN = BN / 32;           // 16x16 tiles per wave in M and N
    constexpr int NA = BM * BK / 1024, NB = BN * BK / 1024;
    __shared__ __attribute__((aligned(16))) float As[2][BK * LDA_S];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB_S];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
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

    float4 ra[NA], rb[NB];
    auto load_g = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int v = tid + 256 * i;
            if (A_K) ra[i] = *reinterpret_cast<const float4*>(p.A + (size_t)(MG ? min(m0 + (v >> 2), p.M - 1) : m0 + (v >> 2)) * p.lda + k0 + (v & 3) * 4);
            else     ra[i] = *reinterpret_cast<const float4*>(p.A + (size_t)(k0 + v / (BM / 4)) * p.lda + m0 + (v % (BM / 4)) * 4);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int v = tid + 256 * i;
            if (B_K) rb[i] = *reinterpret_cast<const float4*>(p.B + (size_t)(n0 + (v >> 2)) * p.ldb + k0 + (v & 3) * 4);
            else     rb[i] = *reinterpret_cast<const float4*>(p.B + (size_t)(k0 + v / (BN / 4)) * p.ldb + n0 + (v % (BN / 4)) * 4);
        }
    };
    // LDS element (k, row) lives at  k*LD + (row ^ 8*(k>>2))
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int v = tid + 256 * i;
            if (A_K) {                                   // 4 consecutive k of one row: k = 4*kq + c, swizzle 8*kq
                const int kq = v & 3, row = (v >> 2) ^ (kq * 8);
                float* d = &As[buf][(kq * 4) * LDA_S + row];
                d[0] = ra[i].x; d[LDA_S] = ra[i].y; d[2 * LDA_S] = ra[i].z; d[3 * LDA_S] = ra[i].w;
            } else {                                     // 4 consecutive rows of one k
                const int k = v / (BM / 4), row = ((v % (BM / 4)) * 4) ^ ((k >> 2) * 8);
                *reinterpret_cast<float4*>(&As[buf][k * LDA_S + row]) = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int v = tid + 256 * i;
            if (B_K) {
                const int kq = v & 3, row = (v >> 2) ^ (kq * 8);
                float* d = &Bs[buf][(kq * 4) * LDB_S + row];
                d[0] = rb[i].x; d[LDB_S] = rb[i].y; d[2 * LDB_S] = rb[i].z; d[3 * LDB_S] = rb[i].w;
            } else {
                const int k = v / (BN / 4), row = ((v % (BN / 4)) * 4) ^ ((k >> 2) * 8);
                *reinterpret_cast<float4*>(&Bs[buf][k * LDB_S + row]) = rb[i];
            }
        }
    };

    if (ntiles > 0) {
        load_g(kbeg);
        store_lds(0);
        __syncthreads();
    }
    const int kl = lane >> 4, ml = lane & 15;
    const int a_col = wm * (BM / 2) + ml, b_col = wn * (BN / 2) + ml;
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) load_g(kbeg + (t + 1) * BK);
        const float* as = As[buf];
        const float* bs = Bs[buf];
#pragma unroll
        for (int kq = 0; kq < BK / 4; ++kq) {            // one MFMA k-step = the 4 k-rows 4*kq .. 4*kq+3
            float a[TM], b[TN];
            const int krow = kq * 4 + kl, swz = kq * 8;
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = as[krow * LDA_S + ((a_col + i * 16) ^ swz)];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = bs[krow * LDB_S + ((b_col + j * 16) ^ swz)];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < ntiles) store_lds(buf ^ 1);
        __syncthreads();
    }

    // epilogue: C/D layout of the 16x16 MFMA: col = lane&15, row = 4*(lane>>4) + r
    {
        const int wu = __builtin_amdgcn_readfirstlane(wave);
        epilogue_elems<-1, TM, TN, MG>(p, acc, m0 + (wu >> 1) * (BM / 2), n0 + (wu & 1) * (BN / 2), ml, kl);
    }
}

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
    const float* ga = p.A + (size_t)(MG ? min(m0 + srow, p.M - 1) : m0 + srow) * p.lda + kbeg + sch * 4;
    // B rows are staged PERMUTED (see epilogue_rows): LDS row j*16 + m of every 16*TN-row block holds global row TN*m + j of that block
    const int srow_b = (srow / (16 * TN)) * (16 * TN) + TN * (srow & 15) + (srow % (16 * TN)) / 16;
    const float* gb = p.B + (size_t)(n0 + srow_b) * p.ldb + kbeg + sch * 4;
    const int s_off = srow * 16 + 4 * (sch ^ ((4 - ((srow >> 2) & 3)) & 3));
    const size_t stride_a = MG ? (size_t)(min(m0 + srow + 64, p.M - 1) - min(m0 + srow, p.M - 1)) * p.lda : (size_t)64 * p.lda;
    const size_t stride_b = (size_t)64 * p.ldb;
    // staging registers as named scalars (NA, NB <= 2): arrays indexed inside the helper lambdas are not promoted to
    // registers by hipcc here and would round-trip through scratch memory in the main loop
    float4 ra0, ra1, rb0, rb1;
    ra0 = ra1 = rb0 = rb1 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_g = [&](int t) {
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

    if (ntiles > 0) {
        load_g(0);
        store_lds(0, 0);
        __syncthreads();
    }
    const int kl = lane >> 4, ml = lane & 15;
    const int hsw = (4 - ((ml >> 2) & 3)) & 3;                        // row-block bases are multiples of 16: H depends on ml only
    const int a_off = (wm * (BM / 2) + ml) * 16 + 4 * (kl ^ hsw);
    const int b_off = (wn * (BN / 2) + ml) * 16 + 4 * (kl ^ hsw);
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
            load_g(min(1, last));
            read_frags(F0, 0);
            store_lds(1, 0); load_g(min(2, last));
            lds_barrier();
            for (int t = 0; t < ntiles; t += 2) {
                read_frags(F1, 1);
                mfma_tile(F0);
                store_lds(0, 0); load_g(min(t + 3, last));
                lds_barrier();
                read_frags(F0, 0);
                mfma_tile(F1);
                store_lds(1, 0); load_g(min(t + 4, last));
                lds_barrier();
            }
        }
    } else {
    for (int t = 0; t + 1 < ntiles; ++t) {              // steady state: fetch tile t+1 while computing tile t
        load_g(t + 1);
        compute(t & 1);
        store_lds((t & 1) ^ 1, t + 1);
        __syncthreads();
    }
    if (ntiles > 0) compute((ntiles - 1) & 1);
    }

    {
        const int wu = __builtin_amdgcn_readfirstlane(wave);
        epilogue_rows<ACT, TM, TN, MG, FX != 0, (FX & FX_NOSTORE) == 0>(p, acc, m0 + (wu >> 1) * (BM / 2), n0 + (wu & 1) * (BN / 2), ml, kl);
    }
    if constexpr (FX == 0) return;
    if (p.partial) return;
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
                        const size_t g = (size_t)(m0 + wm * 64 + h * 32) / 32;
                        p.fx.gmax[g * p.N + col] = hb[h];
                        if (p.fx.garg) p.fx.garg[g * p.N + col] = hi[h];
                    }
                } else {                                              // one group of 64 rows: the first half wins ties
                    const bool second = hb[1] > hb[0];
                    const size_t g = (size_t)(m0 + wm * 64) / 64;
                    p.fx.gmax[g * p.N + col] = second ? hb[1] : hb[0];
                    if (p.fx.garg) p.fx.garg[g * p.N + col] = second ? hi[1] + 32 : hi[0];
                }
            }
        }
    }
    if constexpr ((FX & FX_COLSTATS) != 0) {
        // per-tile column mean and sum of squared deviations over the tile's 128 rows (two passes over the accumulators, so no
        // E[x^2] - E[x]^2 cancellation); bn_tiles_finalize merges the tiles_m partials of a column in a fixed order
        __syncthreads();                                              // As is free now: [2 wm][BN] exchange buffer
        float* red = &As[0][0];
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

bool launch_sgemm_nt16_fx(const GemmParams& p, int tile, int fx, dim3 grid, hipStream_t s) {
#define FXL(BN_, MASK) \
    if (p.epi.act == ACT_EPI_NONE) hipLaunchKernelGGL((sgemm_nt16_kernel<128, BN_, false, MASK, false, ACT_EPI_NONE>), grid, dim3(256), 0, s, p); \
    else                           hipLaunchKernelGGL((sgemm_nt16_kernel<128, BN_, false, MASK>), grid, dim3(256), 0, s, p); \
    return true
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

// ---------------------------------------------------------------------------------------------------------------------------
// NT kernel with 32-deep K tiles (round 3).  benchmarks/micro/load_path.hip: the L2 -> CU operand stream runs at 8.4 TB/s when a tile row
// contributes 64 B per K step (BK = 16 fp32: half a cache line per row and instruction) and at 15 TB/s with 128-B rows; on the 128 x 64 tiles
// the teacher's wide GEMMs use, three co-resident workgroups ask for 36 KB per round of 3,072 matrix-pipe cycles = 88 % of the BK = 16 rate.
// Here a staging instruction reads 8 full 128-byte rows (thread = row tid>>3, chunk tid&7), the LDS image is two [row][16 k] halves with the
// swizzle of sgemm_nt16_kernel, and one barrier covers 8 MFMA k-steps.  Same products in the same order as the BK = 16 kernel: bit-identical.
// Full tiles only (M % BM == N % BN == 0, K range % 32 == 0).
template <int BM, int BN>
__global__ __launch_bounds__(256, BM * BN <= 128 * 64 ? 3 : 2) void sgemm_nt32_kernel(const GemmParams p) {
    constexpr int BK = 32;
    constexpr int TM = BM / 32, TN = BN / 32;
    constexpr int NA = BM / 32, NB = BN / 32;                            // staging passes of 32 rows
    static_assert(NA == 4 && (NB == 2 || NB == 4), "128 x 128 and 128 x 64");
    __shared__ __attribute__((aligned(16))) float As[2][2 * BM * 16];
    __shared__ __attribute__((aligned(16))) float Bs[2][2 * BN * 16];

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

    const int srow = tid >> 3, sch = tid & 7;
    const float* ga = p.A + (size_t)(m0 + srow) * p.lda + kbeg + sch * 4;
    // B rows staged permuted (see epilogue_rows): LDS row srow + 32 * pass holds global row TN * (srow & 15) + (srow >> 4) + {0, 2, 64, 66}[pass] (TN = 4)
    // or + 32 * pass (TN = 2)
    const float* gb = p.B + (size_t)(n0 + TN * (srow & 15) + (srow >> 4)) * p.ldb + kbeg + sch * 4;
    const size_t pa = (size_t)32 * p.lda;
    const size_t pb1 = (size_t)(TN == 4 ? 2 : 32) * p.ldb, pb2 = (size_t)64 * p.ldb, pb3 = (size_t)66 * p.ldb;
    const int s_off_a = (sch >> 2) * (BM * 16) + srow * 16 + 4 * ((sch & 3) ^ ((4 - ((srow >> 2) & 3)) & 3));
    const int s_off_b = (sch >> 2) * (BN * 16) + srow * 16 + 4 * ((sch & 3) ^ ((4 - ((srow >> 2) & 3)) & 3));
    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    ra0 = ra1 = ra2 = ra3 = rb0 = rb1 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_g = [&](int t) {
        ra0 = *reinterpret_cast<const float4*>(ga + t * BK);
        ra1 = *reinterpret_cast<const float4*>(ga + pa + t * BK);
        ra2 = *reinterpret_cast<const float4*>(ga + 2 * pa + t * BK);
        ra3 = *reinterpret_cast<const float4*>(ga + 3 * pa + t * BK);
        rb0 = *reinterpret_cast<const float4*>(gb + t * BK);
        rb1 = *reinterpret_cast<const float4*>(gb + pb1 + t * BK);
        if constexpr (NB > 2) {
            rb2 = *reinterpret_cast<const float4*>(gb + pb2 + t * BK);
            rb3 = *reinterpret_cast<const float4*>(gb + pb3 + t * BK);
        }
    };
    auto store_lds = [&](int buf) {
        *reinterpret_cast<float4*>(&As[buf][s_off_a]) = ra0;
        *reinterpret_cast<float4*>(&As[buf][s_off_a + 512]) = ra1;
        *reinterpret_cast<float4*>(&As[buf][s_off_a + 1024]) = ra2;
        *reinterpret_cast<float4*>(&As[buf][s_off_a + 1536]) = ra3;
        *reinterpret_cast<float4*>(&Bs[buf][s_off_b]) = rb0;
        *reinterpret_cast<float4*>(&Bs[buf][s_off_b + 512]) = rb1;
        if constexpr (NB > 2) {
            *reinterpret_cast<float4*>(&Bs[buf][s_off_b + 1024]) = rb2;
            *reinterpret_cast<float4*>(&Bs[buf][s_off_b + 1536]) = rb3;
        }
    };
    if (ntiles > 0) { load_g(0); store_lds(0); __syncthreads(); }
    const int kl = lane >> 4, ml = lane & 15;
    const int hsw = (4 - ((ml >> 2) & 3)) & 3;
    const int a_off = (wm * (BM / 2) + ml) * 16 + 4 * (kl ^ hsw);
    const int b_off = (wn * (BN / 2) + ml) * 16 + 4 * (kl ^ hsw);
    auto compute = [&](int buf) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            float4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(&As[buf][half * (BM * 16) + a_off + i * 256]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(&Bs[buf][half * (BN * 16) + b_off + j * 256]);
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
        }
    };
    for (int t = 0; t + 1 < ntiles; ++t) {
        load_g(t + 1);
        compute(t & 1);
        store_lds((t & 1) ^ 1);
        __syncthreads();
    }
    if (ntiles > 0) compute((ntiles - 1) & 1);

    const int wu = __builtin_amdgcn_readfirstlane(wave);
    epilogue_rows<-1, TM, TN, false, false, true>(p, acc, m0 + (wu >> 1) * (BM / 2), n0 + (wu & 1) * (BN / 2), ml, kl);
}
void launch_sgemm_nt32(const GemmParams& p, int tile, dim3 grid, hipStream_t s) {
    if (tile == 0) hipLaunchKernelGGL((sgemm_nt32_kernel<128, 128>), grid, dim3(256), 0, s, p);
    else           hipLaunchKernelGGL((sgemm_nt32_kernel<128, 64>), grid, dim3(256), 0, s, p);
}

// tile: 0 = 128x128, 1 = 64x128, 2 = 64x64, 3 = 128x64 (1..3 NN only: A K-contiguous; 2, 3: 4 x 1 waves).  false: no such kernel.
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

bool launch_sgemm_q16_fx(const GemmParams& p, int a_kmajor, int fx_mask, dim3 grid, hipStream_t s) {
#define QL(AK, FB, FA, FE) \
    if (p.epi.act == ACT_EPI_NONE) hipLaunchKernelGGL((sgemm_q16_kernel<128, 128, AK, false, false, FB, FA, FE, ACT_EPI_NONE>), grid, dim3(256), 0, s, p); \
    else                           hipLaunchKernelGGL((sgemm_q16_kernel<128, 128, AK, false, false, FB, FA, FE>), grid, dim3(256), 0, s, p); \
    return true
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

template <int BM, int BN>
static void launch16(const GemmParams& p, int ak, int bk, dim3 grid, hipStream_t s) {
    if (p.M % BM != 0) {                                 // M tail: A must be K-major (rows = tokens)
        if (bk) hipLaunchKernelGGL((sgemm16_kernel<BM, BN, true, true, true>), grid, dim3(256), 0, s, p);
        else    hipLaunchKernelGGL((sgemm16_kernel<BM, BN, true, false, true>), grid, dim3(256), 0, s, p);
        return;
    }
    if (ak && bk)        hipLaunchKernelGGL((sgemm16_kernel<BM, BN, true, true>), grid, dim3(256), 0, s, p);
    else if (ak && !bk)  hipLaunchKernelGGL((sgemm16_kernel<BM, BN, true, false>), grid, dim3(256), 0, s, p);
    else if (!ak && !bk) hipLaunchKernelGGL((sgemm16_kernel<BM, BN, false, false>), grid, dim3(256), 0, s, p);
    else                 hipLaunchKernelGGL((sgemm16_kernel<BM, BN, false, true>), grid, dim3(256), 0, s, p);
}

void launch_sgemm_nt16(const GemmParams& p, int tile, dim3 grid, hipStream_t s) {
    if (tile == 3) { hipLaunchKernelGGL((sgemm_nt16_kernel<128, 128, false, 0, true>), grid, dim3(256), 0, s, p); return; }   // pipelined loop: full tiles only
    if (tile == 4) { hipLaunchKernelGGL((sgemm_nt16_kernel<128, 64, false, 0, true>), grid, dim3(256), 0, s, p); return; }
    const int bm = tile == 2 ? 64 : 128;
    if (p.M % bm != 0) {
        if (tile == 0)      hipLaunchKernelGGL((sgemm_nt16_kernel<128, 128, true>), grid, dim3(256), 0, s, p);
        else if (tile == 1) hipLaunchKernelGGL((sgemm_nt16_kernel<128, 64, true>), grid, dim3(256), 0, s, p);
        else                hipLaunchKernelGGL((sgemm_nt16_kernel<64, 64, true>), grid, dim3(256), 0, s, p);
        return;
    }
    static const int spec = [] { const char* e = getenv("ACT_GEMM_EPI_SPEC"); return e ? atoi(e) : 1; }();
    if (!spec) {
        if (tile == 0)      hipLaunchKernelGGL((sgemm_nt16_kernel<128, 128>), grid, dim3(256), 0, s, p);
        else if (tile == 1) hipLaunchKernelGGL((sgemm_nt16_kernel<128, 64>), grid, dim3(256), 0, s, p);
        else                hipLaunchKernelGGL((sgemm_nt16_kernel<64, 64>), grid, dim3(256), 0, s, p);
        return;
    }
#define NT16_ACT(BM_, BN_) \
    switch (p.epi.act) { \
        case ACT_EPI_NONE: hipLaunchKernelGGL((sgemm_nt16_kernel<BM_, BN_, false, 0, false, ACT_EPI_NONE>), grid, dim3(256), 0, s, p); break; \
        case ACT_EPI_GELU: hipLaunchKernelGGL((sgemm_nt16_kernel<BM_, BN_, false, 0, false, ACT_EPI_GELU>), grid, dim3(256), 0, s, p); break; \
        case ACT_EPI_RELU: hipLaunchKernelGGL((sgemm_nt16_kernel<BM_, BN_, false, 0, false, ACT_EPI_RELU>), grid, dim3(256), 0, s, p); break; \
        default:           hipLaunchKernelGGL((sgemm_nt16_kernel<BM_, BN_>), grid, dim3(256), 0, s, p); break; \
    }
    if (tile == 0)      { NT16_ACT(128, 128) }
    else if (tile == 1) { NT16_ACT(128, 64) }
    else                { NT16_ACT(64, 64) }
#undef NT16_ACT
}

void launch_sgemm16(const GemmParams& p, int tile, int a_kmajor, int b_kmajor, dim3 grid, hipStream_t s) {
    if (tile == 0)      launch16<128, 128>(p, a_kmajor, b_kmajor, grid, s);
    else if (tile == 1) launch16<128, 64>(p, a_kmajor, b_kmajor, grid, s);
    else                launch16<64, 64>(p, a_kmajor, b_kmajor, grid, s);
}
