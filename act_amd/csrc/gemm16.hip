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

void launch_sgemm16(const GemmParams& p, int tile, int a_kmajor, int b_kmajor, dim3 grid, hipStream_t s) {
    if (tile == 0)      launch16<128, 128>(p, a_kmajor, b_kmajor, grid, s);
    else if (tile == 1) launch16<128, 64>(p, a_kmajor, b_kmajor, grid, s);
    else                launch16<64, 64>(p, a_kmajor, b_kmajor, grid, s);
}
