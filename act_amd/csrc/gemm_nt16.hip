// gemm_nt16.hip -- launchers (= instantiations) of the plain NT b128 kernels, and the 32-deep-K variant
#include "gemm_nt16_kernel.h"

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

