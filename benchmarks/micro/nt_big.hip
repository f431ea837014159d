// nt_big.hip -- does a bigger WORKGROUP tile (less staging traffic per flop) with the same 64x64 per-wave tile pay?  NT b128 main loop of
// act_amd/csrc/gemm16.hip with a WM x WN grid of waves: 128x128 (4 waves), 256x128 (8 waves), 256x256 (16 waves); register-staged, BK = 16 | 32,
// plain / grouped + XCD-banded tile order.  RESULT (round 3): NO.  In this harness the 8- and 16-wave tiles reach 121-142 TFLOP/s and the 4-wave tile
// 89-101, but that baseline is an artefact (its staging registers are arrays indexed inside lambdas, which hipcc leaves in scratch); the same kernels built
// into the library next to the production 4-wave kernel (named scalar staging registers, tile ids 24-26 of an experiment that was not kept) gave
// 8192x8192x2304: 139.0 (128x128 / 4 waves) vs 137-139; 16384x3072x768: 132 vs 128-130; 8192x3072x768: 127 vs 125 (256x128) / 96 (256x256: 384 tiles on
// 256 CUs); torch.mm (hipBLASLt) 148.6 / 139.2 / 136.9 on the same operands.  hipBLASLt's kernels for these shapes (rocprofv3 of torch.mm,
// benchmarks/hipblaslt_names.py): MT128x128x64 MI16x16x1 stream-K (SK3), 256 persistent workgroups = ONE 4-wave workgroup per CU, 64-deep K tiles in a
// single 66 KB LDS buffer, global prefetch distance 2 -- latency hidden by a hand-scheduled instruction stream, not by occupancy.
// hipcc --offload-arch=gfx950 -O3 -w nt_big.hip -o nt_big ; ./nt_big [zero]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int xcd_remap(int wg, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, loc = wg >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}
__device__ __forceinline__ void tile_coords(int tiles_m, int tiles_n, int group_m, int wg, int& tile_m, int& tile_n) {
    const int per_group = group_m * tiles_n;
    const int group = wg / per_group, first_m = group * group_m;
    const int gsz = min(tiles_m - first_m, group_m);
    const int in_group = wg - group * per_group;
    tile_m = first_m + in_group % gsz;
    tile_n = in_group / gsz;
}
template <int WM, int WN, int OCC, int BK, int RASTER>
__global__ __launch_bounds__(64 * WM * WN, OCC) void k(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    constexpr int BM = 64 * WM, BN = 64 * WN, NT = 64 * WM * WN, TM = 4, TN = 4, KC = BK / 4, RP = NT / KC;
    constexpr int NA = BM / RP, NB = BN / RP;                  // float4 per thread per operand tile (passes of RP rows)
    static_assert(NA >= 1 && NB >= 1 && RP % 16 == 0, "");
    __shared__ __attribute__((aligned(16))) float As[2][BM * BK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * BK];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / WN, wn = wave % WN;
    const int tiles_n = N / BN, tiles_m = M / BM; int tile_m, tile_n;
    if (RASTER) tile_coords(tiles_m, tiles_n, RASTER, xcd_remap(blockIdx.x, tiles_m * tiles_n), tile_m, tile_n);
    else { tile_m = blockIdx.x / tiles_n; tile_n = blockIdx.x % tiles_n; }
    const int m0 = tile_m * BM, n0 = tile_n * BN, ntiles = K / BK;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int srow = tid / KC, sch = tid % KC, h = (4 - ((srow >> 2) & 3)) & 3;
    const int kl = lane >> 4, ml = lane & 15, hsw = (4 - ((ml >> 2) & 3)) & 3;
    const int a_off = (wm * 64 + ml) * 16 + 4 * (kl ^ hsw), b_off = (wn * 64 + ml) * 16 + 4 * (kl ^ hsw);
    const float* ga = A + (size_t)(m0 + srow) * K + sch * 4;
    const float* gb = B + (size_t)(n0 + srow) * K + sch * 4;
    const int s_offa = (sch >> 2) * (BM * 16) + srow * 16 + 4 * ((sch & 3) ^ h), s_offb = (sch >> 2) * (BN * 16) + srow * 16 + 4 * ((sch & 3) ^ h);
    float4 ra[NA], rb[NB];
    auto load_g = [&](int t) {
#pragma unroll
        for (int i = 0; i < NA; ++i) ra[i] = *reinterpret_cast<const float4*>(ga + (size_t)(i * RP) * K + t * BK);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const float4*>(gb + (size_t)(i * RP) * K + t * BK);
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) *reinterpret_cast<float4*>(&As[buf][s_offa + i * RP * 16]) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<float4*>(&Bs[buf][s_offb + i * RP * 16]) = rb[i];
    };
    auto compute = [&](const float* as0, const float* bs0) {
#pragma unroll
      for (int half = 0; half < BK / 16; ++half) {
        const float* as = as0 + half * (BM * 16); const float* bs = bs0 + half * (BN * 16);
        float4 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const float4*>(as + a_off + i * 256);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const float4*>(bs + b_off + j * 256);
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
    load_g(0); store_lds(0); __syncthreads();
    for (int t = 0; t + 1 < ntiles; ++t) { load_g(t + 1); compute(As[t & 1], Bs[t & 1]); store_lds((t & 1) ^ 1); __syncthreads(); }
    compute(As[(ntiles - 1) & 1], Bs[(ntiles - 1) & 1]);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                C[(size_t)(m0 + wm * 64 + i * 16 + kl * 4 + r) * N + n0 + wn * 64 + j * 16 + ml] = acc[i][j][r];
}

template <int WM, int WN, int OCC, int BK = 16, int RASTER = 0>
float run(const float* A, const float* B, float* C, int M, int N, int K) {
    if (M % (64 * WM) || N % (64 * WN)) return 1e9f;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid((M / (64 * WM)) * (N / (64 * WN)));
    hipLaunchKernelGGL((k<WM, WN, OCC, BK, RASTER>), grid, dim3(64 * WM * WN), 0, 0, A, B, C, M, N, K); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<WM, WN, OCC, BK, RASTER>), grid, dim3(64 * WM * WN), 0, 0, A, B, C, M, N, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 10;
}

int main(int argc, char** argv) {
    const bool zero = argc > 1;
    const int shapes[6][3] = {{16384, 4096, 768}, {16384, 3072, 768}, {8192, 3072, 768}, {8192, 2304, 768}, {8192, 768, 3072}, {8192, 8192, 2304}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        float *A, *B, *C0, *C1;
        hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C0, (size_t)M * N * 4); hipMalloc(&C1, (size_t)M * N * 4);
        float* hA = (float*)malloc((size_t)M * K * 4); float* hB = (float*)malloc((size_t)N * K * 4);
        srand(1); for (size_t i = 0; i < (size_t)M * K; ++i) hA[i] = zero ? 0.f : (rand() % 2001 - 1000) * 1e-3f;
        for (size_t i = 0; i < (size_t)N * K; ++i) hB[i] = zero ? 0.f : (rand() % 2001 - 1000) * 1e-3f;
        hipMemcpy(A, hA, (size_t)M * K * 4, hipMemcpyHostToDevice); hipMemcpy(B, hB, (size_t)N * K * 4, hipMemcpyHostToDevice);
        const double fl = 2.0 * M * N * K / 1e9;
        const float t0 = run<2, 2, 3>(A, B, C0, M, N, K);
        float* h0 = (float*)malloc((size_t)M * N * 4); float* h1 = (float*)malloc((size_t)M * N * 4);
        hipMemcpy(h0, C0, (size_t)M * N * 4, hipMemcpyDeviceToHost);
        auto chk = [&]() { hipMemcpy(h1, C1, (size_t)M * N * 4, hipMemcpyDeviceToHost); double md = 0; for (size_t i = 0; i < (size_t)M * N; ++i) md = fmax(md, fabs((double)h0[i] - h1[i])); return md; };
        const float r0 = run<2, 2, 3, 16, 8>(A, B, C1, M, N, K); const double d0 = chk();
        const float r1 = run<4, 2, 2, 16, 8>(A, B, C1, M, N, K); const double d1 = chk();
        const float r2 = run<4, 4, 1, 16, 8>(A, B, C1, M, N, K); const double d2 = chk();
        const float r3 = run<4, 2, 2, 32, 8>(A, B, C1, M, N, K); const double d3 = chk();
        const float r4 = run<4, 4, 1, 32, 8>(A, B, C1, M, N, K); const double d4 = chk();
        const float r5 = run<4, 4, 1, 16, 4>(A, B, C1, M, N, K);
        const float r6 = run<4, 2, 2, 16, 4>(A, B, C1, M, N, K);
        const float r7 = run<4, 4, 1, 16, 0>(A, B, C1, M, N, K);
        printf("%5dx%5dx%5d %s  128x128 plain %6.1f raster8 %6.1f | 256x128/8w r8 %6.1f bk32 %6.1f r4 %6.1f | 256x256/16w r8 %6.1f bk32 %6.1f r4 %6.1f plain %6.1f TF   max|diff| %.1e %.1e %.1e %.1e %.1e\n",
               M, N, K, zero ? "zeros " : "random", fl / t0, fl / r0, fl / r1, fl / r3, fl / r6, fl / r2, fl / r4, fl / r5, fl / r7, d0, d1, d2, d3, d4);
        free(hA); free(hB); free(h0); free(h1); hipFree(A); hipFree(B); hipFree(C0); hipFree(C1);
    }
    return 0;
}
