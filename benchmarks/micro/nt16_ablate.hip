// nt16_ablate.hip -- where does the NT b128 GEMM kernel (act_amd/csrc/gemm16.hip, 128x64 tile) lose time?  Same main loop, pieces
// removed by template flags.  hipcc --offload-arch=gfx950 -O3 -w nt16_ablate.hip -o nt16_ablate   (dev microbenchmark)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
enum { NO_GLOBAL = 1, NO_LDS_STORE = 2, NO_BARRIER = 4, NO_LDS_READ = 8 };

template <int ABL>
__global__ __launch_bounds__(256, 3) void k(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    constexpr int BM = 128, BN = 64, BK = 16, TM = 4, TN = 2;
    __shared__ __attribute__((aligned(16))) float As[2][BM * BK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * BK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int tiles_n = N / BN, tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN, ntiles = K / BK;
    f32x4 acc[TM][TN];
    for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int srow = tid >> 2, sch = tid & 3;
    const float* ga = A + (size_t)(m0 + srow) * K + sch * 4;
    const float* gb = B + (size_t)(n0 + srow) * K + sch * 4;
    const int s_off = srow * 16 + 4 * (sch ^ ((4 - ((srow >> 2) & 3)) & 3));
    float4 ra0 = make_float4(1.f, 1.f, 1.f, 1.f), ra1 = ra0, rb0 = ra0;
    auto load_g = [&](int t) {
        if (ABL & NO_GLOBAL) return;
        ra0 = *reinterpret_cast<const float4*>(ga + t * BK);
        ra1 = *reinterpret_cast<const float4*>(ga + (size_t)64 * K + t * BK);
        rb0 = *reinterpret_cast<const float4*>(gb + t * BK);
    };
    auto store_lds = [&](int buf) {
        if (ABL & NO_LDS_STORE) return;
        *reinterpret_cast<float4*>(&As[buf][s_off]) = ra0;
        *reinterpret_cast<float4*>(&As[buf][s_off + 1024]) = ra1;
        *reinterpret_cast<float4*>(&Bs[buf][s_off]) = rb0;
    };
    for (int i = tid; i < BM * BK; i += 256) { As[0][i] = 1.f; As[1][i] = 1.f; }
    for (int i = tid; i < BN * BK; i += 256) { Bs[0][i] = 1.f; Bs[1][i] = 1.f; }
    load_g(0); store_lds(0); __syncthreads();
    const int kl = lane >> 4, ml = lane & 15, hsw = (4 - ((ml >> 2) & 3)) & 3;
    const int a_off = (wm * 64 + ml) * 16 + 4 * (kl ^ hsw), b_off = (wn * 32 + ml) * 16 + 4 * (kl ^ hsw);
    float4 cf = make_float4(1.f + lane * 1e-6f, 1.f, 1.f, 1.f);
    auto compute = [&](int buf) {
        float4 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = (ABL & NO_LDS_READ) ? cf : *reinterpret_cast<const float4*>(&As[buf][a_off + i * 256]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = (ABL & NO_LDS_READ) ? cf : *reinterpret_cast<const float4*>(&Bs[buf][b_off + j * 256]);
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
    for (int t = 0; t + 1 < ntiles; ++t) {
        load_g(t + 1);
        compute(t & 1);
        store_lds((t & 1) ^ 1);
        if (!(ABL & NO_BARRIER)) __syncthreads();
    }
    compute((ntiles - 1) & 1);
    for (int i = 0; i < TM; ++i)
        for (int j = 0; j < TN; ++j)
            for (int r = 0; r < 4; ++r)
                C[(size_t)(m0 + wm * 64 + i * 16 + kl * 4 + r) * N + n0 + wn * 32 + j * 16 + ml] = acc[i][j][r];
}

template <int ABL>
void run(const char* name, const float* A, const float* B, float* C, int M, int N, int K) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid((M / 128) * (N / 64));
    hipLaunchKernelGGL(k<ABL>, grid, dim3(256), 0, 0, A, B, C, M, N, K); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<ABL>, grid, dim3(256), 0, 0, A, B, C, M, N, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-58s %7.1f us  %6.1f TFLOP/s\n", name, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
}

int main() {
    const int shapes[3][3] = {{8192, 3072, 768}, {8192, 768, 3072}, {16384, 3072, 768}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        float *A, *B, *C; hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
        hipMemset(A, 0, (size_t)M * K * 4); hipMemset(B, 0, (size_t)N * K * 4);
        printf("--- %d x %d x %d\n", M, N, K);
        run<0>("full kernel", A, B, C, M, N, K);
        run<NO_GLOBAL>("no global loads", A, B, C, M, N, K);
        run<NO_GLOBAL | NO_LDS_STORE>("no global loads, no LDS stores", A, B, C, M, N, K);
        run<NO_GLOBAL | NO_LDS_STORE | NO_BARRIER>("no global loads, no LDS stores, no barrier", A, B, C, M, N, K);
        run<NO_GLOBAL | NO_LDS_STORE | NO_BARRIER | NO_LDS_READ>("MFMA only (no memory operations in the loop)", A, B, C, M, N, K);
        run<NO_LDS_READ>("global loads + LDS stores + barrier, no LDS fragment reads", A, B, C, M, N, K);
        run<NO_BARRIER>("everything but the barrier (racy, timing only)", A, B, C, M, N, K);
        hipFree(A); hipFree(B); hipFree(C);
    }
    return 0;
}
