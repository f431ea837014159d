// nt16_dma.hip -- NT b128 GEMM main loop with direct-to-LDS loads (global_load_lds_dwordx4), three LDS stages and counted vmcnt
// waits, against the register-staged loop of act_amd/csrc/gemm16.hip.  Standalone correctness + timing harness (dev).
// hipcc --offload-arch=gfx950 -O3 -w nt16_dma.hip -o nt16_dma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int BM, int BN, bool DMA>
__global__ __launch_bounds__(256, (BM * BN > 128 * 128 ? 2 : 3)) void k(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
    constexpr int BK = 16, TM = BM / 32, TN = BN / 32, NA = BM / 64, NB = BN / 64, NST = DMA ? 3 : 2;
    __shared__ __attribute__((aligned(1024))) float As[NST][BM * BK];
    __shared__ __attribute__((aligned(1024))) float Bs[NST][BN * BK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int tiles_n = N / BN, tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN, ntiles = K / BK;
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int srow = tid >> 2, sch = tid & 3, h = (4 - ((srow >> 2) & 3)) & 3;
    const int kl = lane >> 4, ml = lane & 15, hsw = (4 - ((ml >> 2) & 3)) & 3;
    const int a_off = (wm * (BM / 2) + ml) * 16 + 4 * (kl ^ hsw), b_off = (wn * (BN / 2) + ml) * 16 + 4 * (kl ^ hsw);
    auto compute = [&](const float* as, const float* bs) {
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
    };
    if constexpr (!DMA) {
        const float* ga = A + (size_t)(m0 + srow) * K + sch * 4;
        const float* gb = B + (size_t)(n0 + srow) * K + sch * 4;
        const int s_off = srow * 16 + 4 * (sch ^ h);
        float4 ra0, ra1, ra2, ra3, rb0, rb1;
        auto load_g = [&](int t) {
            ra0 = *reinterpret_cast<const float4*>(ga + t * BK);
            if constexpr (NA > 1) ra1 = *reinterpret_cast<const float4*>(ga + (size_t)64 * K + t * BK);
            if constexpr (NA > 2) { ra2 = *reinterpret_cast<const float4*>(ga + (size_t)128 * K + t * BK); ra3 = *reinterpret_cast<const float4*>(ga + (size_t)192 * K + t * BK); }
            rb0 = *reinterpret_cast<const float4*>(gb + t * BK);
            if constexpr (NB > 1) rb1 = *reinterpret_cast<const float4*>(gb + (size_t)64 * K + t * BK);
        };
        auto store_lds = [&](int buf) {
            *reinterpret_cast<float4*>(&As[buf][s_off]) = ra0;
            if constexpr (NA > 1) *reinterpret_cast<float4*>(&As[buf][s_off + 1024]) = ra1;
            if constexpr (NA > 2) { *reinterpret_cast<float4*>(&As[buf][s_off + 2048]) = ra2; *reinterpret_cast<float4*>(&As[buf][s_off + 3072]) = ra3; }
            *reinterpret_cast<float4*>(&Bs[buf][s_off]) = rb0;
            if constexpr (NB > 1) *reinterpret_cast<float4*>(&Bs[buf][s_off + 1024]) = rb1;
        };
        load_g(0); store_lds(0); __syncthreads();
        for (int t = 0; t + 1 < ntiles; ++t) { load_g(t + 1); compute(As[t & 1], Bs[t & 1]); store_lds((t & 1) ^ 1); __syncthreads(); }
        compute(As[(ntiles - 1) & 1], Bs[(ntiles - 1) & 1]);
    } else {
        // lane l of wave w lands at LDS byte (w*16 rows)*64 + 16*l = row (w*16 + l>>2), chunk slot l&3, which must hold the logical
        // chunk (l&3)^h(row): the swizzle moves to the SOURCE address, the destination stays linear
        const float* ga = A + (size_t)(m0 + srow) * K + 4 * (sch ^ h);
        const float* gb = B + (size_t)(n0 + srow) * K + 4 * (sch ^ h);
        const unsigned a_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)&As[0][wave * 256]);
        const unsigned b_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)&Bs[0][wave * 256]);
        constexpr int L = NA + NB;                          // loads per thread per tile
        auto issue = [&](int t, int st) {
            glds16(ga + t * BK, a_lds + st * (BM * BK * 4));
            if constexpr (NA > 1) glds16(ga + (size_t)64 * K + t * BK, a_lds + st * (BM * BK * 4) + 4096);
            glds16(gb + t * BK, b_lds + st * (BN * BK * 4));
            if constexpr (NB > 1) glds16(gb + (size_t)64 * K + t * BK, b_lds + st * (BN * BK * 4) + 4096);
        };
        issue(0, 0);
        if (ntiles > 1) { issue(1, 1); wait_vmcnt<L>(); } else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        int st = 0;                                         // stage holding tile t
        for (int t = 0; t < ntiles; ++t) {
            const int st2 = st == 0 ? 2 : st - 1;           // (t+2) % 3 == (t-1) % 3
            if (t + 2 < ntiles) issue(t + 2, st2);
            compute(As[st], Bs[st]);
            if (t + 2 < ntiles) wait_vmcnt<L>(); else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            st = st == 2 ? 0 : st + 1;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                C[(size_t)(m0 + wm * (BM / 2) + i * 16 + kl * 4 + r) * N + n0 + wn * (BN / 2) + j * 16 + ml] = acc[i][j][r];
}

template <int BM, int BN, bool DMA>
float run(const float* A, const float* B, float* C, int M, int N, int K) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    dim3 grid((M / BM) * (N / BN));
    hipLaunchKernelGGL((k<BM, BN, DMA>), grid, dim3(256), 0, 0, A, B, C, M, N, K); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<BM, BN, DMA>), grid, dim3(256), 0, 0, A, B, C, M, N, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 20;
}

int main(int argc, char** argv) {
    const bool zero = argc > 1;
    const int shapes[5][3] = {{8192, 3072, 768}, {8192, 768, 3072}, {8192, 2304, 768}, {8192, 8192, 2304}, {1024, 512, 64}};
    for (auto& sh : shapes) {
        const int M = sh[0], N = sh[1], K = sh[2];
        float *A, *B, *C0, *C1;
        hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C0, (size_t)M * N * 4); hipMalloc(&C1, (size_t)M * N * 4);
        float* hA = (float*)malloc((size_t)M * K * 4); float* hB = (float*)malloc((size_t)N * K * 4);
        srand(1); for (size_t i = 0; i < (size_t)M * K; ++i) hA[i] = zero ? 0.f : (rand() % 2001 - 1000) * 1e-3f;
        for (size_t i = 0; i < (size_t)N * K; ++i) hB[i] = zero ? 0.f : (rand() % 2001 - 1000) * 1e-3f;
        hipMemcpy(A, hA, (size_t)M * K * 4, hipMemcpyHostToDevice); hipMemcpy(B, hB, (size_t)N * K * 4, hipMemcpyHostToDevice);
        const float t0 = run<128, 64, false>(A, B, C0, M, N, K), t1 = run<128, 64, true>(A, B, C1, M, N, K);
        const float t2 = run<128, 128, false>(A, B, C0, M, N, K), t3 = run<128, 128, true>(A, B, C1, M, N, K);
        const float t4 = (M % 256 == 0) ? run<256, 128, false>(A, B, C1, M, N, K) : 1e9f;
        {   // check 256x128 against the 128x64 staged result
            run<128, 64, false>(A, B, C0, M, N, K);
            float* g0 = (float*)malloc((size_t)M * N * 4); float* g1 = (float*)malloc((size_t)M * N * 4);
            hipMemcpy(g0, C0, (size_t)M * N * 4, hipMemcpyDeviceToHost); hipMemcpy(g1, C1, (size_t)M * N * 4, hipMemcpyDeviceToHost);
            double md2 = 0; for (size_t i = 0; i < (size_t)M * N; ++i) md2 = fmax(md2, fabs((double)g0[i] - g1[i]));
            printf("   256x128 staged: %7.1f us %6.1f TF  max|diff| %.2e\n", t4 * 1e3, 2.0 * M * N * K / t4 / 1e9, md2);
            free(g0); free(g1);
        }
        run<128, 64, false>(A, B, C0, M, N, K); run<128, 64, true>(A, B, C1, M, N, K);
        float* h0 = (float*)malloc((size_t)M * N * 4); float* h1 = (float*)malloc((size_t)M * N * 4);
        hipMemcpy(h0, C0, (size_t)M * N * 4, hipMemcpyDeviceToHost); hipMemcpy(h1, C1, (size_t)M * N * 4, hipMemcpyDeviceToHost);
        double md = 0; for (size_t i = 0; i < (size_t)M * N; ++i) md = fmax(md, fabs((double)h0[i] - h1[i]));
        const double fl = 2.0 * M * N * K;
        printf("%5dx%5dx%5d  128x64: staged %7.1f us %6.1f TF | dma %7.1f us %6.1f TF   128x128: staged %6.1f TF | dma %6.1f TF   max|diff| %.2e\n",
               M, N, K, t0 * 1e3, fl / t0 / 1e9, t1 * 1e3, fl / t1 / 1e9, fl / t2 / 1e9, fl / t3 / 1e9, md);
        free(hA); free(hB); free(h0); free(h1); hipFree(A); hipFree(B); hipFree(C0); hipFree(C1);
    }
    return 0;
}
