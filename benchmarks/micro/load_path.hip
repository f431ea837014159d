// load_path.hip -- how fast can a GEMM-shaped workgroup grid pull K-contiguous row panels from L2 into registers (the first half of the LDS staging) on gfx950, as a function of the
// contiguous bytes per row per K-step (64 / 128 / 256 B) and of the loads in flight per thread?  No MFMAs: staging loop only.
// Access pattern = the A/B operand staging of an NT GEMM with 128x128 tiles over [M][K] / [N][K] row-major matrices of ELEM-byte elements.
// build: hipcc --offload-arch=gfx950 -O3 -o load_path load_path.hip ; run: ./load_path
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define HIPCHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// ROWB = contiguous bytes per row per K-step; a 128-row panel is 128 x ROWB bytes per operand per step; both operands are loaded.
// Every thread issues NLD 16-byte loads per step (NLD = 2 * 128 * ROWB / 16 / 256) and stores them to LDS; DEPTH steps are kept in flight.
template <int ROWB, int DEPTH>
__global__ __launch_bounds__(256, 2) void stage_kernel(const char* __restrict__ A, const char* __restrict__ B, int M, int N, long long row_bytes,
                                                       float* __restrict__ sink) {
    constexpr int LPR = ROWB / 16;                   // lanes per row
    constexpr int RPP = 256 / LPR;                   // rows covered by one pass of the workgroup
    constexpr int NP = 128 / RPP;                    // passes per operand
    const int tid = threadIdx.x;
    const int tiles_n = N / 128;
    const int m0 = (blockIdx.x / tiles_n) * 128, n0 = (blockIdx.x % tiles_n) * 128;
    const int r = tid / LPR, c = tid % LPR;
    const char* ga = A + (size_t)(m0 + r) * row_bytes + c * 16;
    const char* gb = B + (size_t)(n0 + r) * row_bytes + c * 16;
    const int steps = (int)(row_bytes / ROWB);
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint4 ra[DEPTH][NP], rb[DEPTH][NP];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            ra[d][p] = *reinterpret_cast<const uint4*>(ga + (size_t)p * RPP * row_bytes + (size_t)d * ROWB);
            rb[d][p] = *reinterpret_cast<const uint4*>(gb + (size_t)p * RPP * row_bytes + (size_t)d * ROWB);
        }
    for (int t = 0; t < steps; t += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                acc.x ^= ra[d][p].x ^ rb[d][p].y; acc.y += ra[d][p].w + rb[d][p].z;
            }
            const int tn = t + DEPTH + d;
            if (tn < steps) {
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    ra[d][p] = *reinterpret_cast<const uint4*>(ga + (size_t)p * RPP * row_bytes + (size_t)tn * ROWB);
                    rb[d][p] = *reinterpret_cast<const uint4*>(gb + (size_t)p * RPP * row_bytes + (size_t)tn * ROWB);
                }
            }
        }
    }
    if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[blockIdx.x * 256 + tid] = 1.f;     // never true: keeps the loads alive
}

template <int ROWB, int DEPTH>
static void run(const char* dA, const char* dB, int M, int N, long long row_bytes, float* sink) {
    dim3 grid((M / 128) * (N / 128));
    hipLaunchKernelGGL((stage_kernel<ROWB, DEPTH>), grid, dim3(256), 0, 0, dA, dB, M, N, row_bytes, sink);
    HIPCHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    HIPCHECK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((stage_kernel<ROWB, DEPTH>), grid, dim3(256), 0, 0, dA, dB, M, N, row_bytes, sink);
    HIPCHECK(hipEventRecord(e1)); HIPCHECK(hipEventSynchronize(e1));
    float ms; HIPCHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    const double bytes = (double)grid.x * 2.0 * 128.0 * (double)row_bytes;
    printf("  %3d B per row per step, %d steps in flight (%2d x 16 B loads per thread): %.3f ms  %.2f TB/s L2->CU\n", ROWB, DEPTH,
           2 * (128 * ROWB / 16 / 256) * DEPTH, ms, bytes / (ms * 1e-3) / 1e12);
}

int main() {
    const int M = 16384, N = 3072;
    for (long long row_bytes : {3072LL, 4608LL, 12288LL}) {           // K = 768 fp32 | K = 768 as three bf16 planes back to back | K = 3072 fp32
        char *dA, *dB; float* sink;
        HIPCHECK(hipMalloc(&dA, (size_t)M * row_bytes)); HIPCHECK(hipMalloc(&dB, (size_t)N * row_bytes)); HIPCHECK(hipMalloc(&sink, 1 << 24));
        HIPCHECK(hipMemset(dA, 1, (size_t)M * row_bytes)); HIPCHECK(hipMemset(dB, 2, (size_t)N * row_bytes));
        printf("M=%d N=%d row = %lld bytes  (tile traffic %.2f GB per launch)\n", M, N, row_bytes, (double)(M / 128) * (N / 128) * 256.0 * row_bytes / 1e9);
        run<64, 1>(dA, dB, M, N, row_bytes, sink);
        run<64, 2>(dA, dB, M, N, row_bytes, sink);
        run<64, 4>(dA, dB, M, N, row_bytes, sink);
        run<128, 1>(dA, dB, M, N, row_bytes, sink);
        run<128, 2>(dA, dB, M, N, row_bytes, sink);
        run<256, 1>(dA, dB, M, N, row_bytes, sink);
        run<256, 2>(dA, dB, M, N, row_bytes, sink);
        HIPCHECK(hipFree(dA)); HIPCHECK(hipFree(dB)); HIPCHECK(hipFree(sink));
    }
    return 0;
}
