// bf16x6_nt.hip -- feasibility probe: fp32 GEMM C[M,N] = A[M,K] . B[N,K]^T on the bf16 matrix cores by exact 3-way splitting.
//
// Every fp32 operand x is decomposed EXACTLY into three bf16 terms x = h + m + l (8 + 8 + 8 significand bits, residuals computed in
// fp32 without rounding error); the six cross products of weight >= 2^-16 (hh, hm, mh, hl, mm, lh) are accumulated in fp32 by
// v_mfma_f32_16x16x32_bf16.  The neglected terms are < 3 * 2^-24 |a b|: the size of one fp32 rounding of the product.  The bf16 MFMA
// rate is 16x the f32-input MFMA rate, so six products still leave 2.67x head-room over v_mfma_f32_16x16x4_f32.
//
// build: hipcc --offload-arch=gfx950 -O3 -o bf16x6_nt bf16x6_nt.hip ; run: ./bf16x6_nt [M N K]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned bf16_rne(float x) {           // round-to-nearest-even bf16 bits (finite inputs)
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}
// x -> (h, m, l) bf16 bit patterns with x == h + m + l exactly
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = bf16_rne(x);
    const float r1 = x - __uint_as_float(h << 16);
    m = bf16_rne(r1);
    const float r2 = r1 - __uint_as_float(m << 16);
    l = bf16_rne(r2);
}

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDS_ROW = 40;                 // bf16 per LDS row (32 + 8 pad: 80-byte stride -> conflict-free ds_read_b128)
constexpr int PLANE = BM * LDS_ROW;         // bf16 per split plane

template <int NPROD>
__global__ __launch_bounds__(256, 2) void gemm_nt_bf16split(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                            int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) unsigned short As[2][3 * PLANE];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][3 * PLANE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = N / BN;
    const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
    const int ntiles = K / BK;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging: 128 rows x 32 k floats = 1024 float4 per operand; thread v owns (row = v >> 3 (+32 per extra load), chunk = v & 7)
    const int srow = tid >> 3, sch = tid & 7;
    const float* ga = A + (size_t)(m0 + srow) * K + sch * 4;
    const float* gb = B + (size_t)(n0 + srow) * K + sch * 4;
    float4 ra[4], rb[4];
    auto load_g = [&](int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = *reinterpret_cast<const float4*>(ga + (size_t)(32 * i) * K + t * BK);
            rb[i] = *reinterpret_cast<const float4*>(gb + (size_t)(32 * i) * K + t * BK);
        }
    };
    auto store_one = [&](unsigned short* base, const float4& v, int row) {
        unsigned h0, m0_, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
        split3(v.x, h0, m0_, l0); split3(v.y, h1, m1, l1); split3(v.z, h2, m2, l2); split3(v.w, h3, m3, l3);
        const int off = row * LDS_ROW + sch * 4;
        *reinterpret_cast<uint2*>(base + off) = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
        *reinterpret_cast<uint2*>(base + PLANE + off) = make_uint2(m0_ | (m1 << 16), m2 | (m3 << 16));
        *reinterpret_cast<uint2*>(base + 2 * PLANE + off) = make_uint2(l0 | (l1 << 16), l2 | (l3 << 16));
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { store_one(As[buf], ra[i], srow + 32 * i); store_one(Bs[buf], rb[i], srow + 32 * i); }
    };

    load_g(0); store_lds(0); __syncthreads();
    const int r = lane & 15, g = lane >> 4;
    const int a_off = (wm * 64 + r) * LDS_ROW + g * 8, b_off = (wn * 64 + r) * LDS_ROW + g * 8;
    for (int t = 0; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) load_g(t + 1);
        bf16x8 af[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int s = 0; s < 3; ++s) af[i][s] = *reinterpret_cast<const bf16x8*>(&As[buf][s * PLANE + a_off + i * 16 * LDS_ROW]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bf16x8 bf[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) bf[s] = *reinterpret_cast<const bf16x8*>(&Bs[buf][s * PLANE + b_off + j * 16 * LDS_ROW]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x4 c = acc[i][j];
                if (NPROD >= 6) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][2], bf[0], c, 0, 0, 0);   // l h
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[2], c, 0, 0, 0);   // h l
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][1], bf[1], c, 0, 0, 0);   // m m
                }
                if (NPROD >= 3) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][1], bf[0], c, 0, 0, 0);   // m h
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[1], c, 0, 0, 0);   // h m
                }
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][0], bf[0], c, 0, 0, 0);       // h h
                acc[i][j] = c;
            }
        }
        if (t + 1 < ntiles) store_lds(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 64 + j * 16 + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) C[(size_t)(m0 + wm * 64 + i * 16 + g * 4 + q) * N + col] = acc[i][j][q];
        }
}

// plain fp32 FMA reference on the device (one thread per output) for the error of an fp32 implementation
__global__ void gemm_nt_f32_ref(const float* A, const float* B, float* C, int M, int N, int K, int rows) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N || m >= rows) return;
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc = fmaf(A[(size_t)m * K + k], B[(size_t)n * K + k], acc);
    C[(size_t)m * N + n] = acc;
}

#define HIPCHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NPROD>
static void run(const float* dA, const float* dB, float* dC, int M, int N, int K, const std::vector<float>& hA, const std::vector<float>& hB,
                const std::vector<float>& f32ref, int rows) {
    dim3 grid((M / BM) * (N / BN));
    hipLaunchKernelGGL(gemm_nt_bf16split<NPROD>, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K);
    HIPCHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_nt_bf16split<NPROD>, grid, dim3(256), 0, 0, dA, dB, dC, M, N, K);
    hipEventRecord(e1); HIPCHECK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<float> hC((size_t)rows * N);
    HIPCHECK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
    double emax = 0, e2 = 0, r2 = 0, rmax = 0, f32max = 0, f32e2 = 0;
    for (int m = 0; m < rows; ++m)
        for (int n = 0; n < N; n += 7) {
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)m * K + k] * (double)hB[(size_t)n * K + k];
            const double e = hC[(size_t)m * N + n] - ref, ef = f32ref[(size_t)m * N + n] - ref;
            emax = fmax(emax, fabs(e)); e2 += e * e; r2 += ref * ref; rmax = fmax(rmax, fabs(ref));
            f32max = fmax(f32max, fabs(ef)); f32e2 += ef * ef;
        }
    printf("bf16x%d: %.3f ms  %.1f TFLOP/s (algorithmic 2MNK)  | vs fp64: max|e|/max|ref| %.3e  ||e||/||ref|| %.3e   [plain fp32 FMA chain: %.3e  %.3e]\n",
           NPROD, ms, 2.0 * M * N * K / (ms * 1e-3) / 1e12, emax / rmax, sqrt(e2 / r2), f32max / rmax, sqrt(f32e2 / r2));
}

int main(int argc, char** argv) {
    int M = 16384, N = 3072, K = 768;
    if (argc >= 4) { M = atoi(argv[1]); N = atoi(argv[2]); K = atoi(argv[3]); }
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) * (1.0f / 16777216.0f) - 0.5f); };
    for (auto& v : hA) v = rnd() * 4.0f;                       // activations ~ U(-2, 2)
    for (auto& v : hB) v = rnd() * 0.1f;                       // weights ~ U(-0.05, 0.05)
    hA[5] = 1e-30f; hA[6] = 3.0e4f; hB[9] = -7.25e-12f;        // a few extreme magnitudes
    float *dA, *dB, *dC;
    HIPCHECK(hipMalloc(&dA, hA.size() * 4)); HIPCHECK(hipMalloc(&dB, hB.size() * 4)); HIPCHECK(hipMalloc(&dC, (size_t)M * N * 4));
    HIPCHECK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    const int rows = 64;
    std::vector<float> f32ref((size_t)rows * N);
    hipLaunchKernelGGL(gemm_nt_f32_ref, dim3((N + 255) / 256, rows), dim3(256), 0, 0, dA, dB, dC, M, N, K, rows);
    HIPCHECK(hipMemcpy(f32ref.data(), dC, f32ref.size() * 4, hipMemcpyDeviceToHost));
    printf("M=%d N=%d K=%d\n", M, N, K);
    run<6>(dA, dB, dC, M, N, K, hA, hB, f32ref, rows);
    run<3>(dA, dB, dC, M, N, K, hA, hB, f32ref, rows);
    run<1>(dA, dB, dC, M, N, K, hA, hB, f32ref, rows);
    return 0;
}
