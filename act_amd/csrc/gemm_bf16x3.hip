// gemm_bf16x3.hip -- OPT-IN NT GEMM on the bf16 matrix cores with split operands ("bf16x3"), for the FROZEN teacher only (round 5).
//
//   C[M,N] = epilogue( (A_hi + A_lo) . (B_hi + B_lo)^T )  with the lo.lo term dropped:  A_hi.B_hi + A_hi.B_lo + A_lo.B_hi, fp32 accumulation
//
// hi = bf16(x) (round to nearest even), lo = bf16(x - hi): 16 significand bits per operand instead of 24.  Every product of two planes is exact in the
// MFMA's fp32 accumulator; what is lost is the 2^-17 tail of each operand and the 2^-16 lo.lo term: 4.2e-6 relative per product against fp64, and the
// frozen teacher's features move by 6e-6 .. 7e-6 of their range when the 48 Linear layers of its ViT blocks run this way (parity bar 1e-4; measured with
// the CPU oracle, benchmarks/bf16x3_teacher_numerics.py, and on the device, tests/test_gpu_bf16x3.py).  It is NOT the default: the product path is
// f32-input MFMA and every headline number is measured on it; ACT_TEACHER_BF16X3=1 routes the teacher's five GEMMs per ViT layer here and bench.py
// reports that configuration on a separate line.
//
// Why it is fast: v_mfma_f32_16x16x32_bf16 has 16 x the rate of the f32-input MFMA, three products cost 3/16 of the matrix time, and two bf16 planes are
// the same 4 bytes per element as fp32 -- so the kernel moves exactly the bytes of the fp32 kernel and is bound by the L2 -> LDS fill path and the LDS,
// not by the matrix pipe: 128 x 128 tile, 32-deep K tiles, double-buffered LDS (ONE barrier per K tile), global prefetch two K tiles ahead in two register
// sets.  8192 x 3072 x 768: 282 TFLOP/s-equivalent (shipped f32 kernel: 135), benchmarks/micro/bf16x3_planes.hip -> profiles/r05_micro_bf16x3_planes.txt.
// A lesson kept in the code: the staging arrays are NATIVE vector types (ext_vector_type); arrays of HIP's `uint4` struct stay in scratch memory.
#include "gemm_common.h"
#include <stdlib.h>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ unsigned bf16_rne(float x) {          // finite inputs (activations / weights): no NaN handling needed
    unsigned u = __float_as_uint(x);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return u >> 16;
}

// fp32 [R][K] (row stride ldx) -> hi [R][K], lo [R][K] bf16, four elements per thread (K % 4 == 0)
__global__ __launch_bounds__(256) void split_bf16x2_kernel(const float* __restrict__ x, int K4, int ldx, long long n4, unsigned short* __restrict__ hi,
                                                           unsigned short* __restrict__ lo) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / K4; const int c4 = (int)(i - row * K4);
        const float4 v = *reinterpret_cast<const float4*>(x + row * ldx + c4 * 4);
        const float e[4] = {v.x, v.y, v.z, v.w};
        unsigned h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            h[j] = bf16_rne(e[j]);
            l[j] = bf16_rne(e[j] - __uint_as_float(h[j] << 16));
        }
        reinterpret_cast<uint2*>(hi)[i] = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
        reinterpret_cast<uint2*>(lo)[i] = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
    }
}

constexpr int BK = 32;                      // bf16 per K tile = 64-byte rows
constexpr int LROW = 40;                    // bf16 per LDS row: 32 + 8 pad (80-byte pitch: conflict-free ds_read_b128 / ds_write_b128)

struct X3Params {
    const unsigned short *Ah, *Al, *Bh, *Bl;   // planes, row-major [rows][K]
    float* C; int ldc;                         // fp32 result (nullable when the planes below are given)
    unsigned short *Oh, *Ol;                   // optional: the result ALSO / INSTEAD as (hi, lo) bf16 planes [M][N] -- the A operand of the next split-bf16 product
    int M, N, K;
    act_gemm_epilogue_t epi;
};

template <int BM, int BN, int ACT, bool PLANES>
__global__ __launch_bounds__(256, 2) void sgemm_nt_bf16x3_kernel(const X3Params p) {
    constexpr int TM = BM / 32, TN = BN / 32;                 // 16 x 16 tiles per wave (2 x 2 waves)
    constexpr int PA = BM * LROW, PB = BN * LROW;             // bf16 per plane image
    constexpr int NA = BM / 64, NB = BN / 64;                 // 16-byte chunks per thread, plane and K tile
    constexpr int STAGE = 2 * (PA + PB);
    __shared__ __attribute__((aligned(16))) unsigned short L[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = p.N / BN, tiles_m = p.M / BM;
    int wg = blockIdx.x;
    {                                                             // XCD-aware remap (workgroup b runs on XCD b % 8) + grouped rasterisation (gemm_common.h)
        const int nwg = tiles_m * tiles_n, q = nwg >> 3, r = nwg & 7, xcd = wg & 7, loc = wg >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    constexpr int GM = 8;
    const int per_group = GM * tiles_n, group = wg / per_group, first_m = group * GM;
    const int gsz = min(tiles_m - first_m, GM), in_group = wg - group * per_group;
    const int tile_m = first_m + in_group % gsz, tile_n = in_group / gsz;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int K = p.K, ntiles = K / BK, last = ntiles - 1;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int srow = tid >> 2, sch = tid & 3;                 // staging: (row, 16-byte chunk) of a [rows][32 bf16] plane tile
    const size_t goa = (size_t)(m0 + srow) * K + sch * 8, gob = (size_t)(n0 + srow) * K + sch * 8;
    const unsigned short* gA[2] = {p.Ah + goa, p.Al + goa};
    const unsigned short* gB[2] = {p.Bh + gob, p.Bl + gob};
    u32x4v ra0[2][NA], rb0[2][NB], ra1[2][NA], rb1[2][NB];      // two staging sets (native vectors: arrays of HIP's uint4 struct would sit in scratch)
    const int s_off = srow * LROW + sch * 8;
    const int r = lane & 15, g = lane >> 4;
    const int a_off = (wm * (BM / 2) + r) * LROW + g * 8, b_off = (wn * (BN / 2) + r) * LROW + g * 8;
#define X3_LOAD(RA, RB, T)                                                                                                             \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < NA; ++i_) RA[s_][i_] = *reinterpret_cast<const u32x4v*>(gA[s_] + (size_t)(64 * i_) * K + (T) * BK); \
        _Pragma("unroll") for (int i_ = 0; i_ < NB; ++i_) RB[s_][i_] = *reinterpret_cast<const u32x4v*>(gB[s_] + (size_t)(64 * i_) * K + (T) * BK); \
    }
#define X3_STORE(RA, RB, STG)                                                                                                          \
    _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) {                                                                                 \
        _Pragma("unroll") for (int i_ = 0; i_ < NA; ++i_) *reinterpret_cast<u32x4v*>(&L[(STG) * STAGE + s_ * PA + s_off + 64 * i_ * LROW]) = RA[s_][i_];           \
        _Pragma("unroll") for (int i_ = 0; i_ < NB; ++i_) *reinterpret_cast<u32x4v*>(&L[(STG) * STAGE + 2 * PA + s_ * PB + s_off + 64 * i_ * LROW]) = RB[s_][i_];  \
    }
#define X3_COMPUTE(STG)                                                                                                                \
    {                                                                                                                                  \
        const unsigned short* As_ = L + (STG) * STAGE; const unsigned short* Bs_ = As_ + 2 * PA;                                       \
        bf16x8 af[TM][2];                                                                                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_)                                                                              \
            _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) af[i_][s_] = *reinterpret_cast<const bf16x8*>(&As_[s_ * PA + a_off + i_ * 16 * LROW]); \
        _Pragma("unroll") for (int j_ = 0; j_ < TN; ++j_) {                                                                            \
            bf16x8 bf[2];                                                                                                              \
            _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) bf[s_] = *reinterpret_cast<const bf16x8*>(&Bs_[s_ * PB + b_off + j_ * 16 * LROW]); \
            /* smallest terms first: lo.hi, hi.lo, hi.hi.  The B fragment is the MFMA's FIRST operand: the 16 x 16 result block comes out transposed in the   */ \
            /* register layout (lane (r, g) holds row r, columns 4g .. 4g+3), i.e. every lane owns FOUR CONSECUTIVE COLUMNS of one row -> vector epilogue  */ \
            _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0], af[i_][1], acc[i_][j_], 0, 0, 0); \
            _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[1], af[i_][0], acc[i_][j_], 0, 0, 0); \
            _Pragma("unroll") for (int i_ = 0; i_ < TM; ++i_) acc[i_][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[0], af[i_][0], acc[i_][j_], 0, 0, 0); \
        }                                                                                                                              \
    }
    // loads / stores are unconditional: past the end they re-read the last K tile and fill a stage nobody reads (no divergent code around the staging registers)
    X3_LOAD(ra0, rb0, 0)
    X3_LOAD(ra1, rb1, min(1, last))
    X3_STORE(ra0, rb0, 0)
    __syncthreads();
    for (int t = 0; t < ntiles; t += 2) {                         // ntiles is even (K % 64 == 0, checked by the launcher)
        X3_LOAD(ra0, rb0, min(t + 2, last))                       // LDS stage 0 = tile t; set 1 = tile t+1 (in flight); request tile t+2
        X3_COMPUTE(0)
        X3_STORE(ra1, rb1, 1)
        __syncthreads();
        X3_LOAD(ra1, rb1, min(t + 3, last))                       // LDS stage 1 = tile t+1; set 0 = tile t+2 (in flight); request tile t+3
        X3_COMPUTE(1)
        X3_STORE(ra0, rb0, 0)
        __syncthreads();
    }
#undef X3_LOAD
#undef X3_STORE
#undef X3_COMPUTE
    // epilogue: lane (r, g) holds row i*16 + r, columns j*16 + 4g .. 4g+3 of its wave's 64 x 64 block (operands swapped above): one float4 of bias / residual /
    // result and one 8-byte store per plane, per 16 x 16 block; same per-element arithmetic and order as the f32 kernels (epilogue_apply4_act)
    const bool vec = (((uintptr_t)p.C | (uintptr_t)p.epi.bias | (uintptr_t)p.epi.res | (uintptr_t)p.epi.aux) & 15) == 0 &&
                     ((p.ldc | (p.epi.res ? p.epi.ldr : 0) | (p.epi.aux ? p.epi.ldaux : 0)) & 3) == 0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int row = m0 + wm * (BM / 2) + i * 16 + r;
            const int col = n0 + wn * (BN / 2) + j * 16 + g * 4;
            float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            if (vec) v = epilogue_apply4_act<ACT>(p.epi, v, row, col);
            else {
                v.x = epilogue_apply<ACT>(p.epi, v.x, row, col); v.y = epilogue_apply<ACT>(p.epi, v.y, row, col + 1);
                v.z = epilogue_apply<ACT>(p.epi, v.z, row, col + 2); v.w = epilogue_apply<ACT>(p.epi, v.w, row, col + 3);
            }
            if (!PLANES || p.C) {
                float* cp = p.C + (size_t)row * p.ldc + col;
                if (vec) *reinterpret_cast<float4*>(cp) = v; else { cp[0] = v.x; cp[1] = v.y; cp[2] = v.z; cp[3] = v.w; }
            }
            if constexpr (PLANES) {
                const float e[4] = {v.x, v.y, v.z, v.w};
                unsigned h[4], l[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { h[q] = bf16_rne(e[q]); l[q] = bf16_rne(e[q] - __uint_as_float(h[q] << 16)); }
                const size_t o = (size_t)row * p.N + col;                 // N % 128 == 0, col % 4 == 0: 8-byte aligned
                *reinterpret_cast<uint2*>(p.Oh + o) = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
                *reinterpret_cast<uint2*>(p.Ol + o) = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
            }
        }
}

template <int BM, int BN>
int launch_x3(const X3Params& p, hipStream_t s) {
    const dim3 grid((p.M / BM) * (p.N / BN));
    const bool planes = p.Oh != nullptr;
    switch (p.epi.act) {
        case ACT_EPI_NONE:
            if (planes) hipLaunchKernelGGL((sgemm_nt_bf16x3_kernel<BM, BN, ACT_EPI_NONE, true>), grid, dim3(256), 0, s, p);
            else        hipLaunchKernelGGL((sgemm_nt_bf16x3_kernel<BM, BN, ACT_EPI_NONE, false>), grid, dim3(256), 0, s, p);
            break;
        case ACT_EPI_GELU:
            if (planes) hipLaunchKernelGGL((sgemm_nt_bf16x3_kernel<BM, BN, ACT_EPI_GELU, true>), grid, dim3(256), 0, s, p);
            else        hipLaunchKernelGGL((sgemm_nt_bf16x3_kernel<BM, BN, ACT_EPI_GELU, false>), grid, dim3(256), 0, s, p);
            break;
        case ACT_EPI_MUL_GELU_GRAD:                                  // (the MLP's backward through GELU: aux = the saved pre-activation)
            if (planes) hipLaunchKernelGGL((sgemm_nt_bf16x3_kernel<BM, BN, ACT_EPI_MUL_GELU_GRAD, true>), grid, dim3(256), 0, s, p);
            else        hipLaunchKernelGGL((sgemm_nt_bf16x3_kernel<BM, BN, ACT_EPI_MUL_GELU_GRAD, false>), grid, dim3(256), 0, s, p);
            break;
        default: return ACT_E_BADARG;
    }
    ACT_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int act_split_bf16x2_f32(const float* x, int R, int K, int ldx, uint16_t* hi, uint16_t* lo, act_stream_t stream) {
    if (!x || !hi || !lo) return ACT_E_NULLPTR;
    if (R < 0 || K <= 0 || (K & 3) || ldx < K || (ldx & 3) || (((uintptr_t)x | (uintptr_t)hi | (uintptr_t)lo) & 15)) return ACT_E_BADARG;
    const long long n4 = (long long)R * K / 4; if (n4 == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_ELTWISE, s, 0.0, 8.0 * R * (double)K);
    long long g = (n4 + 255) / 256; if (g > 8192) g = 8192;
    hipLaunchKernelGGL(split_bf16x2_kernel, dim3((unsigned)g), dim3(256), 0, s, x, K / 4, ldx, n4, hi, lo);
    ACT_LAUNCH_CHECK(); return 0;
}

extern "C" int act_sgemm_nt_bf16x3_supported(int M, int N, int K) { return M > 0 && N > 0 && K > 0 && M % 128 == 0 && N % 128 == 0 && K % 64 == 0; }

extern "C" int act_sgemm_nt_bf16x3_f32(int M, int N, int K, const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* b_hi, const uint16_t* b_lo,
                                       float* C, int ldc, const act_gemm_epilogue_t* epilogue, act_stream_t stream) {
    return act_sgemm_nt_bf16x3_planes_f32(M, N, K, a_hi, a_lo, b_hi, b_lo, C, ldc, nullptr, nullptr, epilogue, stream);
}

extern "C" int act_sgemm_nt_bf16x3_planes_f32(int M, int N, int K, const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* b_hi, const uint16_t* b_lo,
                                              float* C, int ldc, uint16_t* out_hi, uint16_t* out_lo, const act_gemm_epilogue_t* epilogue,
                                              act_stream_t stream) {
    if (!a_hi || !a_lo || !b_hi || !b_lo || (!C && !out_hi) || ((out_hi == nullptr) != (out_lo == nullptr))) return ACT_E_NULLPTR;
    if (!act_sgemm_nt_bf16x3_supported(M, N, K) || (C && ldc < N)) return ACT_E_BADARG;
    if ((((uintptr_t)a_hi | (uintptr_t)a_lo | (uintptr_t)b_hi | (uintptr_t)b_lo | (uintptr_t)out_hi | (uintptr_t)out_lo) & 15)) return ACT_E_BADARG;
    X3Params p{};
    p.Ah = a_hi; p.Al = a_lo; p.Bh = b_hi; p.Bl = b_lo; p.C = C; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.Oh = out_hi; p.Ol = out_lo;
    if (epilogue) p.epi = *epilogue; else { p.epi = act_gemm_epilogue_t{}; p.epi.alpha = 1.0f; }
    if (p.epi.accumulate || (p.epi.act != ACT_EPI_NONE && p.epi.act != ACT_EPI_GELU && p.epi.act != ACT_EPI_MUL_GELU_GRAD)) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_GEMM_BF16X3, s, 2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    return launch_x3<128, 128>(p, s);
}
