// pool_bwd.hip -- the two products of the mini-PointNet backward whose left operand is the gradient of a max-pool (gfx950).
//
// Encoder.forward ends with  feature_global = torch.max(feature, dim=2)  over the n points of each group (models/dvae.py:209-216, :262-275), so
// the gradient that reaches the last conv,  dh[g*n + j][c] = (arg[g][c] == j ? dout[g][c] : 0),  has exactly ONE non-zero per (group, channel):
// of the n * C entries of a group only C are live.  The dense GEMMs that consume dh (gemm_q16 with the scatter generated on load) spend
// n = 32 .. 64 times the arithmetic the products need -- 2 x 0.84 ms of the Stage-II step, 2 x 6.5 ms at C5.  These kernels walk the live
// entries instead; both are bound by the bytes of the DENSE operand (da written once, the activated conv input read once), not by flops.
//
//   da[g*n + j][:] = sum over {c : arg[g][c] == j} of dout[g][c] * W[c][:]                       (pool_bwd_dx_kernel)
//   dW[c][:]       = sum over g of dout[g][c] * act(X[g*n + arg[g][c]][:])                       (pool_bwd_dw_kernel)
//
// Summation order is fixed by the shapes alone (ascending c inside a row; ascending g inside a split, splits folded in order), so the results
// are run-to-run and process-to-process identical; they differ from the dense path's MFMA summation order in the last bits.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------------------ da = dh . W
// One workgroup per group.  The C channels are bucketed by their arg row (stable: ascending c inside a bucket) with two passes of a
// (row, channel-segment) decomposition over the 256 threads, then every thread owns VEC adjacent columns and walks the buckets in order:
// one W row (L2-resident, C x N x 4 bytes) per live entry, one store per output row.
template <int VEC> struct VecT;
template <> struct VecT<1> { using T = float; };
template <> struct VecT<2> { using T = float2; };
template <> struct VecT<4> { using T = float4; };
__device__ __forceinline__ void vfma(float& a, float d, float w) { a = fmaf(d, w, a); }
__device__ __forceinline__ void vfma(float2& a, float d, float2 w) { a.x = fmaf(d, w.x, a.x); a.y = fmaf(d, w.y, a.y); }
__device__ __forceinline__ void vfma(float4& a, float d, float4 w) {
    a.x = fmaf(d, w.x, a.x); a.y = fmaf(d, w.y, a.y); a.z = fmaf(d, w.z, a.z); a.w = fmaf(d, w.w, a.w);
}
__device__ __forceinline__ void vzero(float& a) { a = 0.f; }
__device__ __forceinline__ void vzero(float2& a) { a = make_float2(0.f, 0.f); }
__device__ __forceinline__ void vzero(float4& a) { a = make_float4(0.f, 0.f, 0.f, 0.f); }

template <int VEC>
__global__ __launch_bounds__(256) void pool_bwd_dx_kernel(const float* __restrict__ dout, const int32_t* __restrict__ arg, const float* __restrict__ W,
                                                          int ldw, int n, int C, float* __restrict__ out, int ldo) {
    using V = typename VecT<VEC>::T;
    extern __shared__ unsigned char smem[];
    int* s_arg = reinterpret_cast<int*>(smem);                                   // [C]
    float* s_d = reinterpret_cast<float*>(s_arg + C);                            // [C]
    int* s_start = reinterpret_cast<int*>(s_d + C);                              // [n + 1]
    int* s_wsum = s_start + n + 1;                                               // [4]
    unsigned short* s_ord = reinterpret_cast<unsigned short*>(s_wsum + 4);       // [C]
    const int tid = threadIdx.x, g = blockIdx.x;
    for (int c = tid; c < C; c += 256) { s_arg[c] = arg[(size_t)g * C + c]; s_d[c] = dout[(size_t)g * C + c]; }
    __syncthreads();
    // thread (r, sg): row r, channel segment sg of S = 256 / n
    const int S = 256 / n, r_ = tid / S, sg = tid - r_ * S, L = (C + S - 1) / S, c_lo = sg * L, c_hi = min(C, c_lo + L);
    int cnt = 0;
    for (int c = c_lo; c < c_hi; ++c) cnt += (s_arg[c] == r_);
    // exclusive scan of cnt in tid order (= row-major, segments ascending: the stable bucket order)
    int incl = cnt;
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; ++w) base += s_wsum[w];
    int p = base + incl - cnt;
    if (sg == 0) s_start[r_] = p;
    if (tid == 255) s_start[n] = base + incl;
    for (int c = c_lo; c < c_hi; ++c) if (s_arg[c] == r_) s_ord[p++] = (unsigned short)c;
    __syncthreads();

    const int total = s_start[n];
    const float* __restrict__ Wc = W + (size_t)tid * VEC;
    float* __restrict__ orow = out + (size_t)g * n * ldo + (size_t)tid * VEC;
    constexpr int U = 8;
    int r = 0, nb = s_start[1];
    V acc; vzero(acc);
    for (int j0 = 0; j0 < total; j0 += U) {
        V w[U]; float dv[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int j = min(j0 + k, total - 1);
            const int c = s_ord[j];
            dv[k] = s_d[c];
            w[k] = *reinterpret_cast<const V*>(Wc + (size_t)c * ldw);
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int j = j0 + k;
            if (j < total) {                                                     // (uniform)
                while (j == nb) { *reinterpret_cast<V*>(orow + (size_t)r * ldo) = acc; vzero(acc); ++r; nb = s_start[r + 1]; }
                vfma(acc, dv[k], w[k]);
            }
        }
    }
    for (; r < n; ++r) { *reinterpret_cast<V*>(orow + (size_t)r * ldo) = acc; vzero(acc); }
}

// ------------------------------------------------------------------------------------------------------------------- dW = dh^T . act(X)
// Workgroup = (64-column slice, tile of 4 * CPW channels, range of groups).  Per group the [n x 64] slice of X goes through LDS (BatchNorm
// affine + ReLU applied on the way) together with the tile's (arg, dout) pairs -- arg already as the byte offset of its slice row, entries
// outside [0, n) zeroed -- double-buffered, the next group's data in flight in registers.  Wave w owns CPW consecutive channels and every
// lane one column: per (group, channel) one broadcast LDS read of the pair (4 channels per ds_read_b128), one LDS read of the arg row, one FMA.
template <int CPW, bool AFF>
__global__ __launch_bounds__(256, (CPW > 64 ? 2 : 3)) void pool_bwd_dw_kernel(const float* __restrict__ dout, const int32_t* __restrict__ arg,
                                                                              const float* __restrict__ X, int ldx, const float* __restrict__ scale,
                                                                              const float* __restrict__ shift, int n, int C, int G, int gps,
                                                                              float* __restrict__ part, int ldp, size_t split_stride) {
    constexpr int CT = 4 * CPW, NP = (CT + 255) / 256;
    extern __shared__ float4 smem4[];
    unsigned char* smem = reinterpret_cast<unsigned char*>(smem4);
    const int slice_bytes = n * 256, buf_bytes = slice_bytes + CT * 8;           // [n][64] floats | CT row offsets | CT dout
    const int tid = threadIdx.x, cl = tid & 63, wv = tid >> 6;
    const int n0 = blockIdx.x * 64, ctile = blockIdx.y * CT;
    const int g0 = blockIdx.z * gps, g1 = min(G, g0 + gps);
    const int nf4 = n * 16;                                                      // float4s of one slice
    constexpr int MAXF = 4;                                                      // n <= 64
    const int c4 = tid & 15;
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (AFF) { sc4 = *reinterpret_cast<const float4*>(scale + n0 + c4 * 4); sh4 = *reinterpret_cast<const float4*>(shift + n0 + c4 * 4); }
    float4 stg[MAXF]; int sa[NP]; float sdv[NP];
    auto fetch = [&](int g) {
#pragma unroll
        for (int i = 0; i < MAXF; ++i) {
            const int f = tid + 256 * i;
            if (f < nf4) stg[i] = *reinterpret_cast<const float4*>(X + ((size_t)g * n + (f >> 4)) * ldx + n0 + c4 * 4);
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int c = tid + 256 * i;
            if (c < CT) { sa[i] = arg[(size_t)g * C + ctile + c]; sdv[i] = dout[(size_t)g * C + ctile + c]; }
        }
    };
    auto put = [&](int buf) {
        unsigned char* b = smem + buf * buf_bytes;
#pragma unroll
        for (int i = 0; i < MAXF; ++i) {
            const int f = tid + 256 * i;
            if (f < nf4) {
                float4 v = stg[i];
                if (AFF) {
                    v.x = fmaxf(v.x * sc4.x + sh4.x, 0.f); v.y = fmaxf(v.y * sc4.y + sh4.y, 0.f);
                    v.z = fmaxf(v.z * sc4.z + sh4.z, 0.f); v.w = fmaxf(v.w * sc4.w + sh4.w, 0.f);
                }
                *reinterpret_cast<float4*>(b + f * 16) = v;
            }
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int c = tid + 256 * i;
            if (c < CT) {
                const bool ok = (unsigned)sa[i] < (unsigned)n;
                *reinterpret_cast<int*>(b + slice_bytes + c * 4) = ok ? sa[i] * 256 : 0;
                *reinterpret_cast<float*>(b + slice_bytes + CT * 4 + c * 4) = ok ? sdv[i] : 0.f;
            }
        }
    };
    float acc[CPW];
#pragma unroll
    for (int i = 0; i < CPW; ++i) acc[i] = 0.f;
    if (g0 < g1) { fetch(g0); put(0); }
    __syncthreads();
    int buf = 0;
    for (int g = g0; g < g1; ++g) {
        if (g + 1 < g1) fetch(g + 1);
        const unsigned char* b = smem + buf * buf_bytes;
        const unsigned char* col = b + cl * 4;
        const int4* ro = reinterpret_cast<const int4*>(b + slice_bytes + wv * CPW * 4);
        const float4* dd = reinterpret_cast<const float4*>(b + slice_bytes + CT * 4 + wv * CPW * 4);
#pragma unroll
        for (int i = 0; i < CPW; i += 4) {
            const int4 r4 = ro[i / 4];                                           // (same address in every lane: broadcast)
            const float4 d4 = dd[i / 4];
            acc[i + 0] = fmaf(d4.x, *reinterpret_cast<const float*>(col + r4.x), acc[i + 0]);
            acc[i + 1] = fmaf(d4.y, *reinterpret_cast<const float*>(col + r4.y), acc[i + 1]);
            acc[i + 2] = fmaf(d4.z, *reinterpret_cast<const float*>(col + r4.z), acc[i + 2]);
            acc[i + 3] = fmaf(d4.w, *reinterpret_cast<const float*>(col + r4.w), acc[i + 3]);
            if ((i & 15) == 12) {                                                // 16 channels in flight at a time: pin the chunk's FMAs here
                float* q = acc + i - 12;                                         // (the scheduler otherwise hoists all CPW reads and spills)
                asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]), "+v"(q[8]), "+v"(q[9]),
                             "+v"(q[10]), "+v"(q[11]), "+v"(q[12]), "+v"(q[13]), "+v"(q[14]), "+v"(q[15]) : : "memory");
            }
        }
        if (g + 1 < g1) put(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    float* __restrict__ o = part + (size_t)blockIdx.z * split_stride + (size_t)(ctile + wv * CPW) * ldp + n0 + cl;
#pragma unroll
    for (int i = 0; i < CPW; ++i) o[(size_t)i * ldp] = acc[i];
}

__global__ __launch_bounds__(256) void pool_bwd_fold_kernel(const float* __restrict__ part, int splits, size_t split_stride, int C, int N4,
                                                            float* __restrict__ out, int ldo) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)C * N4) return;
    const int c = (int)(i / N4), q = (int)(i - (long long)c * N4);
    const float4* __restrict__ p = reinterpret_cast<const float4*>(part) + i;
    float4 a = p[0];
    for (int s = 1; s < splits; ++s) { const float4 v = p[(size_t)s * (split_stride / 4)]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    *reinterpret_cast<float4*>(out + (size_t)c * ldo + q * 4) = a;
}

bool pool_geom_ok(int n, int C) { return n > 0 && n <= 64 && 256 % n == 0 && C > 0 && C < 65536; }
int dw_cpw(int C) { return C % 384 == 0 ? 96 : C % 256 == 0 ? 64 : C % 128 == 0 ? 32 : 0; }
int dw_splits(int G, int C, int N, size_t workspace_bytes) {
    const int tiles = (N / 64) * (C / (4 * dw_cpw(C)));
    int s = (1024 + tiles - 1) / tiles;                                          // ~4 workgroups per CU
    s = min(s, max(1, G / 8));                                                   // at least 8 groups per workgroup
    const size_t cap = workspace_bytes / ((size_t)C * N * sizeof(float));
    if ((size_t)s > cap) s = (int)cap;
    return s;
}
}  // namespace

extern "C" int act_group_max_bwd_matmul_f32(const float* dout, const int32_t* arg, int G, int n, int C, const float* w, int ldw, int N, float* dx,
                                            int lddx, act_stream_t stream) {
    if (!dout || !arg || !w || !dx) return ACT_E_NULLPTR;
    if (G < 0 || !pool_geom_ok(n, C) || N <= 0 || (N != 256 && N != 512 && N != 1024) || ldw < N || lddx < N || (ldw & 3) || (lddx & 3) ||
        ((uintptr_t)w & 15) || ((uintptr_t)dx & 15))
        return ACT_E_BADARG;
    if (G == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_MAXPOOL_BWD, s, 2.0 * G * (double)C * N, 4.0 * G * ((double)n * N + 2.0 * C));
    const size_t lds = (size_t)C * 8 + (size_t)(n + 1 + 4) * 4 + (size_t)C * 2;
    if (N == 256)      hipLaunchKernelGGL(pool_bwd_dx_kernel<1>, dim3(G), dim3(256), lds, s, dout, arg, w, ldw, n, C, dx, lddx);
    else if (N == 512) hipLaunchKernelGGL(pool_bwd_dx_kernel<2>, dim3(G), dim3(256), lds, s, dout, arg, w, ldw, n, C, dx, lddx);
    else               hipLaunchKernelGGL(pool_bwd_dx_kernel<4>, dim3(G), dim3(256), lds, s, dout, arg, w, ldw, n, C, dx, lddx);
    ACT_LAUNCH_CHECK(); return 0;
}

extern "C" size_t act_group_max_bwd_wgrad_workspace(int G, int n, int C, int N) {
    if (G <= 0 || !pool_geom_ok(n, C) || dw_cpw(C) == 0 || N <= 0 || (N & 63)) return 0;
    const int s = dw_splits(G, C, N, (size_t)-1);
    return s > 1 ? (size_t)s * C * N * sizeof(float) : 0;
}

extern "C" int act_group_max_bwd_wgrad_f32(const float* dout, const int32_t* arg, int G, int n, int C, const float* x, int ldx, int N, const float* scale,
                                           const float* shift, float* dw, int lddw, float* workspace, size_t workspace_bytes, act_stream_t stream) {
    if (!dout || !arg || !x || !dw) return ACT_E_NULLPTR;
    if ((scale == nullptr) != (shift == nullptr)) return ACT_E_NULLPTR;
    const int cpw = pool_geom_ok(n, C) ? dw_cpw(C) : 0;
    if (G <= 0 || cpw == 0 || N <= 0 || (N & 63) || ldx < N || lddw < N || (ldx & 3) || (lddw & 3) || ((uintptr_t)x & 15) || ((uintptr_t)dw & 15) ||
        (scale && (((uintptr_t)scale | (uintptr_t)shift) & 15)))
        return ACT_E_BADARG;
    int splits = dw_splits(G, C, N, workspace ? workspace_bytes : 0);
    if (splits < 1) splits = 1;
    if (splits > 1 && (((uintptr_t)workspace) & 15)) return ACT_E_BADARG;
    const int gps = (G + splits - 1) / splits;
    splits = (G + gps - 1) / gps;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_MAXPOOL_BWD, s, 2.0 * G * (double)C * N, 4.0 * ((double)G * n * N * (C / (4 * cpw)) + 2.0 * G * C * (N / 64) + (double)C * N));
    float* part = splits > 1 ? workspace : dw;
    const int ldp = splits > 1 ? N : lddw;
    const size_t stride = (size_t)C * N;
    const dim3 grid(N / 64, C / (4 * cpw), splits);
    const size_t lds = (size_t)2 * ((size_t)n * 256 + (size_t)4 * cpw * 8);
#define LAUNCH_DW(CPW_)                                                                                                                          \
    do {                                                                                                                                         \
        if (scale) hipLaunchKernelGGL((pool_bwd_dw_kernel<CPW_, true>), grid, dim3(256), lds, s, dout, arg, x, ldx, scale, shift, n, C, G, gps,    \
                                      part, ldp, stride);                                                                                        \
        else       hipLaunchKernelGGL((pool_bwd_dw_kernel<CPW_, false>), grid, dim3(256), lds, s, dout, arg, x, ldx, scale, shift, n, C, G, gps,   \
                                      part, ldp, stride);                                                                                        \
    } while (0)
    if (cpw == 96) LAUNCH_DW(96); else if (cpw == 64) LAUNCH_DW(64); else LAUNCH_DW(32);
#undef LAUNCH_DW
    ACT_LAUNCH_CHECK();
    if (splits > 1) {
        const long long total = (long long)C * (N / 4);
        hipLaunchKernelGGL(pool_bwd_fold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, part, splits, stride, C, N / 4, dw, lddw);
        ACT_LAUNCH_CHECK();
    }
    return 0;
}
