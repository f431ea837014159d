// chamfer.hip -- Chamfer nearest-neighbour distance forward / backward for gfx950.
//
// Reference semantics: extensions/chamfer_dist/chamfer.cu:15-229 (strict '<' scan => lowest index on ties;
// backward = scatter of 2*g*(p1-p2)).  Re-designed for the shapes the ACT Stage-I loss actually feeds it
// (models/dvae.py:455-460: 8192 independent pairs of <=32-point clouds) and for whole-cloud validation:
//  * small path (n,m <= 64): ONE WAVE PER PAIR.  Both clouds live in registers (one point per lane); the scan
//    over the other cloud is a sequence of v_readlane broadcasts (SGPR operands), both directions in one launch.
//  * large path: 256 query points per workgroup, the other cloud streamed through a 1024-point LDS tile.
//  * backward is a GATHER (each point sums the contributions that reference it, ascending index), so it is
//    bit-reproducible, unlike the reference's atomicAdd scatter.
#include "common.h"

__device__ __forceinline__ float rl(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// Squared distance in the two roundings a maintainer can meet.  FMA = false: every product and sum rounded (the oracle convention, what a
// build with contraction off computes).  FMA = true: the form an FMA-contracting compiler (nvcc's default -fmad=true, LLVM's fadd(fmul, fmul)
// combine) makes of chamfer.cu:43-57's  x2*x2 + y2*y2 + z2*z2  with x2 = b - a:  fma(z2, z2, fma(x2, x2, y2*y2))  -- the left product of
// the first sum is fused, the right one is a rounded multiply, the last product is fused again.  Same distances up to two ulps; on (near)
// ties the strict '<' scan may then keep another index, which is the point of the switch.
template <bool FMA>
__device__ __forceinline__ float ch_sqdist(float bx, float by, float bz, float ax, float ay, float az) {
    if constexpr (!FMA) return sqdist3(bx, by, bz, ax, ay, az);
    const float dx = __fsub_rn(bx, ax), dy = __fsub_rn(by, ay), dz = __fsub_rn(bz, az);
    return __fmaf_rn(dz, dz, __fmaf_rn(dx, dx, __fmul_rn(dy, dy)));
}

// ------------------------------------------------------------------ small: one wave per pair
template <bool FMA>
__global__ __launch_bounds__(256) void chamfer_small_fwd(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                         int B, int n, int m, float* __restrict__ dist1,
                                                         float* __restrict__ dist2, int32_t* __restrict__ idx1,
                                                         int32_t* __restrict__ idx2) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= B) return;
    float ax = 0.f, ay = 0.f, az = 0.f, bx = 0.f, by = 0.f, bz = 0.f;
    if (lane < n) { const float* p = xyz1 + ((size_t)i * n + lane) * 3; ax = p[0]; ay = p[1]; az = p[2]; }
    if (lane < m) { const float* p = xyz2 + ((size_t)i * m + lane) * 3; bx = p[0]; by = p[1]; bz = p[2]; }
    float best1 = 0.f, best2 = 0.f; int bi1 = 0, bi2 = 0;
    for (int k = 0; k < m; ++k) {             // for every point of cloud 1: nearest in cloud 2
        const float d = ch_sqdist<FMA>(rl(bx, k), rl(by, k), rl(bz, k), ax, ay, az);
        if (k == 0 || d < best1) { best1 = d; bi1 = k; }
    }
    for (int k = 0; k < n; ++k) {
        const float d = ch_sqdist<FMA>(rl(ax, k), rl(ay, k), rl(az, k), bx, by, bz);
        if (k == 0 || d < best2) { best2 = d; bi2 = k; }
    }
    if (lane < n) { dist1[(size_t)i * n + lane] = best1; idx1[(size_t)i * n + lane] = bi1; }
    if (lane < m) { dist2[(size_t)i * m + lane] = best2; idx2[(size_t)i * m + lane] = bi2; }
}

__global__ __launch_bounds__(256) void chamfer_small_bwd(const float* __restrict__ xyz1, const float* __restrict__ xyz2,
                                                         const int32_t* __restrict__ idx1, const int32_t* __restrict__ idx2,
                                                         const float* __restrict__ g1, const float* __restrict__ g2,
                                                         int B, int n, int m, float* __restrict__ gx1, float* __restrict__ gx2) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (i >= B) return;
    float ax = 0.f, ay = 0.f, az = 0.f, bx = 0.f, by = 0.f, bz = 0.f, ga = 0.f, gb = 0.f; int ia = -1, ib = -1;
    if (lane < n) { const float* p = xyz1 + ((size_t)i * n + lane) * 3; ax = p[0]; ay = p[1]; az = p[2];
                    ga = 2.0f * g1[(size_t)i * n + lane]; ia = idx1[(size_t)i * n + lane]; }
    if (lane < m) { const float* p = xyz2 + ((size_t)i * m + lane) * 3; bx = p[0]; by = p[1]; bz = p[2];
                    gb = 2.0f * g2[(size_t)i * m + lane]; ib = idx2[(size_t)i * m + lane]; }
    // own terms: grad_xyz1[j] += g*(p1[j]-p2[idx1[j]]) ; grad_xyz2[k] += g*(p2[k]-p1[idx2[k]])
    float r1x = 0.f, r1y = 0.f, r1z = 0.f, r2x = 0.f, r2y = 0.f, r2z = 0.f;
    if (lane < n) { const float* p = xyz2 + ((size_t)i * m + ia) * 3; r1x = ga * (ax - p[0]); r1y = ga * (ay - p[1]); r1z = ga * (az - p[2]); }
    if (lane < m) { const float* p = xyz1 + ((size_t)i * n + ib) * 3; r2x = gb * (bx - p[0]); r2y = gb * (by - p[1]); r2z = gb * (bz - p[2]); }
    // referenced-by terms, ascending index
    for (int k = 0; k < m; ++k) {             // cloud-2 point k points at cloud-1 point idx2[k]
        const int t = __builtin_amdgcn_readlane(ib, k);
        const float g = rl(gb, k), x = rl(bx, k), y = rl(by, k), z = rl(bz, k);
        if (t == lane) { r1x -= g * (x - ax); r1y -= g * (y - ay); r1z -= g * (z - az); }
    }
    for (int k = 0; k < n; ++k) {
        const int t = __builtin_amdgcn_readlane(ia, k);
        const float g = rl(ga, k), x = rl(ax, k), y = rl(ay, k), z = rl(az, k);
        if (t == lane) { r2x -= g * (x - bx); r2y -= g * (y - by); r2z -= g * (z - bz); }
    }
    if (lane < n) { float* o = gx1 + ((size_t)i * n + lane) * 3; o[0] = r1x; o[1] = r1y; o[2] = r1z; }
    if (lane < m) { float* o = gx2 + ((size_t)i * m + lane) * 3; o[0] = r2x; o[1] = r2y; o[2] = r2z; }
}

// ------------------------------------------------------------------ large: LDS-tiled
#define CH_TILE 1024
template <bool FMA>
__global__ __launch_bounds__(256) void chamfer_large_fwd(const float* __restrict__ a, int n, const float* __restrict__ bsrc, int m,
                                                         float* __restrict__ dist, int32_t* __restrict__ idx) {
    __shared__ float buf[CH_TILE * 3];
    const int i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    float x1 = 0.f, y1 = 0.f, z1 = 0.f;
    if (j < n) { const float* p = a + ((size_t)i * n + j) * 3; x1 = p[0]; y1 = p[1]; z1 = p[2]; }
    float best = 0.f; int bi = 0;
    for (int k2 = 0; k2 < m; k2 += CH_TILE) {
        const int cnt = min(CH_TILE, m - k2);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt * 3; t += blockDim.x) buf[t] = bsrc[((size_t)i * m + k2) * 3 + t];
        __syncthreads();
        if (j < n) {
            for (int k = 0; k < cnt; ++k) {
                const float d = ch_sqdist<FMA>(buf[k * 3], buf[k * 3 + 1], buf[k * 3 + 2], x1, y1, z1);
                if ((k2 + k) == 0 || d < best) { best = d; bi = k2 + k; }
            }
        }
    }
    if (j < n) { dist[(size_t)i * n + j] = best; idx[(size_t)i * n + j] = bi; }
}

// grad_a[j] = 2*ga[j]*(a[j]-b[idxa[j]])  -  sum_{k : idxb[k]==j} 2*gb[k]*(b[k]-a[j])     (ascending k)
__global__ __launch_bounds__(256) void chamfer_large_bwd(const float* __restrict__ a, int n, const float* __restrict__ b, int m,
                                                         const int32_t* __restrict__ idxa, const int32_t* __restrict__ idxb,
                                                         const float* __restrict__ ga, const float* __restrict__ gb,
                                                         float* __restrict__ grad_a) {
    __shared__ float buf[CH_TILE * 4];
    __shared__ int ibuf[CH_TILE];
    const int i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    float x1 = 0.f, y1 = 0.f, z1 = 0.f, rx = 0.f, ry = 0.f, rz = 0.f;
    if (j < n) {
        const float* p = a + ((size_t)i * n + j) * 3; x1 = p[0]; y1 = p[1]; z1 = p[2];
        const float* q = b + ((size_t)i * m + idxa[(size_t)i * n + j]) * 3;
        const float g = 2.0f * ga[(size_t)i * n + j];
        rx = g * (x1 - q[0]); ry = g * (y1 - q[1]); rz = g * (z1 - q[2]);
    }
    for (int k2 = 0; k2 < m; k2 += CH_TILE) {
        const int cnt = min(CH_TILE, m - k2);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += blockDim.x) {
            const float* q = b + ((size_t)i * m + k2 + t) * 3;
            buf[t * 4] = q[0]; buf[t * 4 + 1] = q[1]; buf[t * 4 + 2] = q[2]; buf[t * 4 + 3] = 2.0f * gb[(size_t)i * m + k2 + t];
            ibuf[t] = idxb[(size_t)i * m + k2 + t];
        }
        __syncthreads();
        if (j < n) {
            for (int k = 0; k < cnt; ++k) {
                if (ibuf[k] == j) {
                    const float g = buf[k * 4 + 3];
                    rx -= g * (buf[k * 4] - x1); ry -= g * (buf[k * 4 + 1] - y1); rz -= g * (buf[k * 4 + 2] - z1);
                }
            }
        }
    }
    if (j < n) { float* o = grad_a + ((size_t)i * n + j) * 3; o[0] = rx; o[1] = ry; o[2] = rz; }
}

extern "C" int act_chamfer_fwd_ex_f32(const float* xyz1, const float* xyz2, int B, int n, int m, float* dist1, float* dist2,
                                      int32_t* idx1, int32_t* idx2, int fma_contract, act_stream_t stream) {
    if (B == 0) return 0;
    if (!xyz1 || !xyz2 || !dist1 || !dist2 || !idx1 || !idx2) return ACT_E_NULLPTR;
    if (B < 0 || n <= 0 || m <= 0 || B > 65535 * 4) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_CHAMFER_FWD, s, 0.0, (double)B * 20.0 * (n + m));      // 12(n+m) read + 8(n+m) write
    if (n <= 64 && m <= 64) {
        if (fma_contract) hipLaunchKernelGGL(chamfer_small_fwd<true>, dim3((B + 3) / 4), dim3(256), 0, s, xyz1, xyz2, B, n, m, dist1, dist2, idx1, idx2);
        else              hipLaunchKernelGGL(chamfer_small_fwd<false>, dim3((B + 3) / 4), dim3(256), 0, s, xyz1, xyz2, B, n, m, dist1, dist2, idx1, idx2);
    } else {
        if (B > 65535) return ACT_E_BADARG;
        if (fma_contract) {
            hipLaunchKernelGGL(chamfer_large_fwd<true>, dim3((n + 255) / 256, B), dim3(256), 0, s, xyz1, n, xyz2, m, dist1, idx1);
            hipLaunchKernelGGL(chamfer_large_fwd<true>, dim3((m + 255) / 256, B), dim3(256), 0, s, xyz2, m, xyz1, n, dist2, idx2);
        } else {
            hipLaunchKernelGGL(chamfer_large_fwd<false>, dim3((n + 255) / 256, B), dim3(256), 0, s, xyz1, n, xyz2, m, dist1, idx1);
            hipLaunchKernelGGL(chamfer_large_fwd<false>, dim3((m + 255) / 256, B), dim3(256), 0, s, xyz2, m, xyz1, n, dist2, idx2);
        }
    }
    ACT_LAUNCH_CHECK();
    return 0;
}
extern "C" int act_chamfer_fwd_f32(const float* xyz1, const float* xyz2, int B, int n, int m, float* dist1, float* dist2,
                                   int32_t* idx1, int32_t* idx2, act_stream_t stream) {
    return act_chamfer_fwd_ex_f32(xyz1, xyz2, B, n, m, dist1, dist2, idx1, idx2, 0, stream);
}

extern "C" int act_chamfer_bwd_f32(const float* xyz1, const float* xyz2, const int32_t* idx1, const int32_t* idx2,
                                   const float* g1, const float* g2, int B, int n, int m, float* gx1, float* gx2,
                                   act_stream_t stream) {
    if (B == 0) return 0;
    if (!xyz1 || !xyz2 || !idx1 || !idx2 || !g1 || !g2 || !gx1 || !gx2) return ACT_E_NULLPTR;
    if (B < 0 || n <= 0 || m <= 0) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_CHAMFER_BWD, s, 0.0, (double)B * 32.0 * (n + m));
    if (n <= 64 && m <= 64) {
        hipLaunchKernelGGL(chamfer_small_bwd, dim3((B + 3) / 4), dim3(256), 0, s, xyz1, xyz2, idx1, idx2, g1, g2, B, n, m, gx1, gx2);
    } else {
        if (B > 65535) return ACT_E_BADARG;
        hipLaunchKernelGGL(chamfer_large_bwd, dim3((n + 255) / 256, B), dim3(256), 0, s, xyz1, n, xyz2, m, idx1, idx2, g1, g2, gx1);
        hipLaunchKernelGGL(chamfer_large_bwd, dim3((m + 255) / 256, B), dim3(256), 0, s, xyz2, m, xyz1, n, idx2, idx1, g2, g1, gx2);
    }
    ACT_LAUNCH_CHECK();
    return 0;
}
