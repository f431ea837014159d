// pointnet.hip -- row-major [rows, channels] kernels of the mini-PointNet patch embedding and the FoldingNet decoder
// (models/dvae.py:185-275): train-mode BatchNorm (batch statistics over all rows) + ReLU forward / backward and the
// per-group max-pool forward / backward.  All HBM-bound: every kernel streams its operands exactly once with float4
// accesses; column reductions are two-stage with a fixed summation order (deterministic).
#include "common.h"

// ---------------------------------------------------------------------------------------------- column statistics
// partial[blk][0][c] = sum_r f(r,c), partial[blk][1][c] = sum_r g(r,c) over the block's rows.
//   MODE 0: f = x,            g = x*x                       (BatchNorm batch statistics)
//   MODE 1: f = dyh,          g = dyh * xhat                (BatchNorm backward sums), dyh = relu ? dy*(y>0) : dy
// block = 64 float4-columns?  no: 64 columns x 4 row-lanes like colsum, 4 rows in flight per thread.
template <int MODE>
__global__ __launch_bounds__(256) void colstats_stage1(const float* __restrict__ x, const float* __restrict__ dy,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const float* __restrict__ mean, const float* __restrict__ rstd,
                                                       int relu, int R, int C, int rows_per_block, float* __restrict__ partial,
                                                       const int32_t* __restrict__ live = nullptr, int live_n = 1) {
    __shared__ float sh[2][4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
    float f0 = 0.f, f1 = 0.f, f2 = 0.f, f3 = 0.f, g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
    if (c < C) {
        float sc = 0.f, sf = 0.f, mu = 0.f, rs = 0.f, pv = 0.f;
        if (MODE == 1) { sc = scale[c]; sf = shift[c]; mu = mean[c]; rs = rstd[c]; }
        else pv = x[c];                                  // pivot (row 0) keeps E[x^2]-E[x]^2 free of cancellation
        auto term = [&](int r, float& f, float& g) {
            // (MODE 1) rows of a group whose dy is all zero add exact zeros to both sums: skipped without changing a bit
            if (MODE == 1 && live && !live[r / live_n]) return;
            const float xv = x[(size_t)r * C + c];
            if (MODE == 0) { const float t = xv - pv; f += t; g += t * t; }
            else {
                float d = dy[(size_t)r * C + c];
                if (relu && !(xv * sc + sf > 0.f)) d = 0.f;
                f += d; g += d * ((xv - mu) * rs);
            }
        };
        int r = r0 + ry;
        for (; r + 12 < r1; r += 16) { term(r, f0, g0); term(r + 4, f1, g1); term(r + 8, f2, g2); term(r + 12, f3, g3); }
        for (; r < r1; r += 4) term(r, f0, g0);
    }
    sh[0][ry][cx] = (f0 + f1) + (f2 + f3);
    sh[1][ry][cx] = (g0 + g1) + (g2 + g3);
    __syncthreads();
    if (ry == 0 && c < C) {
        partial[((size_t)blockIdx.y * 2 + 0) * C + c] = (sh[0][0][cx] + sh[0][1][cx]) + (sh[0][2][cx] + sh[0][3][cx]);
        partial[((size_t)blockIdx.y * 2 + 1) * C + c] = (sh[1][0][cx] + sh[1][1][cx]) + (sh[1][2][cx] + sh[1][3][cx]);
    }
}

// MODE 1 with four columns per thread (C % 4 == 0, 16-byte aligned operands): workgroup = 16 column quads x 16 row-lanes, four rows in flight
// per thread = eight 16-byte loads; the 4-byte form moves 2.4 TB/s on the two-operand backward sums, this one see DESIGN (round 4).  Same
// partial layout; the row-lanes are folded through LDS in a fixed order.
__global__ __launch_bounds__(256) void colstats_bwd4_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            int relu, int R, int C, int rows_per_block, float* __restrict__ partial,
                                                            const int32_t* __restrict__ live, int live_n) {
    __shared__ float4 sh[2][15][16];
    const int cq = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c = blockIdx.x * 64 + cq * 4;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
    float4 f[4], g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { f[u] = make_float4(0.f, 0.f, 0.f, 0.f); g[u] = make_float4(0.f, 0.f, 0.f, 0.f); }
    if (c < C) {
        const float4 sc = *reinterpret_cast<const float4*>(scale + c), sf = *reinterpret_cast<const float4*>(shift + c);
        const float4 mu = *reinterpret_cast<const float4*>(mean + c), rs = *reinterpret_cast<const float4*>(rstd + c);
        auto term = [&](int r, float4& ff, float4& gg) {
            if (live && !live[r / live_n]) return;                      // (exact zeros: skipped without changing a bit)
            const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)r * C + c);
            float4 d = *reinterpret_cast<const float4*>(dy + (size_t)r * C + c);
            if (relu) {
                if (!(xv.x * sc.x + sf.x > 0.f)) d.x = 0.f;
                if (!(xv.y * sc.y + sf.y > 0.f)) d.y = 0.f;
                if (!(xv.z * sc.z + sf.z > 0.f)) d.z = 0.f;
                if (!(xv.w * sc.w + sf.w > 0.f)) d.w = 0.f;
            }
            ff.x += d.x; ff.y += d.y; ff.z += d.z; ff.w += d.w;
            gg.x += d.x * ((xv.x - mu.x) * rs.x); gg.y += d.y * ((xv.y - mu.y) * rs.y);
            gg.z += d.z * ((xv.z - mu.z) * rs.z); gg.w += d.w * ((xv.w - mu.w) * rs.w);
        };
        int r = r0 + ry;
        for (; r + 48 < r1; r += 64) { term(r, f[0], g[0]); term(r + 16, f[1], g[1]); term(r + 32, f[2], g[2]); term(r + 48, f[3], g[3]); }
        for (; r < r1; r += 16) term(r, f[0], g[0]);
    }
    auto add4 = [](const float4& a, const float4& b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
    float4 fs = add4(add4(f[0], f[1]), add4(f[2], f[3])), gs = add4(add4(g[0], g[1]), add4(g[2], g[3]));
    if (ry > 0) { sh[0][ry - 1][cq] = fs; sh[1][ry - 1][cq] = gs; }
    __syncthreads();
    if (ry == 0 && c < C) {
        for (int l = 0; l < 15; ++l) { fs = add4(fs, sh[0][l][cq]); gs = add4(gs, sh[1][l][cq]); }
        *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.y * 2 + 0) * C + c) = fs;
        *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.y * 2 + 1) * C + c) = gs;
    }
}

// BatchNorm finalize: mean, biased var -> scale = gamma*rstd, shift = beta - mean*scale; running stats (momentum, unbiased var)
// fold the stage-1 partial rows [nparts][2][C]: workgroup = 64 columns x 4 part-lanes, 8 loads in flight per thread, lanes folded
// through LDS in a fixed order (a single thread per column walking ~1000 partial rows is a 150 us chain of dependent loads)
__device__ __forceinline__ void fold_partials(const float* __restrict__ partial, int nparts, int C, int c, int lane, float (*red)[3][64],
                                              float& s, float& q) {
    s = 0.f; q = 0.f;
    if (c < C) {
#pragma unroll 8
        for (int p = lane; p < nparts; p += 4) { s += partial[((size_t)p * 2) * C + c]; q += partial[((size_t)p * 2 + 1) * C + c]; }
    }
    const int cl = threadIdx.x & 63;
    if (lane > 0) { red[0][lane - 1][cl] = s; red[1][lane - 1][cl] = q; }
    __syncthreads();
    if (lane == 0) {
        s = (s + red[0][0][cl]) + (red[0][1][cl] + red[0][2][cl]);
        q = (q + red[1][0][cl]) + (red[1][1][cl] + red[1][2][cl]);
    }
}
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ x, const float* __restrict__ partial, int nparts, int C, int R, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float eps, float momentum, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                   float* __restrict__ scale_out, float* __restrict__ shift_out) {
    __shared__ float red[2][3][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), lane = threadIdx.x >> 6;
    float s, q;
    fold_partials(partial, nparts, C, c, lane, red, s, q);
    if (lane != 0 || c >= C) return;
    const float dm = s / (float)R;                      // mean of (x - pivot)
    const float mean = x[c] + dm;
    float var = q / (float)R - dm * dm;
    var = fmaxf(var, 0.f);
    const float rstd = rsqrtf(var + eps);
    const float sc = gamma[c] * rstd;
    mean_out[c] = mean; rstd_out[c] = rstd; scale_out[c] = sc; shift_out[c] = beta[c] - mean * sc;
    if (running_mean) {
        const float unb = R > 1 ? var * ((float)R / (float)(R - 1)) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
    }
}
// sums of stage 1 -> out0[c], out1[c]
__global__ __launch_bounds__(256) void colstats_stage2(const float* __restrict__ partial, int nparts, int C, float* __restrict__ out0, float* __restrict__ out1) {
    __shared__ float red[2][3][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), lane = threadIdx.x >> 6;
    float s, q;
    fold_partials(partial, nparts, C, c, lane, red, s, q);
    if (lane == 0 && c < C) { out0[c] = s; out1[c] = q; }
}

// y = relu?(x*scale[c] + shift[c])   (float4)
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, int relu, long long total4, int C4,
                                                         float* __restrict__ y) {
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
    const float4* __restrict__ s4 = reinterpret_cast<const float4*>(scale);
    const float4* __restrict__ h4 = reinterpret_cast<const float4*>(shift);
    float4* __restrict__ y4 = reinterpret_cast<float4*>(y);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        const float4 v = x4[i], s = s4[c], h = h4[c];
        float4 o; o.x = v.x * s.x + h.x; o.y = v.y * s.y + h.y; o.z = v.z * s.z + h.z; o.w = v.w * s.w + h.w;
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        y4[i] = o;
    }
}
// dx = scale[c] * (dyh - s1[c]/R - xhat * s2[c]/R),  dyh = relu ? dy*(x*scale+shift>0) : dy      (scale = gamma*rstd)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ s1, const float* __restrict__ s2, int relu,
                                                           float invR, long long total, int C, float* __restrict__ dx,
                                                           const int32_t* __restrict__ live = nullptr, int live_n = 1) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const float xv = x[i], sc = scale[c];
        float d = (live && !live[(i / C) / live_n]) ? 0.f : dy[i];    // (a dead group's dy is zero by contract and is not read)
        if (relu && !(xv * sc + shift[c] > 0.f)) d = 0.f;
        const float xh = (xv - mean[c]) * rstd[c];
        dx[i] = sc * (d - s1[c] * invR - xh * s2[c] * invR);
    }
}
// the same map four columns per thread (C % 4 == 0, 16-byte aligned tensors)
__global__ __launch_bounds__(256) void bn_bwd_apply4_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ s1, const float* __restrict__ s2, int relu,
                                                            float invR, long long total4, int C4, float* __restrict__ dx,
                                                            const int32_t* __restrict__ live, int live_n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const float4 xv = reinterpret_cast<const float4*>(x)[i], sc = reinterpret_cast<const float4*>(scale)[c4], sf = reinterpret_cast<const float4*>(shift)[c4];
        const float4 mu = reinterpret_cast<const float4*>(mean)[c4], rs = reinterpret_cast<const float4*>(rstd)[c4];
        const float4 a1 = reinterpret_cast<const float4*>(s1)[c4], a2 = reinterpret_cast<const float4*>(s2)[c4];
        float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!(live && !live[(i / C4) / live_n])) d = reinterpret_cast<const float4*>(dy)[i];
        if (relu) {
            if (!(xv.x * sc.x + sf.x > 0.f)) d.x = 0.f;
            if (!(xv.y * sc.y + sf.y > 0.f)) d.y = 0.f;
            if (!(xv.z * sc.z + sf.z > 0.f)) d.z = 0.f;
            if (!(xv.w * sc.w + sf.w > 0.f)) d.w = 0.f;
        }
        float4 o;
        o.x = sc.x * (d.x - a1.x * invR - ((xv.x - mu.x) * rs.x) * a2.x * invR);
        o.y = sc.y * (d.y - a1.y * invR - ((xv.y - mu.y) * rs.y) * a2.y * invR);
        o.z = sc.z * (d.z - a1.z * invR - ((xv.z - mu.z) * rs.z) * a2.z * invR);
        o.w = sc.w * (d.w - a1.w * invR - ((xv.w - mu.w) * rs.w) * a2.w * invR);
        reinterpret_cast<float4*>(dx)[i] = o;
    }
}

// ------------------------------------------------------------------------------------------------ group max-pool
// in [G*n, C] -> out [G, C] = max over the n rows of a group; arg [G,C] = first row attaining it (torch.max semantics)
__global__ __launch_bounds__(256) void group_max_kernel(const float* __restrict__ in, int n, int C, long long total,
                                                        float* __restrict__ out, int32_t* __restrict__ arg) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); const long long g = i / C;
        const float* __restrict__ p = in + (size_t)g * n * C + c;
        float best = p[0]; int bi = 0;
        for (int r = 1; r < n; ++r) { const float v = p[(size_t)r * C]; if (v > best) { best = v; bi = r; } }
        out[i] = best;
        if (arg) arg[i] = bi;
    }
}
// din[g*n + r, c] (+)= (arg[g,c] == r) ? dout[g,c] : 0
__global__ __launch_bounds__(256) void group_max_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ arg, int n,
                                                            int C, long long total, int accumulate, float* __restrict__ din) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); const long long row = i / C; const long long g = row / n; const int r = (int)(row % n);
        const float v = arg[g * C + c] == r ? dout[g * C + c] : 0.f;
        din[i] = accumulate ? din[i] + v : v;
    }
}
// out[g, c] = sum over the n rows of a group   (gradient of a per-group broadcast add)
__global__ __launch_bounds__(256) void group_sum_kernel(const float* __restrict__ in, int n, int C, long long total, float* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C); const long long g = i / C;
        const float* __restrict__ p = in + (size_t)g * n * C + c;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f; int r = 0;
        for (; r + 3 < n; r += 4) { a0 += p[(size_t)r * C]; a1 += p[(size_t)(r + 1) * C]; a2 += p[(size_t)(r + 2) * C]; a3 += p[(size_t)(r + 3) * C]; }
        for (; r < n; ++r) a0 += p[(size_t)r * C];
        out[i] = (a0 + a1) + (a2 + a3);
    }
}

static inline unsigned grid_for(long long total, int block) {
    long long g = (total + block - 1) / block; if (g > 8192) g = 8192; if (g < 1) g = 1; return (unsigned)g;
}
static int stats_parts(int R, int C) {
    const int cb = (C + 63) / 64;
    int parts = (2048 + cb - 1) / cb;
    const int maxp = (R + 63) / 64; if (parts > maxp) parts = maxp;
    if (parts > 1024) parts = 1024; if (parts < 1) parts = 1;
    return parts;
}
extern "C" size_t act_colstats_workspace(int R, int C) { return (size_t)stats_parts(R, C) * 2 * C * sizeof(float); }

// train-mode BatchNorm statistics of x [R,C] (models/dvae.py:189-200 nn.BatchNorm1d over B*G*n samples per channel):
// writes mean, rstd, scale = gamma*rstd, shift = beta - mean*scale and updates running stats in place (nullable).
extern "C" int act_bn_stats_f32(const float* x, int R, int C, const float* gamma, const float* beta, float eps, float momentum,
                                float* running_mean, float* running_var, float* mean, float* rstd, float* scale, float* shift,
                                float* workspace, size_t workspace_bytes, act_stream_t stream) {
    if (!x || !gamma || !beta || !mean || !rstd || !scale || !shift || !workspace) return ACT_E_NULLPTR;
    if (R <= 0 || C <= 0) return ACT_E_BADARG;
    const int parts = stats_parts(R, C);
    if (workspace_bytes < (size_t)parts * 2 * C * sizeof(float)) return ACT_E_BADARG;
    int rpb = (R + parts - 1) / parts; rpb = (rpb + 3) / 4 * 4;
    const int nparts = (R + rpb - 1) / rpb;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_BN_STATS, s, 0.0, 4.0 * R * (double)C);
    hipLaunchKernelGGL(colstats_stage1<0>, dim3((C + 63) / 64, nparts), dim3(256), 0, s, x, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
                       R, C, rpb, workspace);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(256), 0, s, x, workspace, nparts, C, R, gamma, beta, eps, momentum,
                       running_mean, running_var, mean, rstd, scale, shift);
    ACT_LAUNCH_CHECK(); return 0;
}

extern "C" int act_affine_act_f32(const float* x, const float* scale, const float* shift, int relu, int R, int C, float* y,
                                  act_stream_t stream) {
    if (!x || !scale || !shift || !y) return ACT_E_NULLPTR;
    if (R < 0 || C <= 0 || (C & 3)) return ACT_E_BADARG;
    const long long total4 = (long long)R * C / 4; if (total4 == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_BN_APPLY, s, 0.0, 8.0 * R * (double)C);
    hipLaunchKernelGGL(affine_act_kernel, dim3(grid_for(total4, 256)), dim3(256), 0, s, x, scale, shift, relu, total4, C / 4, y);
    ACT_LAUNCH_CHECK(); return 0;
}

// BatchNorm(+ReLU) backward: dy = gradient w.r.t. the (ReLU'd) output; x = BN input; -> dx, dgamma, dbeta
extern "C" int act_bn_bwd_f32(const float* x, const float* dy, const float* scale, const float* shift, const float* mean,
                              const float* rstd, int relu, int R, int C, float* dx, float* dgamma, float* dbeta,
                              float* workspace, size_t workspace_bytes, act_stream_t stream) {
    return act_bn_bwd_groups_f32(x, dy, scale, shift, mean, rstd, relu, R, C, nullptr, 1, dx, dgamma, dbeta, workspace, workspace_bytes, stream);
}
extern "C" int act_bn_bwd_groups_f32(const float* x, const float* dy, const float* scale, const float* shift, const float* mean,
                                     const float* rstd, int relu, int R, int C, const int32_t* live, int n, float* dx, float* dgamma, float* dbeta,
                                     float* workspace, size_t workspace_bytes, act_stream_t stream) {
    if (!x || !dy || !scale || !shift || !mean || !rstd || !dx || !dgamma || !dbeta || !workspace) return ACT_E_NULLPTR;
    if (R <= 0 || C <= 0 || (live && (n <= 0 || R % n))) return ACT_E_BADARG;
    const int parts = stats_parts(R, C);
    if (workspace_bytes < (size_t)parts * 2 * C * sizeof(float)) return ACT_E_BADARG;
    int rpb = (R + parts - 1) / parts; rpb = (rpb + 3) / 4 * 4;
    const int nparts = (R + rpb - 1) / rpb;
    hipStream_t s = (hipStream_t)stream;
    // compulsory bytes (round 6; the two-pass kernel re-reads x / dy, mostly from L2): x read once, dy of the LIVE rows read once, dx written once
    const double live_frac = live ? act_prof_live_fraction(live, R / n, s) : 1.0;
    ActProfScope ps(KID_BN_BWD, s, 0.0, 4.0 * R * (double)C * (2.0 + live_frac));
    // the kernel choice (and with it the fp32 summation order of dgamma / dbeta) depends on the SHAPE only: C % 4 == 0 takes the 16-byte
    // forms and requires 16-byte aligned operands (BADARG otherwise) instead of silently falling back to the scalar order
    const bool vec = (C % 4 == 0);
    if (vec && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)scale | (uintptr_t)shift | (uintptr_t)mean | (uintptr_t)rstd |
                 (uintptr_t)dbeta | (uintptr_t)dgamma | (uintptr_t)workspace) & 15) != 0) return ACT_E_BADARG;
    static const bool stats4 = [] { const char* e = getenv("ACT_BN_BWD_STATS4"); return !(e && e[0] == '0'); }();
    if (vec && stats4)
        hipLaunchKernelGGL(colstats_bwd4_kernel, dim3((C + 63) / 64, nparts), dim3(256), 0, s, x, dy, scale, shift, mean, rstd, relu, R, C, rpb, workspace,
                           live, n);
    else
        hipLaunchKernelGGL(colstats_stage1<1>, dim3((C + 63) / 64, nparts), dim3(256), 0, s, x, dy, scale, shift, mean, rstd, relu, R, C, rpb, workspace,
                           live, n);
    hipLaunchKernelGGL(colstats_stage2, dim3((C + 63) / 64), dim3(256), 0, s, workspace, nparts, C, dbeta, dgamma);
    const long long total = (long long)R * C;
    if (vec) hipLaunchKernelGGL(bn_bwd_apply4_kernel, dim3(grid_for(total / 4, 256)), dim3(256), 0, s, x, dy, scale, shift, mean, rstd, dbeta, dgamma, relu,
                                1.0f / (float)R, total / 4, C / 4, dx, live, n);
    else     hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, x, dy, scale, shift, mean, rstd, dbeta, dgamma, relu,
                                1.0f / (float)R, total, C, dx, live, n);
    ACT_LAUNCH_CHECK(); return 0;
}

extern "C" int act_group_max_f32(const float* in, int G, int n, int C, float* out, int32_t* arg, act_stream_t stream) {
    if (!in || !out) return ACT_E_NULLPTR;
    if (G < 0 || n <= 0 || C <= 0) return ACT_E_BADARG;
    const long long total = (long long)G * C; if (total == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_MAXPOOL, s, 0.0, 4.0 * G * (double)C * (n + 2));
    hipLaunchKernelGGL(group_max_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, in, n, C, total, out, arg);
    ACT_LAUNCH_CHECK(); return 0;
}
extern "C" int act_group_max_bwd_f32(const float* dout, const int32_t* arg, int G, int n, int C, int accumulate, float* din,
                                     act_stream_t stream) {
    if (!dout || !arg || !din) return ACT_E_NULLPTR;
    if (G < 0 || n <= 0 || C <= 0) return ACT_E_BADARG;
    const long long total = (long long)G * n * C; if (total == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_MAXPOOL_BWD, s, 0.0, 4.0 * G * (double)C * (n * (accumulate ? 2 : 1) + 2));
    hipLaunchKernelGGL(group_max_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, dout, arg, n, C, total, accumulate, din);
    ACT_LAUNCH_CHECK(); return 0;
}
extern "C" int act_group_sum_f32(const float* in, int G, int n, int C, float* out, act_stream_t stream) {
    if (!in || !out) return ACT_E_NULLPTR;
    if (G < 0 || n <= 0 || C <= 0) return ACT_E_BADARG;
    const long long total = (long long)G * C; if (total == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_ELTWISE, s, 0.0, 4.0 * G * (double)C * (n + 1));
    hipLaunchKernelGGL(group_sum_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, in, n, C, total, out);
    ACT_LAUNCH_CHECK(); return 0;
}

// eval-mode BatchNorm as an affine map (nn.BatchNorm1d.eval(): running statistics)
__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rm,
                                      const float* __restrict__ rv, float eps, int C, float* __restrict__ scale, float* __restrict__ shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] * rsqrtf(rv[c] + eps);
    scale[c] = sc; shift[c] = beta[c] - rm[c] * sc;
}
extern "C" int act_bn_eval_affine_f32(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                                      int C, float* scale, float* shift, act_stream_t stream) {
    if (!gamma || !beta || !running_mean || !running_var || !scale || !shift) return ACT_E_NULLPTR;
    if (C <= 0) return ACT_E_BADARG;
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean, running_var, eps, C,
                       scale, shift);
    ACT_LAUNCH_CHECK(); return 0;
}

// ---- pieces of the BatchNorm reductions as separate entry points: SyncBatchNorm (statistics all-reduced across ranks between them) ----
__global__ __launch_bounds__(256) void meanvar_finalize_kernel(const float* __restrict__ x, const float* __restrict__ partial, int nparts, int C, int R,
                                                               float* __restrict__ mean_out, float* __restrict__ var_out) {
    __shared__ float red[2][3][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), lane = threadIdx.x >> 6;
    float s, q;
    fold_partials(partial, nparts, C, c, lane, red, s, q);
    if (lane != 0 || c >= C) return;
    const float dm = s / (float)R;
    mean_out[c] = x[c] + dm;
    var_out[c] = fmaxf(q / (float)R - dm * dm, 0.f);           // biased variance of this rank's rows
}
// per-column mean and biased variance of x [R,C] (pivoted sums, deterministic)
extern "C" int act_col_mean_var_f32(const float* x, int R, int C, float* mean, float* var, float* workspace, size_t workspace_bytes,
                                    act_stream_t stream) {
    if (!x || !mean || !var || !workspace) return ACT_E_NULLPTR;
    if (R <= 0 || C <= 0) return ACT_E_BADARG;
    const int parts = stats_parts(R, C);
    if (workspace_bytes < (size_t)parts * 2 * C * sizeof(float)) return ACT_E_BADARG;
    int rpb = (R + parts - 1) / parts; rpb = (rpb + 3) / 4 * 4;
    const int nparts = (R + rpb - 1) / rpb;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_BN_STATS, s, 0.0, 4.0 * R * (double)C);
    hipLaunchKernelGGL(colstats_stage1<0>, dim3((C + 63) / 64, nparts), dim3(256), 0, s, x, nullptr, nullptr, nullptr, nullptr, nullptr, 0,
                       R, C, rpb, workspace);
    hipLaunchKernelGGL(meanvar_finalize_kernel, dim3((C + 63) / 64), dim3(256), 0, s, x, workspace, nparts, C, R, mean, var);
    ACT_LAUNCH_CHECK(); return 0;
}
// backward sums of this rank's rows: sum_dy[c] = sum_r dyh, sum_dy_xhat[c] = sum_r dyh * xhat   (dyh = relu ? dy * (x*scale+shift > 0) : dy)
extern "C" int act_bn_bwd_sums_f32(const float* x, const float* dy, const float* scale, const float* shift, const float* mean, const float* rstd,
                                   int relu, int R, int C, float* sum_dy, float* sum_dy_xhat, float* workspace, size_t workspace_bytes,
                                   act_stream_t stream) {
    if (!x || !dy || !scale || !shift || !mean || !rstd || !sum_dy || !sum_dy_xhat || !workspace) return ACT_E_NULLPTR;
    if (R <= 0 || C <= 0) return ACT_E_BADARG;
    const int parts = stats_parts(R, C);
    if (workspace_bytes < (size_t)parts * 2 * C * sizeof(float)) return ACT_E_BADARG;
    int rpb = (R + parts - 1) / parts; rpb = (rpb + 3) / 4 * 4;
    const int nparts = (R + rpb - 1) / rpb;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_BN_BWD, s, 0.0, 8.0 * R * (double)C);
    hipLaunchKernelGGL(colstats_stage1<1>, dim3((C + 63) / 64, nparts), dim3(256), 0, s, x, dy, scale, shift, mean, rstd, relu, R, C, rpb, workspace);
    hipLaunchKernelGGL(colstats_stage2, dim3((C + 63) / 64), dim3(256), 0, s, workspace, nparts, C, sum_dy, sum_dy_xhat);
    ACT_LAUNCH_CHECK(); return 0;
}
// dx = scale * (dyh - sum_dy / count - xhat * sum_dy_xhat / count) with the (all-reduced) sums over `count` rows of all ranks
extern "C" int act_bn_bwd_apply_f32(const float* x, const float* dy, const float* scale, const float* shift, const float* mean, const float* rstd,
                                    const float* sum_dy, const float* sum_dy_xhat, float count, int relu, int R, int C, float* dx,
                                    act_stream_t stream) {
    if (!x || !dy || !scale || !shift || !mean || !rstd || !sum_dy || !sum_dy_xhat || !dx) return ACT_E_NULLPTR;
    if (R <= 0 || C <= 0 || !(count > 0.f)) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_BN_BWD, s, 0.0, 12.0 * R * (double)C);
    const long long total = (long long)R * C;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, x, dy, scale, shift, mean, rstd, sum_dy, sum_dy_xhat, relu,
                       1.0f / count, total, C, dx);
    ACT_LAUNCH_CHECK(); return 0;
}

// BatchNorm statistics from the per-tile column (mean, M2) partials a GEMM epilogue left behind (act_sgemm_fx_f32, tile_stats
// [tiles][2][C], every tile `rows_per_tile` rows): mean = avg of tile means, M2 = sum M2_t + rows_per_tile * sum (mean_t - mean)^2
// (Chan's combination for equal counts), folded in a fixed order (deterministic) -- then exactly what bn_finalize_kernel produces.
// workgroup = 16 columns x 64 tile-lanes (1,024 threads), C / 16 workgroups: a thread folds tiles / 64 partials per pass, the 64 lanes of a
// column are combined by a fixed binary tree in LDS (8 workgroups of 4 tile-lanes walked 512 partials per thread: 79 us for 2,048 tiles)
__global__ __launch_bounds__(1024) void bn_tiles_finalize_kernel(const float* __restrict__ ts, int tiles, int rows_per_tile, int C,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                 float momentum, float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                 float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                                 float* __restrict__ scale_out, float* __restrict__ shift_out) {
    __shared__ float red[64][16];
    const int cl = threadIdx.x & 15, lane = threadIdx.x >> 4, c = blockIdx.x * 16 + cl;
    auto fold = [&](float v) -> float {                    // sum over the 64 tile-lanes of column cl, same value in every lane afterwards
        __syncthreads();
        red[lane][cl] = v;
        __syncthreads();
        for (int off = 32; off > 0; off >>= 1) {
            if (lane < off) red[lane][cl] += red[lane + off][cl];
            __syncthreads();
        }
        return red[0][cl];
    };
    float s = 0.f;
    if (c < C) {
#pragma unroll 4
        for (int t = lane; t < tiles; t += 64) s += ts[((size_t)t * 2) * C + c];
    }
    const float mean = fold(s) / (float)tiles;
    float q = 0.f;
    if (c < C) {
#pragma unroll 4
        for (int t = lane; t < tiles; t += 64) {
            const float d = ts[((size_t)t * 2) * C + c] - mean;
            q += ts[((size_t)t * 2 + 1) * C + c] + (float)rows_per_tile * d * d;
        }
    }
    const float qs = fold(q);
    if (lane != 0 || c >= C) return;
    const float R = (float)tiles * (float)rows_per_tile;
    const float var = fmaxf(qs / R, 0.f);
    const float rstd = rsqrtf(var + eps);
    const float sc = gamma[c] * rstd;
    mean_out[c] = mean; rstd_out[c] = rstd; scale_out[c] = sc; shift_out[c] = beta[c] - mean * sc;
    if (running_mean) {
        const float unb = R > 1.f ? var * (R / (R - 1.f)) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * unb;
    }
}
extern "C" int act_bn_tiles_finalize_f32(const float* tile_stats, int tiles, int rows_per_tile, int C, const float* gamma, const float* beta, float eps,
                                         float momentum, float* running_mean, float* running_var, float* mean, float* rstd, float* scale,
                                         float* shift, act_stream_t stream) {
    if (!tile_stats || !gamma || !beta || !mean || !rstd || !scale || !shift) return ACT_E_NULLPTR;
    if (tiles <= 0 || rows_per_tile <= 0 || C <= 0) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_BN_STATS, s, 0.0, 8.0 * tiles * (double)C);
    hipLaunchKernelGGL(bn_tiles_finalize_kernel, dim3((C + 15) / 16), dim3(1024), 0, s, tile_stats, tiles, rows_per_tile, C, gamma, beta, eps, momentum,
                       running_mean, running_var, mean, rstd, scale, shift);
    ACT_LAUNCH_CHECK(); return 0;
}
