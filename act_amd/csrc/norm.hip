// norm.hip -- row-wise fused kernels around the GEMMs of a Transformer block (gfx950):
//   * pos-add + LayerNorm forward   (models/act.py:87-90,109-112: x = blk(x + pos); blk: x + attn(norm1(x)) ...)
//   * LayerNorm backward (dx fused with the residual-stream gradient; dgamma/dbeta two-stage, deterministic)
//   * column sums (bias gradients)
//   * cosine distillation loss forward / backward (models/act.py:1243-1254, lightly NegativeCosineSimilarity)
// One wave per row: a row of D=384/768 floats sits in registers (float4 per lane), statistics come from DPP
// wave reductions, nothing is staged through LDS.  All kernels are HBM-bound: one read + one write per element.
#include "common.h"
#include <stdlib.h>

#define LN_MAXV 8          // float4 per lane: D <= 64*4*8 = 2048

// (hi, lo) bf16 planes of four consecutive values (round to nearest even; lo = bf16(x - hi)): the operand form of the opt-in split-bf16 GEMM
// (gemm_bf16x3.hip) -- same rounding as its split_bf16x2_kernel, so a producer that writes planes is bit-identical to producing fp32 and splitting it
__device__ __forceinline__ unsigned ln_bf16_rne(float x) { unsigned u = __float_as_uint(x); u += 0x7FFFu + ((u >> 16) & 1u); return u >> 16; }
__device__ __forceinline__ void store_planes4(unsigned short* __restrict__ hi, unsigned short* __restrict__ lo, size_t i4, const float4& o) {
    const float e[4] = {o.x, o.y, o.z, o.w};
    unsigned h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { h[j] = ln_bf16_rne(e[j]); l[j] = ln_bf16_rne(e[j] - __uint_as_float(h[j] << 16)); }
    reinterpret_cast<uint2*>(hi)[i4] = make_uint2(h[0] | (h[1] << 16), h[2] | (h[3] << 16));
    reinterpret_cast<uint2*>(lo)[i4] = make_uint2(l[0] | (l[1] << 16), l[2] | (l[3] << 16));
}

// y = LN(x + pos) * gamma + beta ; optionally stores xin = x + pos (needed for the residual) ; mean/rstd for backward.
// y_hi / y_lo (optional): y also (or, with y == NULL, only) as bf16 planes.
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ pos,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ xin_out, float* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int T, int D, float eps, unsigned short* __restrict__ y_hi = nullptr,
                                                            unsigned short* __restrict__ y_lo = nullptr) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const int nv = D >> 2;                                // float4 per row
    const float4* __restrict__ xr = reinterpret_cast<const float4*>(x + (size_t)row * D);
    const float4* __restrict__ pr = pos ? reinterpret_cast<const float4*>(pos + (size_t)row * D) : nullptr;
    float4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            float4 a = xr[c];
            if (pr) { const float4 b = pr[c]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
            v[i] = a;
            s += (a.x + a.y) + (a.z + a.w);
        } else v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float mean = wave_sum_f32(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float var = wave_sum_f32(q) / (float)D;
    const float rstd = rsqrtf(var + eps);
    float4* __restrict__ yr = reinterpret_cast<float4*>(y + (size_t)row * D);
    float4* __restrict__ xo = xin_out ? reinterpret_cast<float4*>(xin_out + (size_t)row * D) : nullptr;
    const float4* __restrict__ g4 = reinterpret_cast<const float4*>(gamma);
    const float4* __restrict__ b4 = reinterpret_cast<const float4*>(beta);
    // (round 6: batching this kernel's loads the way layernorm_bwd_kernel does costs it two waves per SIMD of occupancy -- 117 VGPRs -- and was 5 % SLOWER: with one
    //  row per wave and eight waves per SIMD the chunk-by-chunk round trips are already covered)
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const float4 g = g4[c], b = b4[c];
            float4 o;
            o.x = (v[i].x - mean) * rstd * g.x + b.x; o.y = (v[i].y - mean) * rstd * g.y + b.y;
            o.z = (v[i].z - mean) * rstd * g.z + b.z; o.w = (v[i].w - mean) * rstd * g.w + b.w;
            if (y) yr[c] = o;
            if (y_hi) store_planes4(y_hi, y_lo, (size_t)row * nv + c, o);
            if (xo) xo[c] = v[i];
        }
    }
    if (lane == 0) { if (mean_out) mean_out[row] = mean; if (rstd_out) rstd_out[row] = rstd; }
}

// dx = dres + rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat)) ; per-block partial dgamma/dbeta
// NW waves per workgroup, one row per wave and pass: with the 16-row workgroups of the small launches (T <= 16K rows) NW = 16 gives every row its own wave --
// the 4-wave form walked four rows one after the other, each a load -> two wave reductions -> store round trip (round 5)
// MAXV = float4 per lane and row (D <= 256 * MAXV): a template parameter so that the 1,024-thread form keeps its row in 128 registers
template <int NW, int MAXV>
__global__ __launch_bounds__(64 * NW) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xin,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ dres,
                                                            float* __restrict__ dx, float* __restrict__ part_dg,
                                                            float* __restrict__ part_db, int T, int D, int rows_per_block) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = D >> 2;
    const float4* __restrict__ g4 = reinterpret_cast<const float4*>(gamma);
    float4 ag[MAXV], ab[MAXV];                   // this lane's dgamma / dbeta accumulators
#pragma unroll
    for (int i = 0; i < MAXV; ++i) { ag[i] = make_float4(0.f, 0.f, 0.f, 0.f); ab[i] = ag[i]; }
    const int r0 = blockIdx.x * rows_per_block;
    const int r1 = min(T, r0 + rows_per_block);
    // Round 6: every load of a row is issued UNCONDITIONALLY and up front (lanes beyond the row read its last float4 and are masked in the arithmetic).  With the
    // loads inside `if (c < nv)` each 64-lane chunk became an exec-masked block -- load, s_waitcnt vmcnt(0), compute, branch -- i.e. MAXV serialised memory round
    // trips per row, and as many again for the residual gradient; gamma is loop-invariant and now loaded once per workgroup.  Same arithmetic per element (results can
    // differ from the previous build in the last bit where hipcc's FMA contraction falls differently: Stage-I loss 0.8377311 vs 0.8377298 after 14 steps).
    int cc[MAXV]; float4 gv[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) { cc[i] = min(lane + 64 * i, nv - 1); gv[i] = g4[cc[i]]; }
    for (int row = r0 + wave; row < r1; row += NW) {
        const float4* __restrict__ dyr = reinterpret_cast<const float4*>(dy + (size_t)row * D);
        const float4* __restrict__ xr = reinterpret_cast<const float4*>(xin + (size_t)row * D);
        const float4* __restrict__ drr = dres ? reinterpret_cast<const float4*>(dres + (size_t)row * D) : nullptr;
        float4 dv[MAXV], xv[MAXV], rv[MAXV];
#pragma unroll
        for (int i = 0; i < MAXV; ++i) { dv[i] = dyr[cc[i]]; xv[i] = xr[cc[i]]; }
        if (drr) {
#pragma unroll
            for (int i = 0; i < MAXV; ++i) rv[i] = drr[cc[i]];
        }
        const float mu = mean[row], rs = rstd[row];
        float4 h[MAXV], w[MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                const float4 d = dv[i], x = xv[i], g = gv[i];
                float4 xh; xh.x = (x.x - mu) * rs; xh.y = (x.y - mu) * rs; xh.z = (x.z - mu) * rs; xh.w = (x.w - mu) * rs;
                float4 dg; dg.x = d.x * g.x; dg.y = d.y * g.y; dg.z = d.z * g.z; dg.w = d.w * g.w;
                h[i] = xh; w[i] = dg;
                s1 += (dg.x + dg.y) + (dg.z + dg.w);
                s2 += (dg.x * xh.x + dg.y * xh.y) + (dg.z * xh.z + dg.w * xh.w);
                ag[i].x += d.x * xh.x; ag[i].y += d.y * xh.y; ag[i].z += d.z * xh.z; ag[i].w += d.w * xh.w;
                ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
            }
        }
        const float m1 = wave_sum_f32(s1) / (float)D, m2 = wave_sum_f32(s2) / (float)D;
        float4* __restrict__ dxr = reinterpret_cast<float4*>(dx + (size_t)row * D);
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nv) {
                float4 o;
                o.x = rs * (w[i].x - m1 - h[i].x * m2); o.y = rs * (w[i].y - m1 - h[i].y * m2);
                o.z = rs * (w[i].z - m1 - h[i].z * m2); o.w = rs * (w[i].w - m1 - h[i].w * m2);
                if (drr) { const float4 r = rv[i]; o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w; }
                dxr[c] = o;
            }
        }
    }
    // combine the NW waves' partials in LDS (fixed order: groups of four, then ascending) -> one partial row per workgroup: [gridDim.x][D]
    if (part_dg) {
        extern __shared__ __attribute__((aligned(16))) float sacc[];      // [2][NW][D]
        float4* sg = reinterpret_cast<float4*>(sacc) + (size_t)wave * nv;
        float4* sb = reinterpret_cast<float4*>(sacc) + (size_t)(NW + wave) * nv;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) { const int c = lane + 64 * i; if (c < nv) { sg[c] = ag[i]; sb[c] = ab[i]; } }
        __syncthreads();
        const float* fg = sacc; const float* fb = sacc + (size_t)NW * D;
        for (int c = threadIdx.x; c < D; c += 64 * NW) {
            float tg = 0.f, tb = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w += 4) {
                const float qg = (fg[(size_t)w * D + c] + fg[(size_t)(w + 1) * D + c]) + (fg[(size_t)(w + 2) * D + c] + fg[(size_t)(w + 3) * D + c]);
                const float qb = (fb[(size_t)w * D + c] + fb[(size_t)(w + 1) * D + c]) + (fb[(size_t)(w + 2) * D + c] + fb[(size_t)(w + 3) * D + c]);
                tg = w ? tg + qg : qg; tb = w ? tb + qb : qb;
            }
            part_dg[(size_t)blockIdx.x * D + c] = tg;
            part_db[(size_t)blockIdx.x * D + c] = tb;
        }
    }
}

// out[c] (+)= sum_r in[r][c]   -- two-stage, fixed order (deterministic).  stage 1: partial[blk][c]
// block = 64 columns x 4 row-lanes; every thread keeps 8 independent loads in flight.
__global__ __launch_bounds__(256) void colsum_stage1(const float* __restrict__ in, int R, int C, int ld, int rows_per_block,
                                                     float* __restrict__ partial) {
    __shared__ float sh[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
    if (c < C) {
        const float* __restrict__ p = in + c;
        int r = r0 + ry;
        for (; r + 28 < r1; r += 32) {
            a0 += p[(size_t)r * ld];        a1 += p[(size_t)(r + 4) * ld];  a2 += p[(size_t)(r + 8) * ld];  a3 += p[(size_t)(r + 12) * ld];
            a4 += p[(size_t)(r + 16) * ld]; a5 += p[(size_t)(r + 20) * ld]; a6 += p[(size_t)(r + 24) * ld]; a7 += p[(size_t)(r + 28) * ld];
        }
        for (; r < r1; r += 4) a0 += p[(size_t)r * ld];
    }
    sh[ry][cx] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    __syncthreads();
    if (ry == 0 && c < C) partial[(size_t)blockIdx.y * C + c] = (sh[0][cx] + sh[1][cx]) + (sh[2][cx] + sh[3][cx]);
}
__global__ __launch_bounds__(256) void colsum_stage2_pair(const float* __restrict__ partial0, const float* __restrict__ partial1,
                                                          int nparts, int C, float* __restrict__ out0, float* __restrict__ out1,
                                                          int accumulate) {
    __shared__ float sh[4][64];
    const float* __restrict__ partial = blockIdx.y ? partial1 : partial0;
    float* __restrict__ out = blockIdx.y ? out1 : out0;
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    // eight loads in flight per thread (round 5: the fold of the student's 112 LayerNorm partial rows was a chain of 14 dependent L2 round trips, 8 us per
    // launch, 30 launches per Stage-II step); fixed order -> deterministic
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f, a5 = 0.f, a6 = 0.f, a7 = 0.f;
    if (c < C) {
        const float* __restrict__ q = partial + c;
        int p = ry;
        for (; p + 28 < nparts; p += 32) {
            a0 += q[(size_t)p * C];        a1 += q[(size_t)(p + 4) * C];  a2 += q[(size_t)(p + 8) * C];  a3 += q[(size_t)(p + 12) * C];
            a4 += q[(size_t)(p + 16) * C]; a5 += q[(size_t)(p + 20) * C]; a6 += q[(size_t)(p + 24) * C]; a7 += q[(size_t)(p + 28) * C];
        }
        for (; p + 4 < nparts; p += 8) { a0 += q[(size_t)p * C]; a1 += q[(size_t)(p + 4) * C]; }
        for (; p < nparts; p += 4) a0 += q[(size_t)p * C];
    }
    sh[ry][cx] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
    __syncthreads();
    if (ry == 0 && c < C) {
        const float acc = (sh[0][cx] + sh[1][cx]) + (sh[2][cx] + sh[3][cx]);
        out[c] = accumulate ? out[c] + acc : acc;
    }
}
__global__ __launch_bounds__(256) void colsum_stage2(const float* __restrict__ partial, int nparts, int C, float* __restrict__ out,
                                                     int accumulate) {
    __shared__ float sh[4][64];
    const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    float a0 = 0.f, a1 = 0.f;
    if (c < C) {
        int p = ry;
        for (; p + 4 < nparts; p += 8) { a0 += partial[(size_t)p * C + c]; a1 += partial[(size_t)(p + 4) * C + c]; }
        for (; p < nparts; p += 4) a0 += partial[(size_t)p * C + c];
    }
    sh[ry][cx] = a0 + a1;
    __syncthreads();
    if (ry == 0 && c < C) {
        const float acc = (sh[0][cx] + sh[1][cx]) + (sh[2][cx] + sh[3][cx]);
        out[c] = accumulate ? out[c] + acc : acc;
    }
}

// cosine distillation loss: rows [R,D] ; loss = mean_r (1 - cos(s_r, t_r)), cos with torch's eps semantics
// (x.y / max(|x|*|y|, eps)).  Stores per-row dot/norms for the backward.
__global__ __launch_bounds__(256) void cosine_fwd_kernel(const float* __restrict__ s, const float* __restrict__ t, int R, int D,
                                                         float eps, float* __restrict__ row_loss, float* __restrict__ stats) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    float dot = 0.f, ss = 0.f, tt = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float a = s[(size_t)row * D + c], b = t[(size_t)row * D + c];
        dot += a * b; ss += a * a; tt += b * b;
    }
    dot = wave_sum_f32(dot); ss = wave_sum_f32(ss); tt = wave_sum_f32(tt);
    const float denom = fmaxf(sqrtf(ss * tt), eps);
    if (lane == 0) {
        row_loss[row] = 1.0f - dot / denom;
        stats[row * 3 + 0] = dot; stats[row * 3 + 1] = ss; stats[row * 3 + 2] = tt;
    }
}
// d loss / d s_r = -gscale * ( t/denom - dot * s * tt / denom^3 )   (denom = sqrt(ss*tt), clamped region has zero 2nd term)
__global__ __launch_bounds__(256) void cosine_bwd_kernel(const float* __restrict__ s, const float* __restrict__ t,
                                                         const float* __restrict__ stats, const float* __restrict__ gout,
                                                         int R, int D, float eps, float inv_rows, float* __restrict__ ds) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float dot = stats[row * 3 + 0], ss = stats[row * 3 + 1], tt = stats[row * 3 + 2];
    const float nrm = sqrtf(ss * tt);
    const float g = -gout[0] * inv_rows;
    float ca, cb;
    if (nrm > eps) { ca = 1.0f / nrm; cb = dot * tt / (nrm * nrm * nrm); }
    else { ca = 1.0f / eps; cb = 0.f; }
    for (int c = lane; c < D; c += 64) {
        const float a = s[(size_t)row * D + c], b = t[(size_t)row * D + c];
        ds[(size_t)row * D + c] = g * (b * ca - a * cb);
    }
}
// mean of a vector (deterministic, single block) -> out[0]
__global__ __launch_bounds__(1024) void mean_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
    __shared__ float sh[16];
    float acc = 0.f;
    const int per = (n + 1023) / 1024, k0 = threadIdx.x * per, k1 = min(n, k0 + per);
    for (int k = k0; k < k1; ++k) acc += v[k];
    acc = wave_sum_f32(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { float tot = 0.f; for (int w = 0; w < 16; ++w) tot += sh[w]; out[0] = tot / (float)n; }
}

extern "C" int act_layernorm_fwd_f32(const float* x, const float* pos, const float* gamma, const float* beta, float* xin_out,
                                     float* y, float* mean, float* rstd, int T, int D, float eps, act_stream_t stream) {
    if (!x || !gamma || !beta || !y) return ACT_E_NULLPTR;
    if (T < 0 || D <= 0 || (D & 3) || D > 64 * 4 * LN_MAXV) return ACT_E_BADARG;
    if (T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_LAYERNORM_FWD, s, 0.0, 4.0 * T * (double)D * (2 + (pos ? 1 : 0) + (xin_out ? 1 : 0)));
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((T + 3) / 4), dim3(256), 0, s, x, pos, gamma, beta, xin_out, y, mean, rstd, T, D, eps, nullptr, nullptr);
    ACT_LAUNCH_CHECK(); return 0;
}
extern "C" int act_layernorm_fwd_planes_f32(const float* x, const float* pos, const float* gamma, const float* beta, float* xin_out, float* y,
                                            uint16_t* y_hi, uint16_t* y_lo, float* mean, float* rstd, int T, int D, float eps, act_stream_t stream) {
    if (!x || !gamma || !beta || !y_hi || !y_lo) return ACT_E_NULLPTR;
    if (T < 0 || D <= 0 || (D & 3) || D > 64 * 4 * LN_MAXV || (((uintptr_t)y_hi | (uintptr_t)y_lo) & 7)) return ACT_E_BADARG;
    if (T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_LAYERNORM_FWD, s, 0.0, 4.0 * T * (double)D * (2 + (pos ? 1 : 0) + (xin_out ? 1 : 0)));
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((T + 3) / 4), dim3(256), 0, s, x, pos, gamma, beta, xin_out, y, mean, rstd, T, D, eps, y_hi, y_lo);
    ACT_LAUNCH_CHECK(); return 0;
}

// Prompt rows of the prompt-tuned Transformer (models/dvae.py:485-498,556-566), frozen-teacher form: for every cloud b and prompt p
//   v = dropout(tok[p,:]) + ppos[p,:]   (inverted dropout, keep mask from Philox keyed by (seed, b*P+p, c/4))   ->  LN(v) * gamma + beta
// one launch instead of expand + dropout + add + LayerNorm, and the [B*P, D] intermediate is never written.
__global__ __launch_bounds__(256) void prompt_layernorm_fwd_kernel(const float* __restrict__ tok, const float* __restrict__ ppos, int P,
                                                                   float drop_p, uint64_t seed, const uint64_t* __restrict__ seed_dev,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, float* __restrict__ y, int T, int D,
                                                                   float eps, unsigned short* __restrict__ y_hi = nullptr,
                                                                   unsigned short* __restrict__ y_lo = nullptr) {
    if (seed_dev) seed ^= seed_dev[0] * 0x9E3779B97F4A7C15ull;      // device-resident step counter (replayable from a hipGraph)
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const int nv = D >> 2, pr = row % P;
    const float4* __restrict__ xr = reinterpret_cast<const float4*>(tok + (size_t)pr * D);
    const float4* __restrict__ qr = reinterpret_cast<const float4*>(ppos + (size_t)pr * D);
    const float inv_keep = 1.0f / (1.0f - drop_p);
    const uint32_t thr = (uint32_t)(drop_p * 16777216.0f);              // drop when the top 24 random bits < thr
    float4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            float4 a = xr[c];
            if (drop_p > 0.f) {
                uint32_t r[4];
                philox4x32_10((uint32_t)c, (uint32_t)row, 1u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
                a.x = (r[0] >> 8) < thr ? 0.f : a.x * inv_keep; a.y = (r[1] >> 8) < thr ? 0.f : a.y * inv_keep;
                a.z = (r[2] >> 8) < thr ? 0.f : a.z * inv_keep; a.w = (r[3] >> 8) < thr ? 0.f : a.w * inv_keep;
            }
            const float4 b = qr[c];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            v[i] = a;
            s += (a.x + a.y) + (a.z + a.w);
        } else v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float mean = wave_sum_f32(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rstd = rsqrtf(wave_sum_f32(q) / (float)D + eps);
    float4* __restrict__ yr = reinterpret_cast<float4*>(y + (size_t)row * D);
    const float4* __restrict__ g4 = reinterpret_cast<const float4*>(gamma);
    const float4* __restrict__ b4 = reinterpret_cast<const float4*>(beta);
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nv) {
            const float4 g = g4[c], b = b4[c];
            float4 o;
            o.x = (v[i].x - mean) * rstd * g.x + b.x; o.y = (v[i].y - mean) * rstd * g.y + b.y;
            o.z = (v[i].z - mean) * rstd * g.z + b.z; o.w = (v[i].w - mean) * rstd * g.w + b.w;
            if (y) yr[c] = o;
            if (y_hi) store_planes4(y_hi, y_lo, (size_t)row * nv + c, o);
        }
    }
}

extern "C" int act_prompt_layernorm_fwd_planes_f32(const float* tok, const float* ppos, int B, int P, int D, float drop_p, uint64_t seed,
                                                   const uint64_t* seed_dev, const float* gamma, const float* beta, float eps, uint16_t* y_hi,
                                                   uint16_t* y_lo, act_stream_t stream) {
    if (!tok || !ppos || !gamma || !beta || !y_hi || !y_lo) return ACT_E_NULLPTR;
    if (B < 0 || P <= 0 || D <= 0 || (D & 3) || D > 64 * 4 * LN_MAXV || drop_p < 0.f || drop_p >= 1.f || (((uintptr_t)y_hi | (uintptr_t)y_lo) & 7)) return ACT_E_BADARG;
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int T = B * P;
    ActProfScope ps(KID_LAYERNORM_FWD, s, 0.0, 4.0 * T * (double)D);
    hipLaunchKernelGGL(prompt_layernorm_fwd_kernel, dim3((T + 3) / 4), dim3(256), 0, s, tok, ppos, P, drop_p, seed, seed_dev, gamma, beta,
                       (float*)nullptr, T, D, eps, y_hi, y_lo);
    ACT_LAUNCH_CHECK(); return 0;
}

extern "C" int act_prompt_layernorm_fwd_f32(const float* tok, const float* ppos, int B, int P, int D, float drop_p, uint64_t seed,
                                            const uint64_t* seed_dev, const float* gamma, const float* beta, float eps, float* y,
                                            act_stream_t stream) {
    if (!tok || !ppos || !gamma || !beta || !y) return ACT_E_NULLPTR;
    if (B < 0 || P <= 0 || D <= 0 || (D & 3) || D > 64 * 4 * LN_MAXV || drop_p < 0.f || drop_p >= 1.f) return ACT_E_BADARG;
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int T = B * P;
    ActProfScope ps(KID_LAYERNORM_FWD, s, 0.0, 4.0 * T * (double)D);
    hipLaunchKernelGGL(prompt_layernorm_fwd_kernel, dim3((T + 3) / 4), dim3(256), 0, s, tok, ppos, P, drop_p, seed, seed_dev, gamma, beta, y, T, D, eps,
                       nullptr, nullptr);
    ACT_LAUNCH_CHECK(); return 0;
}

// Prompt rows of a TRAINED prompt layer (Stage I, models/dvae.py:485-498,556-566): y[b*P+p,:] = dropout(tok[p,:]) + ppos[p,:] for every cloud b
// (one launch instead of expand + dropout + add), and its backward: dppos[p,:] = sum_b dy[b*P+p,:], dtok[p,:] = sum_b dy * keep / (1 - drop_p)
// in a fixed order over b.  keep: the given 0/1 mask [B*P, D] (parity tests inject the reference's draws) or Philox keyed like
// prompt_layernorm_fwd_kernel by (seed, row, column/4), regenerated in the backward.
__global__ __launch_bounds__(256) void prompt_rows_fwd_kernel(const float* __restrict__ tok, const float* __restrict__ ppos,
                                                              const float* __restrict__ mask, int P, int D, float drop_p, uint64_t seed,
                                                              float* __restrict__ y, long long total4) {
    const int nv = D >> 2;
    const float inv_keep = 1.0f / (1.0f - drop_p);
    const uint32_t thr = (uint32_t)(drop_p * 16777216.0f);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % nv); const long long row = i / nv; const int pr = (int)(row % P);
        float4 a = reinterpret_cast<const float4*>(tok + (size_t)pr * D)[c];
        if (mask) {
            const float4 m = reinterpret_cast<const float4*>(mask)[i];
            a.x *= m.x * inv_keep; a.y *= m.y * inv_keep; a.z *= m.z * inv_keep; a.w *= m.w * inv_keep;
        } else if (drop_p > 0.f) {
            uint32_t r[4];
            philox4x32_10((uint32_t)c, (uint32_t)row, 1u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
            a.x = (r[0] >> 8) < thr ? 0.f : a.x * inv_keep; a.y = (r[1] >> 8) < thr ? 0.f : a.y * inv_keep;
            a.z = (r[2] >> 8) < thr ? 0.f : a.z * inv_keep; a.w = (r[3] >> 8) < thr ? 0.f : a.w * inv_keep;
        }
        const float4 b = reinterpret_cast<const float4*>(ppos + (size_t)pr * D)[c];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        reinterpret_cast<float4*>(y)[i] = a;
    }
}
// one thread per (prompt row p, float4 column c): walks the B clouds in order (4 loads in flight)
__global__ __launch_bounds__(256) void prompt_rows_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ mask, int B, int P, int D,
                                                              float drop_p, uint64_t seed, float* __restrict__ dtok, float* __restrict__ dppos) {
    const int nv = D >> 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P * nv) return;
    const int c = i % nv, pr = i / nv;
    const float inv_keep = 1.0f / (1.0f - drop_p);
    const uint32_t thr = (uint32_t)(drop_p * 16777216.0f);
    float4 st = make_float4(0.f, 0.f, 0.f, 0.f), sp = st;
#pragma unroll 4
    for (int b = 0; b < B; ++b) {
        const long long row = (long long)b * P + pr;
        const float4 g = reinterpret_cast<const float4*>(dy + (size_t)row * D)[c];
        sp.x += g.x; sp.y += g.y; sp.z += g.z; sp.w += g.w;
        float4 k = make_float4(inv_keep, inv_keep, inv_keep, inv_keep);
        if (mask) {
            const float4 m = reinterpret_cast<const float4*>(mask + (size_t)row * D)[c];
            k.x *= m.x; k.y *= m.y; k.z *= m.z; k.w *= m.w;
        } else if (drop_p > 0.f) {
            uint32_t r[4];
            philox4x32_10((uint32_t)c, (uint32_t)row, 1u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
            k.x = (r[0] >> 8) < thr ? 0.f : inv_keep; k.y = (r[1] >> 8) < thr ? 0.f : inv_keep;
            k.z = (r[2] >> 8) < thr ? 0.f : inv_keep; k.w = (r[3] >> 8) < thr ? 0.f : inv_keep;
        }
        st.x += g.x * k.x; st.y += g.y * k.y; st.z += g.z * k.z; st.w += g.w * k.w;
    }
    reinterpret_cast<float4*>(dtok + (size_t)pr * D)[c] = st;
    reinterpret_cast<float4*>(dppos + (size_t)pr * D)[c] = sp;
}
extern "C" int act_prompt_rows_fwd_f32(const float* tok, const float* ppos, const float* mask, int B, int P, int D, float drop_p, uint64_t seed,
                                       float* y, act_stream_t stream) {
    if (!tok || !ppos || !y) return ACT_E_NULLPTR;
    if (B < 0 || P <= 0 || D <= 0 || (D & 3) || drop_p < 0.f || drop_p >= 1.f) return ACT_E_BADARG;
    if (B == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const long long total4 = (long long)B * P * (D >> 2);
    long long g = (total4 + 255) / 256; if (g > 8192) g = 8192;
    ActProfScope ps(KID_ELTWISE, s, 0.0, 4.0 * B * P * (double)D * (mask ? 2 : 1));
    hipLaunchKernelGGL(prompt_rows_fwd_kernel, dim3((unsigned)g), dim3(256), 0, s, tok, ppos, mask, P, D, drop_p, seed, y, total4);
    ACT_LAUNCH_CHECK(); return 0;
}
extern "C" int act_prompt_rows_bwd_f32(const float* dy, const float* mask, int B, int P, int D, float drop_p, uint64_t seed, float* dtok,
                                       float* dppos, act_stream_t stream) {
    if (!dy || !dtok || !dppos) return ACT_E_NULLPTR;
    if (B <= 0 || P <= 0 || D <= 0 || (D & 3) || drop_p < 0.f || drop_p >= 1.f) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const int n = P * (D >> 2);
    ActProfScope ps(KID_ELTWISE, s, 0.0, 4.0 * B * P * (double)D * (mask ? 2 : 1));
    hipLaunchKernelGGL(prompt_rows_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dy, mask, B, P, D, drop_p, seed, dtok, dppos);
    ACT_LAUNCH_CHECK(); return 0;
}

// rows per 4-wave workgroup of the LayerNorm backward (a wave walks its rows one after the other: every row is a load -> reduce -> store round trip of
// ~2 us, so 16 rows per workgroup made the 1,792-row launches of the student a 4-round-trip chain on 112 of 256 CUs).  ACT_LN_BWD_RPB overrides (A/B).
static int ln_rows_per_block(int T) {
    static const int env = [] { const char* e = getenv("ACT_LN_BWD_RPB"); return e ? atoi(e) : 0; }();
    const int floor_ = env >= 4 ? (env + 3) / 4 * 4 : 16;
    int r = (T + 1023) / 1024; r = (r + 3) / 4 * 4; return r < floor_ ? floor_ : r;
}
extern "C" size_t act_layernorm_bwd_workspace(int T, int D) {
    const int rpb = ln_rows_per_block(T); const int nblk = (T + rpb - 1) / rpb;
    return (size_t)nblk * D * 2 * sizeof(float);
}

extern "C" int act_layernorm_bwd_f32(const float* dy, const float* xin, const float* gamma, const float* mean, const float* rstd,
                                     const float* dres, float* dx, float* dgamma, float* dbeta, int accumulate_params,
                                     float* workspace, size_t workspace_bytes, int T, int D, act_stream_t stream) {
    if (!dy || !xin || !gamma || !mean || !rstd || !dx) return ACT_E_NULLPTR;
    if (T < 0 || D <= 0 || (D & 3) || D > 64 * 4 * LN_MAXV) return ACT_E_BADARG;
    if (T == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const int rpb = ln_rows_per_block(T); const int nblk = (T + rpb - 1) / rpb;
    const bool params = dgamma && dbeta;
    if (params && (!workspace || workspace_bytes < act_layernorm_bwd_workspace(T, D))) return ACT_E_BADARG;
    ActProfScope ps(KID_LAYERNORM_BWD, s, 0.0, 4.0 * T * (double)D * (3 + (dres ? 1 : 0)));
    float* pg = params ? workspace : nullptr;
    float* pb = params ? workspace + (size_t)nblk * D : nullptr;
    // waves per workgroup: 4.  ACT_LN_BWD_WAVES = 8 / 16 (A/B): one row per wave shortens the kernel (Stage II: 0.54 -> 0.40 ms per step with the streams
    // serialised) but the overlapped step does not move (26.19 / 26.21 vs 26.09 / 26.16 ms): a 1,024-thread workgroup needs a CU's wave slots all at once
    // and queues behind the teacher's GEMM workgroups of the other stream.
    static const int nw_env = [] { const char* e = getenv("ACT_LN_BWD_WAVES"); return e ? atoi(e) : 0; }();
    int nw = 4;
    if (nw_env == 4 || nw_env == 8 || nw_env == 16) nw = nw_env;
    if (nw > rpb) nw = rpb >= 8 ? 8 : 4;
    if (D > 1024) nw = 4; else if (D > 512 && nw > 8) nw = 8;
    const size_t lds = params ? (size_t)2 * nw * D * sizeof(float) : 0;
#define LNB(NW_, MV_) hipLaunchKernelGGL((layernorm_bwd_kernel<NW_, MV_>), dim3(nblk), dim3(64 * NW_), lds, s, dy, xin, gamma, mean, rstd, dres, dx, pg, pb, T, D, rpb)
    if (D <= 512)       { if (nw == 16) LNB(16, 2); else if (nw == 8) LNB(8, 2); else LNB(4, 2); }
    else if (D <= 1024) { if (nw == 8) LNB(8, 4); else LNB(4, 4); }
    else                LNB(4, 8);
#undef LNB
    ACT_LAUNCH_CHECK();
    if (params) {
        hipLaunchKernelGGL(colsum_stage2_pair, dim3((D + 63) / 64, 2), dim3(256), 0, s, pg, pb, nblk, D, dgamma, dbeta, accumulate_params);
        ACT_LAUNCH_CHECK();
    }
    return 0;
}

static int colsum_parts(int R, int C) {
    const int cb = (C + 63) / 64;
    int parts = (2048 + cb - 1) / cb;                       // ~2048 workgroups in total
    const int maxp = (R + 31) / 32; if (parts > maxp) parts = maxp;
    if (parts > 1024) parts = 1024; if (parts < 1) parts = 1;
    return parts;
}
extern "C" size_t act_colsum_workspace(int R, int C) { return (size_t)colsum_parts(R, C) * C * sizeof(float); }

extern "C" int act_colsum_f32(const float* in, int R, int C, int ld, float* out, int accumulate, float* workspace,
                              size_t workspace_bytes, act_stream_t stream) {
    if (!in || !out || !workspace) return ACT_E_NULLPTR;
    if (R < 0 || C <= 0) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const int parts = colsum_parts(R, C);
    if (workspace_bytes < (size_t)parts * C * sizeof(float)) return ACT_E_BADARG;
    int rpb = (R + parts - 1) / parts; rpb = (rpb + 3) / 4 * 4; if (rpb < 4) rpb = 4;
    const int nparts = (R + rpb - 1) / rpb > 0 ? (R + rpb - 1) / rpb : 1;
    ActProfScope ps(KID_COLSUM, s, 0.0, 4.0 * R * (double)C);
    hipLaunchKernelGGL(colsum_stage1, dim3((C + 63) / 64, nparts), dim3(256), 0, s, in, R, C, ld, rpb, workspace);
    hipLaunchKernelGGL(colsum_stage2, dim3((C + 63) / 64), dim3(256), 0, s, workspace, nparts, C, out, accumulate);
    ACT_LAUNCH_CHECK(); return 0;
}

// softmax cross-entropy (nn.CrossEntropyLoss, mean): one wave per row
__global__ __launch_bounds__(256) void xent_fwd_kernel(const float* __restrict__ z, const int64_t* __restrict__ lab, int R, int C,
                                                       float* __restrict__ row_buf) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float* zr = z + (size_t)row * C;
    float m = -INFINITY; int am = 0x7fffffff;
    for (int c = lane; c < C; c += 64) { const float v = zr[c]; if (v > m) { m = v; am = c; } }
    const float mx = wave_max_f32(m, -INFINITY);
    int cand = (m == mx) ? am : 0x7fffffff;                         // lowest index attaining the max (torch argmax)
    cand = -wave_max_i32(-cand, -0x7fffffff);
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += expf(zr[c] - mx);
    se = wave_sum_f32(se);
    if (lane == 0) {
        const float lse = mx + logf(se);
        const int64_t l = lab[row];
        row_buf[row] = lse;
        row_buf[R + row] = lse - zr[l];
        row_buf[2 * R + row] = (cand == (int)l) ? 1.0f : 0.0f;
    }
}
__global__ __launch_bounds__(256) void xent_bwd_kernel(const float* __restrict__ z, const int64_t* __restrict__ lab,
                                                       const float* __restrict__ row_buf, const float* __restrict__ gout,
                                                       int R, int C, float inv_rows, float* __restrict__ dz) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)R * C) return;
    const int row = (int)(i / C), c = (int)(i % C);
    const float p = expf(z[i] - row_buf[row]);
    dz[i] = gout[0] * inv_rows * (p - ((int64_t)c == lab[row] ? 1.0f : 0.0f));
}

extern "C" int act_softmax_xent_fwd_f32(const float* logits, const int64_t* labels, int R, int C, float* loss_out,
                                        float* row_buf, float* acc_out, act_stream_t stream) {
    if (!logits || !labels || !loss_out || !row_buf) return ACT_E_NULLPTR;
    if (R <= 0 || C <= 0) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_COSINE_FWD, s, 0.0, 4.0 * R * (double)C);
    hipLaunchKernelGGL(xent_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, s, logits, labels, R, C, row_buf);
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, s, row_buf + R, R, loss_out);
    if (acc_out) hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, s, row_buf + 2 * (size_t)R, R, acc_out);
    ACT_LAUNCH_CHECK(); return 0;
}

extern "C" int act_softmax_xent_bwd_f32(const float* logits, const int64_t* labels, const float* row_buf, const float* grad_loss,
                                        int R, int C, float* grad_logits, act_stream_t stream) {
    if (!logits || !labels || !row_buf || !grad_loss || !grad_logits) return ACT_E_NULLPTR;
    if (R <= 0 || C <= 0) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_COSINE_BWD, s, 0.0, 8.0 * R * (double)C);
    const long long total = (long long)R * C;
    hipLaunchKernelGGL(xent_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, logits, labels, row_buf, grad_loss,
                       R, C, 1.0f / (float)R, grad_logits);
    ACT_LAUNCH_CHECK(); return 0;
}

// Alternative distillation losses of the reference (models/act.py:1186-1191,1255): 'l2' = nn.MSELoss(mean), 'smoothl1' =
// nn.SmoothL1Loss(mean, beta = 1).  Per-row partial sums (one wave per row, fixed order) -> mean_kernel; deterministic.
__global__ __launch_bounds__(256) void regression_loss_fwd_kernel(const float* __restrict__ s, const float* __restrict__ t, int R, int D,
                                                                  int kind, float* __restrict__ row_loss) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    float acc = 0.f;
    for (int c = lane; c < D; c += 64) {
        const float d = s[(size_t)row * D + c] - t[(size_t)row * D + c];
        const float ad = fabsf(d);
        acc += kind == 0 ? d * d : (ad < 1.f ? 0.5f * d * d : ad - 0.5f);
    }
    acc = wave_sum_f32(acc);
    if (lane == 0) row_loss[row] = acc / (float)D;
}
__global__ __launch_bounds__(256) void regression_loss_bwd_kernel(const float* __restrict__ s, const float* __restrict__ t,
                                                                  const float* __restrict__ gout, long long n, int kind, float inv_n,
                                                                  float* __restrict__ ds) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float d = s[i] - t[i];
        ds[i] = gout[0] * inv_n * (kind == 0 ? 2.f * d : fminf(fmaxf(d, -1.f), 1.f));
    }
}
extern "C" int act_regression_loss_fwd_f32(const float* student, const float* teacher, int R, int D, int kind, float* loss_out,
                                           float* row_loss, act_stream_t stream) {
    if (!student || !teacher || !loss_out || !row_loss) return ACT_E_NULLPTR;
    if (R <= 0 || D <= 0 || (kind != 0 && kind != 1)) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_COSINE_FWD, s, 0.0, 8.0 * R * (double)D);
    hipLaunchKernelGGL(regression_loss_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, s, student, teacher, R, D, kind, row_loss);
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, s, row_loss, R, loss_out);
    ACT_LAUNCH_CHECK(); return 0;
}
extern "C" int act_regression_loss_bwd_f32(const float* student, const float* teacher, const float* grad_loss, int R, int D, int kind,
                                           float* grad_student, act_stream_t stream) {
    if (!student || !teacher || !grad_loss || !grad_student) return ACT_E_NULLPTR;
    if (R <= 0 || D <= 0 || (kind != 0 && kind != 1)) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_COSINE_BWD, s, 0.0, 12.0 * R * (double)D);
    const long long n = (long long)R * D;
    long long g = (n + 255) / 256; if (g > 8192) g = 8192;
    hipLaunchKernelGGL(regression_loss_bwd_kernel, dim3((unsigned)g), dim3(256), 0, s, student, teacher, grad_loss, n, kind,
                       1.0f / (float)n, grad_student);
    ACT_LAUNCH_CHECK(); return 0;
}

extern "C" int act_cosine_loss_fwd_f32(const float* student, const float* teacher, int R, int D, float eps, float* loss_out,
                                       float* row_loss, float* stats, act_stream_t stream) {
    if (!student || !teacher || !loss_out || !row_loss || !stats) return ACT_E_NULLPTR;
    if (R <= 0 || D <= 0) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_COSINE_FWD, s, 0.0, 8.0 * R * (double)D);
    hipLaunchKernelGGL(cosine_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, s, student, teacher, R, D, eps, row_loss, stats);
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(1024), 0, s, row_loss, R, loss_out);
    ACT_LAUNCH_CHECK(); return 0;
}

extern "C" int act_cosine_loss_bwd_f32(const float* student, const float* teacher, const float* stats, const float* grad_loss,
                                       int R, int D, float eps, float* grad_student, act_stream_t stream) {
    if (!student || !teacher || !stats || !grad_loss || !grad_student) return ACT_E_NULLPTR;
    if (R <= 0 || D <= 0) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_COSINE_BWD, s, 0.0, 12.0 * R * (double)D);
    hipLaunchKernelGGL(cosine_bwd_kernel, dim3((R + 3) / 4), dim3(256), 0, s, student, teacher, stats, grad_loss, R, D, eps,
                       1.0f / (float)R, grad_student);
    ACT_LAUNCH_CHECK(); return 0;
}

// y[r,:] = x[r,:] * gate[r / rows_per_scale]   (DropPath gate on a gradient; float4 stream)
__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ x, const float* __restrict__ gate, long long total4, int D4,
                                                         int rows_per_scale, float* __restrict__ y) {
    const float4* __restrict__ x4 = reinterpret_cast<const float4*>(x);
    float4* __restrict__ y4 = reinterpret_cast<float4*>(y);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const float g = gate[(i / D4) / rows_per_scale];
        float4 v = x4[i];
        v.x *= g; v.y *= g; v.z *= g; v.w *= g;
        y4[i] = v;
    }
}
extern "C" int act_scale_rows_f32(const float* x, const float* gate, int T, int D, int rows_per_scale, float* y, act_stream_t stream) {
    if (!x || !gate || !y) return ACT_E_NULLPTR;
    if (T < 0 || D <= 0 || (D & 3) || rows_per_scale <= 0) return ACT_E_BADARG;
    const long long total4 = (long long)T * D / 4; if (total4 == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_ELTWISE, s, 0.0, 8.0 * T * (double)D);
    long long g = (total4 + 255) / 256; if (g > 8192) g = 8192;
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)g), dim3(256), 0, s, x, gate, total4, D / 4, rows_per_scale, y);
    ACT_LAUNCH_CHECK(); return 0;
}

// out = a + b (float4 stream): the accumulation of a gradient that several consumers of one tensor produce (act_block_stack_bwd_f32)
// (no __restrict__: the C ABI allows out to alias a or b -- the stack backward accumulates in place)
__global__ __launch_bounds__(256) void add_kernel(const float* a, const float* b, long long total4, float* out) {
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    float4* o4 = reinterpret_cast<float4*>(out);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const float4 u = a4[i], v = b4[i];
        o4[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
}
extern "C" int act_add_f32(const float* a, const float* b, float* out, long long n, act_stream_t stream) {
    if (!a || !b || !out) return ACT_E_NULLPTR;
    if (n < 0 || (n & 3) || (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15)) return ACT_E_BADARG;
    if (n == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    ActProfScope ps(KID_ELTWISE, s, 0.0, 12.0 * (double)n);
    long long g = (n / 4 + 255) / 256; if (g > 8192) g = 8192;
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)g), dim3(256), 0, s, a, b, n / 4, out);
    ACT_LAUNCH_CHECK(); return 0;
}
