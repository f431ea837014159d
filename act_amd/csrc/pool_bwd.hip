// pool_bwd.hip -- the two products of the mini-PointNet backward whose left operand is the gradient of a max-pool (gfx950).
//
// Encoder.forward ends with  feature_global = torch.max(feature, dim=2)  over the n points of each group (models/dvae.py:209-216, :262-275), so
// the gradient that reaches the last conv,  dh[g*n + j][c] = (arg[g][c] == j ? dout[g][c] : 0),  has exactly ONE non-zero per (group, channel):
// of the n * C entries of a group only C are live.  The dense GEMMs that consume dh (gemm_q16 with the scatter generated on load) spend
// n = 32 .. 64 times the arithmetic the products need -- 2 x 0.84 ms of the Stage-II step, 2 x 6.5 ms at C5.  These kernels walk the live
// entries instead; both are bound by the bytes of the DENSE operand (da written once, the activated conv input read once), not by flops.
//
//   da[g*n + j][:] = sum over {c : arg[g][c] == j} of dout[g][c] * W[c][:]                       (pool_bwd_dx_kernel)
//   dW[c][:]       = sum over g of dout[g][c] * act(X[g*n + arg[g][c]][:])                       (pool_bwd_dw2_kernel)
//
// Summation order is fixed by the shapes and the set of live groups alone (ascending c inside a row; ascending g inside a split, splits folded
// in order), so the results are run-to-run and process-to-process identical; they differ from the dense path's MFMA summation order in the
// last bits.  Groups whose gradient row is entirely zero -- in Stage II the 51 masked of 64 patches per cloud, whose tokens the student never
// reads (models/act.py:269-275) -- are skipped: the weight gradient walks a device-built list of live groups, the row walk returns at once
// (writing zeros, or nothing when the consumer takes the same liveness flags: act_bn_bwd_groups_f32).
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
                                                          int ldw, int n, int C, float* __restrict__ out, int ldo, int skip_zero, const int32_t* __restrict__ glive) {
    using V = typename VecT<VEC>::T;
    extern __shared__ unsigned char smem[];
    int* s_arg = reinterpret_cast<int*>(smem);                                   // [C]
    float* s_d = reinterpret_cast<float*>(s_arg + C);                            // [C]
    int* s_start = reinterpret_cast<int*>(s_d + C);                              // [n + 1]
    int* s_wsum = s_start + n + 1;                                               // [4]
    unsigned short* s_ord = reinterpret_cast<unsigned short*>(s_wsum + 4);       // [C]
    const int tid = threadIdx.x, g = blockIdx.x;
    if (glive && !glive[g]) return;                                              // (the caller knows this group is dead and never reads its rows)
    int live = 0;
    for (int c = tid; c < C; c += 256) { const float d = dout[(size_t)g * C + c]; s_arg[c] = arg[(size_t)g * C + c]; s_d[c] = d; live |= (d != 0.f); }
    const int any_live = __syncthreads_or(live);                                 // (also the barrier between the LDS fill and the bucket passes)
    if (skip_zero && !any_live) {                                               // a group whose gradient is all zero (a masked patch): n zero rows
        V z; vzero(z);
        for (int r = 0; r < n; ++r) *reinterpret_cast<V*>(out + ((size_t)g * n + r) * ldo + (size_t)tid * VEC) = z;
        return;
    }
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

// The same walk with 16-byte loads: a row of N floats takes N / 4 threads, so the workgroup splits into P = 1024 / N parts that take the rows
// r = p (mod P) -- buckets are laid out part-major so every part walks one contiguous run of the sorted entries.  (Half the load instructions of
// the float2 form at N = 512: 469 -> 334 us at the Stage-II geometry, 1643 -> 1146 us at C5.)
template <int P>
__global__ __launch_bounds__(256) void pool_bwd_dx4_kernel(const float* __restrict__ dout, const int32_t* __restrict__ arg, const float* __restrict__ W,
                                                           int ldw, int n, int C, float* __restrict__ out, int ldo, int skip_zero, const int32_t* __restrict__ glive) {
    extern __shared__ unsigned char smem[];
    int* s_arg = reinterpret_cast<int*>(smem);                                   // [C]
    float* s_d = reinterpret_cast<float*>(s_arg + C);                            // [C]
    int* s_start = reinterpret_cast<int*>(s_d + C);                              // [n + 1]  (bucket order)
    int* s_wsum = s_start + n + 1;                                               // [4]
    unsigned short* s_ord = reinterpret_cast<unsigned short*>(s_wsum + 4);       // [C]
    const int tid = threadIdx.x, g = blockIdx.x;
    if (glive && !glive[g]) return;                                              // (the caller knows this group is dead and never reads its rows)
    int live = 0;
    for (int c = tid; c < C; c += 256) { const float d = dout[(size_t)g * C + c]; s_arg[c] = arg[(size_t)g * C + c]; s_d[c] = d; live |= (d != 0.f); }
    const int any_live = __syncthreads_or(live);                                 // (also the barrier between the LDS fill and the bucket passes)
    if (skip_zero && !any_live) {                                               // a group whose gradient is all zero (a masked patch): n zero rows
        constexpr int TPZ = 256 / P;
        const int part = tid / TPZ, q = tid - part * TPZ;
        for (int r = part; r < n; r += P) *reinterpret_cast<float4*>(out + ((size_t)g * n + r) * ldo + (size_t)q * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int npp = n / P;                                                       // rows (buckets) per part
    const int S = 256 / n, b_ = tid / S, sg = tid - b_ * S, L = (C + S - 1) / S, c_lo = sg * L, c_hi = min(C, c_lo + L);
    const int r_ = (b_ % npp) * P + b_ / npp;                                    // bucket b holds row (b % npp) * P + b / npp
    int cnt = 0;
    for (int c = c_lo; c < c_hi; ++c) cnt += (s_arg[c] == r_);
    int incl = cnt;
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) s_wsum[wv] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; ++w) base += s_wsum[w];
    int p = base + incl - cnt;
    if (sg == 0) s_start[b_] = p;
    if (tid == 255) s_start[n] = base + incl;
    for (int c = c_lo; c < c_hi; ++c) if (s_arg[c] == r_) s_ord[p++] = (unsigned short)c;
    __syncthreads();

    constexpr int TP = 256 / P;
    const int part = tid / TP, q = tid - part * TP;                              // (wave-uniform: TP >= 64)
    const int b_end = (part + 1) * npp;
    int b = part * npp;
    const int j_end = s_start[b_end];
    const float* __restrict__ Wc = W + (size_t)q * 4;
    float* __restrict__ obase = out + (size_t)g * n * ldo + (size_t)q * 4;
    constexpr int U = 8;
    int nb = s_start[b + 1];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    auto flush = [&]() {
        *reinterpret_cast<float4*>(obase + (size_t)((b - part * npp) * P + part) * ldo) = acc;
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
        ++b;
    };
    for (int j0 = s_start[b]; j0 < j_end; j0 += U) {
        float4 w[U]; float dv[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int c = s_ord[min(j0 + k, j_end - 1)];
            dv[k] = s_d[c];
            w[k] = *reinterpret_cast<const float4*>(Wc + (size_t)c * ldw);
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int j = j0 + k;
            if (j < j_end) {                                                     // (wave-uniform)
                while (j == nb) { flush(); nb = s_start[b + 1]; }
                vfma(acc, dv[k], w[k]);
            }
        }
    }
    while (b < b_end) flush();
}

// ------------------------------------------------------------------------------------------------------------------- dW = dh^T . act(X)
// Channel-per-lane form of the same product: workgroup = (64 * NW channels, 64 columns, range of groups), one wave per 64 channels with 64
// accumulators per lane.  Each lane loads ITS (arg, dout) pairs with plain coalesced loads, turns arg into the offset of its slice row, and
// reads that row two columns at a time with immediate offsets: per (group, channel, 2 columns) one ds_read_b64 and two FMAs -- no broadcast
// reads, no per-column address arithmetic.  The slice rows are 68 dwords apart: the compiler pairs the adjacent 8-byte reads into
// ds_read2_b64, which the LDS serves like a 16-byte access -- 16 lanes per cycle, four consecutive banks per lane -- so a pitch of 4 (mod 64)
// dwords puts rows r and r' on disjoint bank quads unless r = r' (mod 16).  Measured at the Stage-II geometry, every group live, random arg
// rows: pitch 66 / 70 / 74 / 98: 315 us, 72: 252, 68: 215 (rows 0..31 in lane order: 255, 216, 195).  NW = C / 64 up to 6, so the slice of
// X is read once (C <= 384) or C / 384 times.  One step = GB consecutive groups (GB * n contiguous rows of X): the next step's rows are in flight in registers while
// this step's GB groups are consumed from LDS, one barrier per step -- a memory round trip (~3 us under load) is amortised over GB groups
// instead of paid per group.  The accumulators leave through an LDS transpose: the partial tile is written in 128-byte runs.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int NW, int GB, bool AFF>
__global__ __launch_bounds__(64 * NW, 2) void pool_bwd_dw2_kernel(const float* __restrict__ dout, const int32_t* __restrict__ arg, const float* __restrict__ X,
                                                                  int ldx, const float* __restrict__ scale, const float* __restrict__ shift, int n, int C,
                                                                  int G, int gps, float* __restrict__ part, int ldp, size_t split_stride, int nslices,
                                                                  int xcd_map, const int* __restrict__ live, const int* __restrict__ nlive, int splits) {
    constexpr int NT = 64 * NW, KC = 64, STRIDE = KC + 4;                        // row pitch 68 dwords: the bank note above
    constexpr int MAXF = (GB * 64 * 16 / (GB == 4 ? 2 : 1) + NT - 1) / NT;      // GB = 4: n <= 32;  GB = 2: n <= 64
    extern __shared__ float4 smem4[];
    float* tile = reinterpret_cast<float*>(smem4);                               // [2][GB * n][STRIDE]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // blockIdx.x -> (column slice, split): workgroups are dealt round-robin to the 8 XCDs, so the slices of ONE group range are given ids that
    // land on the same XCD next to each other in time
    int bslice, bsplit;
    if (xcd_map) { const int x = blockIdx.x & 7, t = blockIdx.x >> 3; bslice = t % nslices; bsplit = (t / nslices) * 8 + x; }
    else         { bslice = blockIdx.x % nslices; bsplit = blockIdx.x / nslices; }
    const int n0 = bslice * KC, ch = blockIdx.y * NT + tid;
    // positions [g0, g1) of the list of live groups (groups with a non-zero gradient row, ascending; pool_bwd_compact_kernel) -- or of 0 .. G - 1
    const int total = live ? *nlive : G;
    if (live) gps = (total + splits - 1) / splits;
    const int g0 = min(total, bsplit * gps), g1 = min(total, g0 + gps);
    const int nshift = __ffs(n) - 1;                                             // n is a power of two (256 % n == 0)
    const int rows = GB * n, nf4 = rows * 16, buf_floats = rows * STRIDE;
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);       // column quad (tid & 15) of every float4 this thread stages
    if (AFF) { sc4 = *reinterpret_cast<const float4*>(scale + n0 + (tid & 15) * 4); sh4 = *reinterpret_cast<const float4*>(shift + n0 + (tid & 15) * 4); }
    float4 stg[MAXF]; int na[GB]; float nd[GB];
    auto fetch = [&](int g) {                                                    // the step at list positions g .. g + GB - 1 (clamped to the range)
        int gid[GB];
#pragma unroll
        for (int k = 0; k < GB; ++k) { const int pos = min(g + k, g1 - 1); gid[k] = live ? live[pos] : pos; }
#pragma unroll
        for (int i = 0; i < MAXF; ++i) {
            const int f = tid + NT * i;
            if (f < nf4) {
                const int rr = f >> 4, k = rr >> nshift;
                int gk = gid[0];
#pragma unroll
                for (int j = 1; j < GB; ++j) gk = (k == j) ? gid[j] : gk;
                stg[i] = *reinterpret_cast<const float4*>(X + ((size_t)gk * n + (rr & (n - 1))) * ldx + n0 + (tid & 15) * 4);
            }
        }
#pragma unroll
        for (int k = 0; k < GB; ++k) { na[k] = arg[(size_t)gid[k] * C + ch]; nd[k] = dout[(size_t)gid[k] * C + ch]; }
    };
    auto put = [&](int buf) {
        float* b = tile + (size_t)buf * buf_floats;
#pragma unroll
        for (int i = 0; i < MAXF; ++i) {
            const int f = tid + NT * i;
            if (f < nf4) {
                float4 v = stg[i];
                if (AFF) {
                    v.x = fmaxf(v.x * sc4.x + sh4.x, 0.f); v.y = fmaxf(v.y * sc4.y + sh4.y, 0.f);
                    v.z = fmaxf(v.z * sc4.z + sh4.z, 0.f); v.w = fmaxf(v.w * sc4.w + sh4.w, 0.f);
                }
                *reinterpret_cast<float4*>(b + (f >> 4) * STRIDE + (f & 15) * 4) = v;    // (16-byte aligned: STRIDE % 4 == 0)
            }
        }
    };
    float acc[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) acc[i] = 0.f;
    if (g0 < g1) { fetch(g0); put(0); }
    __syncthreads();
    int buf = 0;
    for (int g = g0; g < g1; g += GB) {
        int row[GB]; float dv[GB];
#pragma unroll
        for (int k = 0; k < GB; ++k) {                                           // this step's pairs -> slice-row offsets (groups past the end: weight 0)
            const bool ok = (unsigned)na[k] < (unsigned)n && g + k < g1;
            row[k] = (k * n + (ok ? na[k] : 0)) * STRIDE;
            dv[k] = ok ? nd[k] : 0.f;
        }
        if (g + GB < g1) fetch(g + GB);
        const float* bp = tile + (size_t)buf * buf_floats;
#pragma unroll
        for (int k = 0; k < GB; ++k) {
            const float* rp = bp + row[k];
            const f32x2 dv2 = {dv[k], dv[k]};
#pragma unroll
            for (int c = 0; c < 64; c += 2) {
                const f32x2 x = *reinterpret_cast<const f32x2*>(rp + c);
                f32x2 a2 = {acc[c], acc[c + 1]};
                a2 = __builtin_elementwise_fma(dv2, x, a2);                      // v_pk_fma_f32: two columns per VALU issue
                acc[c] = a2.x; acc[c + 1] = a2.y;
                if ((c & 15) == 14) {                                            // 16 columns in flight at a time (registers: the scheduler otherwise hoists all 32 reads)
                    float* q = acc + c - 14;
                    asm volatile("" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7]), "+v"(q[8]),
                                 "+v"(q[9]), "+v"(q[10]), "+v"(q[11]), "+v"(q[12]), "+v"(q[13]), "+v"(q[14]), "+v"(q[15]) : : "memory");
                }
            }
        }
        if (g + GB < g1) put(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // transpose through LDS (per wave: 64 channels x 32 columns at a time, rows 33 dwords apart) and store rows
    float* tw = tile + wv * (64 * 33);
    float* o = part + (size_t)bsplit * split_stride + (size_t)(blockIdx.y * NT + wv * 64) * ldp + n0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int k = 0; k < 32; ++k) tw[lane * 33 + k] = acc[32 * h + k];
        __builtin_amdgcn_s_waitcnt(0xc07f);                                      // lgkmcnt(0): the wave's own LDS writes have landed
        for (int i = 0; i < 64; i += 2) {
            const int r = i + (lane >> 5), c = lane & 31;
            o[(size_t)r * ldp + 32 * h + c] = tw[r * 33 + c];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
    }
}

// Live groups: in Stage II the encoder runs on all B * G patches but only the visible 20 % feed the loss (models/act.py:269-275 selects
// x_vis after the encoder), so 80 % of the rows of dout are exactly zero.  flags -> ascending list + count, all on the device.
__global__ __launch_bounds__(256) void pool_bwd_flags_kernel(const float* __restrict__ dout, int G, int C, int* __restrict__ flags) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= G) return;
    int nz = 0;
    for (int c = lane; c < C; c += 64) nz |= (dout[(size_t)g * C + c] != 0.f);
    const unsigned long long b = __ballot(nz);
    if (lane == 0) flags[g] = b != 0ull;
}
__global__ __launch_bounds__(1024) void pool_bwd_compact_kernel(const int* __restrict__ flags, int G, int* __restrict__ live, int* __restrict__ nlive) {
    __shared__ int wsum[16];
    const int tid = threadIdx.x, per = (G + 1023) / 1024, lo = min(G, tid * per), hi = min(G, lo + per);
    int cnt = 0;
    for (int g = lo; g < hi; ++g) cnt += flags[g];
    int incl = cnt;
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; ++w) base += wsum[w];
    int p = base + incl - cnt;
    for (int g = lo; g < hi; ++g) if (flags[g]) live[p++] = g;
    if (tid == 1023) *nlive = base + incl;
}

__global__ __launch_bounds__(256) void pool_bwd_fold_kernel(const float* __restrict__ part, int splits, size_t split_stride, int C, int N4,
                                                            float* __restrict__ out, int ldo) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)C * N4) return;
    const int c = (int)(i / N4), q = (int)(i - (long long)c * N4);
    const float4* __restrict__ p = reinterpret_cast<const float4*>(part) + i;
    const size_t st4 = split_stride / 4;
    float4 a = p[0];
    int s = 1;
    for (; s + 8 <= splits; s += 8) {                                            // 8 loads in flight; the additions stay in split order
        float4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = p[(size_t)(s + k) * st4];
#pragma unroll
        for (int k = 0; k < 8; ++k) { a.x += v[k].x; a.y += v[k].y; a.z += v[k].z; a.w += v[k].w; }
    }
    for (; s < splits; ++s) { const float4 v = p[(size_t)s * st4]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
    *reinterpret_cast<float4*>(out + (size_t)c * ldo + q * 4) = a;
}

// ACT_POOL_BWD_LIVE=0: walk every group, also those whose gradient row is all zero (A/B switch; results are identical)
bool pool_live_on() {
    static const bool on = [] { const char* e = getenv("ACT_POOL_BWD_LIVE"); return !(e && e[0] == '0'); }();
    return on;
}
bool pool_geom_ok(int n, int C) { return n > 0 && n <= 64 && 256 % n == 0 && C > 0 && C < 65536; }
int dw_waves(int C) { return C % 384 == 0 ? 6 : C % 256 == 0 ? 4 : C % 128 == 0 ? 2 : 0; }        // 64 channels per wave
int dw_splits(int G, int C, int N, size_t workspace_bytes) {
    const int tiles = (N / 64) * (C / (64 * dw_waves(C)));
    int s = (512 + tiles - 1) / tiles;                                           // two workgroups per CU (measured: 512 / 1024 / 2048 -> 271 / 290 / 326 us)
    s = min(s, max(1, G / 8));                                                   // at least 8 groups per workgroup
    const size_t cap = workspace_bytes / ((size_t)C * N * sizeof(float));
    if ((size_t)s > cap) s = (int)cap;
    return s;
}
}  // namespace

extern "C" int act_group_live_i32(const float* d, int G, int C, int32_t* live, act_stream_t stream) {
    if (!d || !live) return ACT_E_NULLPTR;
    if (G < 0 || C <= 0) return ACT_E_BADARG;
    if (G == 0) return 0;
    hipLaunchKernelGGL(pool_bwd_flags_kernel, dim3((G + 3) / 4), dim3(256), 0, (hipStream_t)stream, d, G, C, live);
    ACT_LAUNCH_CHECK(); return 0;
}

extern "C" int act_group_max_bwd_matmul_f32(const float* dout, const int32_t* arg, int G, int n, int C, const float* w, int ldw, int N, float* dx,
                                            int lddx, act_stream_t stream) {
    return act_group_max_bwd_matmul_live_f32(dout, arg, G, n, C, w, ldw, N, dx, lddx, nullptr, stream);
}

extern "C" int act_group_max_bwd_matmul_live_f32(const float* dout, const int32_t* arg, int G, int n, int C, const float* w, int ldw, int N, float* dx,
                                                 int lddx, const int32_t* live, act_stream_t stream) {
    if (!dout || !arg || !w || !dx) return ACT_E_NULLPTR;
    if (G < 0 || !pool_geom_ok(n, C) || N <= 0 || (N != 256 && N != 512 && N != 1024) || ldw < N || lddx < N || (ldw & 3) || (lddx & 3) ||
        ((uintptr_t)w & 15) || ((uintptr_t)dx & 15))
        return ACT_E_BADARG;
    if (G == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    // compulsory bytes: dx [G n, N] written, dout + arg [G, C] of the live groups read, w [C, N] read once (dead groups: their dx rows are zero-filled)
    const double lf_dx = (live && pool_live_on()) ? act_prof_live_fraction(live, G, s) : 1.0;
    ActProfScope ps(KID_MAXPOOL_BWD, s, 2.0 * G * lf_dx * (double)C * N, 4.0 * (G * ((double)n * N + 2.0 * C * lf_dx) + (double)C * N));
    const size_t lds = (size_t)C * 8 + (size_t)(n + 1 + 4) * 4 + (size_t)C * 2;
    const int parts = 1024 / N, skip = pool_live_on() ? 1 : 0;
    if (n % parts == 0) {
        if (N == 256)      hipLaunchKernelGGL(pool_bwd_dx4_kernel<4>, dim3(G), dim3(256), lds, s, dout, arg, w, ldw, n, C, dx, lddx, skip, live);
        else if (N == 512) hipLaunchKernelGGL(pool_bwd_dx4_kernel<2>, dim3(G), dim3(256), lds, s, dout, arg, w, ldw, n, C, dx, lddx, skip, live);
        else               hipLaunchKernelGGL(pool_bwd_dx4_kernel<1>, dim3(G), dim3(256), lds, s, dout, arg, w, ldw, n, C, dx, lddx, skip, live);
    } else if (N == 256)   hipLaunchKernelGGL(pool_bwd_dx_kernel<1>, dim3(G), dim3(256), lds, s, dout, arg, w, ldw, n, C, dx, lddx, skip, live);   // (n = 1, 2)
    else                   hipLaunchKernelGGL(pool_bwd_dx_kernel<2>, dim3(G), dim3(256), lds, s, dout, arg, w, ldw, n, C, dx, lddx, skip, live);
    ACT_LAUNCH_CHECK(); return 0;
}

static size_t live_list_bytes(int G) { return (((size_t)2 * G + 4) * sizeof(int) + 15) & ~(size_t)15; }   // flags | live | count

extern "C" size_t act_group_max_bwd_wgrad_workspace(int G, int n, int C, int N) {
    if (G <= 0 || !pool_geom_ok(n, C) || dw_waves(C) == 0 || N <= 0 || (N & 63)) return 0;
    const int s = dw_splits(G, C, N, (size_t)-1);
    return live_list_bytes(G) + (s > 1 ? (size_t)s * C * N * sizeof(float) : 0);
}

extern "C" int act_group_max_bwd_wgrad_f32(const float* dout, const int32_t* arg, int G, int n, int C, const float* x, int ldx, int N, const float* scale,
                                           const float* shift, float* dw, int lddw, float* workspace, size_t workspace_bytes, act_stream_t stream) {
    if (!dout || !arg || !x || !dw) return ACT_E_NULLPTR;
    if ((scale == nullptr) != (shift == nullptr)) return ACT_E_NULLPTR;
    const int nw = pool_geom_ok(n, C) ? dw_waves(C) : 0;
    if (G <= 0 || nw == 0 || N <= 0 || (N & 63) || ldx < N || lddw < N || (ldx & 3) || (lddw & 3) || ((uintptr_t)x & 15) || ((uintptr_t)dw & 15) ||
        (scale && (((uintptr_t)scale | (uintptr_t)shift) & 15)))
        return ACT_E_BADARG;
    if (workspace && (((uintptr_t)workspace) & 15)) return ACT_E_BADARG;
    // workspace: [flags | list of live groups | count] then the split partials; too small for the list -> every group is walked
    int *flags = nullptr, *live = nullptr, *nlive = nullptr;
    if (pool_live_on() && workspace && workspace_bytes >= live_list_bytes(G)) {
        flags = reinterpret_cast<int*>(workspace); live = flags + G; nlive = live + G;
        workspace = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + live_list_bytes(G));
        workspace_bytes -= live_list_bytes(G);
    } else {
        workspace_bytes = workspace ? workspace_bytes : 0;
    }
    int splits = dw_splits(G, C, N, workspace ? workspace_bytes : 0);
    if (splits < 1) splits = 1;
    const int gps = (G + splits - 1) / splits;
    splits = (G + gps - 1) / gps;
    hipStream_t s = (hipStream_t)stream;
    if (live) {
        hipLaunchKernelGGL(pool_bwd_flags_kernel, dim3((G + 3) / 4), dim3(256), 0, s, dout, G, C, flags);
        hipLaunchKernelGGL(pool_bwd_compact_kernel, dim3(1), dim3(1024), 0, s, flags, G, live, nlive);
    }
    // compulsory bytes (round 6): x rows of the LIVE groups read once, their dout + arg read once, dw [C, N] written once.  (The kernel re-reads x once per
    // 64 nw-channel block and dout / arg once per 64-column slice -- L2 traffic, which the previous model counted and which made alg_GBs exceed the HBM peak.)
    const double lf_dw = flags ? act_prof_live_fraction(flags, G, s) : 1.0;
    ActProfScope ps(KID_MAXPOOL_BWD, s, 2.0 * G * lf_dw * (double)C * N, 4.0 * (G * lf_dw * ((double)n * N + 2.0 * C) + (double)C * N));
    float* part = splits > 1 ? workspace : dw;
    const int ldp = splits > 1 ? N : lddw;
    const size_t stride = (size_t)C * N;
    const int gb = n <= 32 ? 4 : 2;
    const size_t lds = max((size_t)2 * gb * n * 68 * sizeof(float), (size_t)nw * 64 * 33 * sizeof(float));
    const int nslices = N / 64, xcd_map = splits % 8 == 0 ? 1 : 0;
    const dim3 grid(nslices * splits, C / (64 * nw), 1);
#define LAUNCH_DW2(NW_, GB_)                                                                                                                      \
    do {                                                                                                                                          \
        if (scale) hipLaunchKernelGGL((pool_bwd_dw2_kernel<NW_, GB_, true>), grid, dim3(64 * NW_), lds, s, dout, arg, x, ldx, scale, shift, n, C,  \
                                      G, gps, part, ldp, stride, nslices, xcd_map, live, nlive, splits);                                                               \
        else       hipLaunchKernelGGL((pool_bwd_dw2_kernel<NW_, GB_, false>), grid, dim3(64 * NW_), lds, s, dout, arg, x, ldx, scale, shift, n, C, \
                                      G, gps, part, ldp, stride, nslices, xcd_map, live, nlive, splits);                                                               \
    } while (0)
    if (gb == 4) { if (nw == 6) LAUNCH_DW2(6, 4); else if (nw == 4) LAUNCH_DW2(4, 4); else LAUNCH_DW2(2, 4); }
    else         { if (nw == 6) LAUNCH_DW2(6, 2); else if (nw == 4) LAUNCH_DW2(4, 2); else LAUNCH_DW2(2, 2); }
#undef LAUNCH_DW2
    ACT_LAUNCH_CHECK();
    if (splits > 1) {
        const long long total = (long long)C * (N / 4);
        hipLaunchKernelGGL(pool_bwd_fold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, part, splits, stride, C, N / 4, dw, lddw);
        ACT_LAUNCH_CHECK();
    }
    return 0;
}
