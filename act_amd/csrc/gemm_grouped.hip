// gemm_grouped.hip -- several weight-gradient GEMMs (+ their bias gradients) in ONE launch.
//
// A Transformer block's backward produces four weight gradients dW_p = dY_p^T . X_p (TN layout: both operands [tokens][features], the reduction
// runs over the tokens) and four bias gradients db_p = column sums of dY_p (models/act.py:25-90).  On the student's 1,792 = 128 x 14 token rows
// each of them is a 9 .. 36-tile launch that cannot fill 256 CUs, plus a split-K reduction launch, plus two column-sum launches: 16 launches
// per block for 6.3 GFLOP.  Here the work items of all problems of a group form ONE grid -- (problem, 128x128 tile, K range) -- with the problem
// table passed by value in the kernel arguments (no device-side table to upload: the call stays allocation- and copy-free), and ONE reduction
// launch folds the K-range partials of every product and of every bias in a fixed order (deterministic).
//   * main loop: the quad-fragment TN loop of sgemm_q16_kernel<128, 128, false, false> (gemm_q16_kernel.h) -- same products in the same order, so with
//     the same split factor the result is bit-identical to act_sgemm_ex_f32(tile 13, splits);
//   * db: the workgroups of tile column 0 add up the dY rows they stage anyway (the A operand passes through their registers on its way to
//     LDS): a thread sums its k-rows of every K-tile, eight thread-rows are folded through LDS in a fixed order -> one [128] partial per
//     (tile row, K range) -> the reduction launch adds the K ranges.
#include "gemm_common.h"
#include "gemm_nt_asm_loop.h"
#include <stdlib.h>
#define GG_MAXP 8

struct GroupedProblem {
    const float* A; const float* B; float* C; float* bias_out;         // A [K][M] (lda), B [K][N] (ldb), C [M][N] (ldc), bias_out [M] or null
    int lda, ldb, ldc, M, N;
    int tile0;                                                          // first work tile of this problem in the launch's tile space
    long long part_off, bias_off;                                       // float offsets of its [splits][M][N] / [splits][M] partials in the workspace
};
struct GroupedParams {
    GroupedProblem p[GG_MAXP];
    int nprob, K, k_per_split, splits, total_tiles;
    float* ws;
};

__global__ __launch_bounds__(256, 2) void sgemm_tn_grouped_kernel(const GroupedParams g) {
    constexpr int BM = 128, BN = 128, BK = 16;
    __shared__ __attribute__((aligned(16))) float As[2][BM * BK];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * BK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // work item -> (K range, problem, tile): K ranges outermost, so the workgroups running at the same time share operand panels
    const int sp = blockIdx.x / g.total_tiles, tl = blockIdx.x - sp * g.total_tiles;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GG_MAXP; ++i) if (i < g.nprob && tl >= g.p[i].tile0) pi = i;
    // (copy the selected problem out of the by-value table with uniform indexing: the table lives in SGPRs / the kernarg segment)
    const float* __restrict__ A = g.p[pi].A; const float* __restrict__ B = g.p[pi].B;
    const int lda = g.p[pi].lda, ldb = g.p[pi].ldb, M = g.p[pi].M, N = g.p[pi].N;
    const int t = tl - g.p[pi].tile0, tiles_n = N / BN;
    const int tile_m = t / tiles_n, tile_n = t - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = sp * g.k_per_split, kend = min(g.K, kbeg + g.k_per_split);
    const int ntiles = (kend - kbeg) / BK;
    const bool want_bias = g.p[pi].bias_out != nullptr && tile_n == 0;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging: thread owns float4 #(tid + 256 i) of the [16][128] tile of each operand: k = tid / 32 (+8), 4 consecutive rows at (tid % 32) * 4
    const float* ga = A + (size_t)(kbeg + tid / 32) * lda + m0 + (tid % 32) * 4;
    const float* gb = B + (size_t)(kbeg + tid / 32) * ldb + n0 + (tid % 32) * 4;
    const size_t sa = (size_t)BK * lda, sb = (size_t)BK * ldb, a2 = (size_t)8 * lda, b2 = (size_t)8 * ldb;
    float4 ra0, ra1, rb0, rb1;
    ra0 = ra1 = rb0 = rb1 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_g = [&](int kt) {
        ra0 = *reinterpret_cast<const float4*>(ga + kt * sa); ra1 = *reinterpret_cast<const float4*>(ga + a2 + kt * sa);
        rb0 = *reinterpret_cast<const float4*>(gb + kt * sb); rb1 = *reinterpret_cast<const float4*>(gb + b2 + kt * sb);
    };
    auto store_lds = [&](int buf) {
        if (want_bias) {                                                // k-rows in ascending order: (t, k) then (t, k + 8)
            bsum.x += ra0.x; bsum.y += ra0.y; bsum.z += ra0.z; bsum.w += ra0.w;
            bsum.x += ra1.x; bsum.y += ra1.y; bsum.z += ra1.z; bsum.w += ra1.w;
        }
        *reinterpret_cast<float4*>(&As[buf][tid * 4]) = ra0; *reinterpret_cast<float4*>(&As[buf][tid * 4 + 1024]) = ra1;
        *reinterpret_cast<float4*>(&Bs[buf][tid * 4]) = rb0; *reinterpret_cast<float4*>(&Bs[buf][tid * 4 + 1024]) = rb1;
    };
    if (ntiles > 0) { load_g(0); store_lds(0); __syncthreads(); }
    const int kl = lane >> 4, ml = lane & 15;
    const int a_off = (4 * kl) * BM + wm * 64 + 4 * ml, b_off = (4 * kl) * BN + wn * 64 + 4 * ml;
    auto compute = [&](int buf) {
        float4 af[4], bf[4];                            // af[s] = the 4 row blocks of k-step s
#pragma unroll
        for (int s = 0; s < 4; ++s) af[s] = *reinterpret_cast<const float4*>(&As[buf][a_off + s * BM]);
#pragma unroll
        for (int s = 0; s < 4; ++s) bf[s] = *reinterpret_cast<const float4*>(&Bs[buf][b_off + s * BN]);
        auto el = [](const float4& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w)); };
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(el(af[s], i), el(bf[s], j), acc[i][j], 0, 0, 0);
    };
    for (int kt = 0; kt + 1 < ntiles; ++kt) {
        load_g(kt + 1);
        compute(kt & 1);
        store_lds((kt & 1) ^ 1);
        __syncthreads();
    }
    if (ntiles > 0) compute((ntiles - 1) & 1);

    // epilogue: row of (block i, MFMA row r16) = wm*64 + 4 r16 + i; a lane owns 4 consecutive columns wn*64 + 4 ml .. +3
    float* __restrict__ dst; int ldd;
    if (g.splits > 1) { dst = g.ws + g.p[pi].part_off + (size_t)sp * M * N; ldd = N; }
    else              { dst = g.p[pi].C; ldd = g.p[pi].ldc; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wm * 64 + 4 * (kl * 4 + r) + i, col = n0 + wn * 64 + 4 * ml;
            *reinterpret_cast<float4*>(dst + (size_t)row * ldd + col) = make_float4(acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]);
        }
    if (want_bias) {                                                    // fold the 8 thread-rows (k mod 8) in a fixed order
        __syncthreads();                                                // As is free now
        float* red = &As[0][0];
        *reinterpret_cast<float4*>(&red[(tid / 32) * BM + (tid % 32) * 4]) = bsum;
        __syncthreads();
        if (tid < BM) {
            float v = red[tid];
#pragma unroll
            for (int q = 1; q < 8; ++q) v += red[q * BM + tid];
            if (g.splits > 1) g.ws[g.p[pi].bias_off + (size_t)sp * M + m0 + tid] = v;
            else              g.p[pi].bias_out[m0 + tid] = v;
        }
    }
}

// ---- the same work decomposition on the hand-scheduled TN main loop (gen_nt_asm.py: tn_asm_loop_4x4 / ..._asum): 32-deep K tiles, every memory
// instruction in an MFMA gap; the bias-gradient column sums ride on the staging registers inside the loop (v_add_f32 in the gaps, chunks in ascending k:
// the order of the compiler loop).  Same products and sums in the same order: bit-identical to sgemm_tn_grouped_kernel.  Needs every K range % 32 == 0.
__global__ __launch_bounds__(256, 2) void sgemm_tn_grouped_asm_kernel(const GroupedParams g) {
    constexpr int BM = 128, BN = 128;
    constexpr int KG = BM * 64, B_BASE = 2 * KG, STAGE = 4 * KG;                 // bytes
    __shared__ __attribute__((aligned(16))) char lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int sp = blockIdx.x / g.total_tiles, tl = blockIdx.x - sp * g.total_tiles;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < GG_MAXP; ++i) if (i < g.nprob && tl >= g.p[i].tile0) pi = i;
    const float* __restrict__ A = g.p[pi].A; const float* __restrict__ B = g.p[pi].B;
    const int lda = g.p[pi].lda, ldb = g.p[pi].ldb, M = g.p[pi].M, N = g.p[pi].N;
    const int t = tl - g.p[pi].tile0, tiles_n = N / BN;
    const int tile_m = t / tiles_n, tile_n = t - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = sp * g.k_per_split, kend = min(g.K, kbeg + g.k_per_split);
    const int ntiles = (kend - kbeg) / 32;
    const bool want_bias = g.p[pi].bias_out != nullptr && tile_n == 0;
    const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    // staging: thread owns float4 #(tid + 256 i) of the [32 k][128] tile of each operand: k = tid / 32 + 8 i, 4 consecutive rows at (tid % 32) * 4
    u32x4 offa, offb;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        offa[i] = (unsigned)(((tid / 32 + 8 * i) * lda + (tid % 32) * 4) * 4);
        offb[i] = (unsigned)(((tid / 32 + 8 * i) * ldb + (tid % 32) * 4) * 4);
    }
    const float* pa = A + (size_t)kbeg * lda + m0;
    const float* pb = B + (size_t)kbeg * ldb + n0;
    const int kl = lane >> 4, ml = lane & 15;
    f32x4 acc[4][4];
    f32x4 bsum = f32x4{0.f, 0.f, 0.f, 0.f};
    if (ntiles > 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {                                    // K-tile 0 -> stage 0 (k-rows in ascending order for the bias sums)
            const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(pa) + offa[i]);
            if (want_bias) { bsum[0] += v.x; bsum[1] += v.y; bsum[2] += v.z; bsum[3] += v.w; }
            *reinterpret_cast<float4*>(lds + tid * 16 + i * 4096) = v;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<float4*>(lds + B_BASE + tid * 16 + j * 4096) = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(pb) + offb[j]);
        __syncthreads();
        const unsigned ra = lbase + (unsigned)(((4 * kl) * BM + wm * 64 + 4 * ml) * 4), rb = lbase + (unsigned)(((4 * kl) * BN + wn * 64 + 4 * ml) * 4);
        const unsigned wb = lbase + tid * 16;
        if (want_bias) tn_asm_loop_4x4_asum(acc, pa, pb, ntiles, (unsigned)(32 * lda * 4), (unsigned)(32 * ldb * 4), offa, offb, wb, wb, ra, rb, bsum);
        else           tn_asm_loop_4x4(acc, pa, pb, ntiles, (unsigned)(32 * lda * 4), (unsigned)(32 * ldb * 4), offa, offb, wb, wb, ra, rb);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float* __restrict__ dst; int ldd;
    if (g.splits > 1) { dst = g.ws + g.p[pi].part_off + (size_t)sp * M * N; ldd = N; }
    else              { dst = g.p[pi].C; ldd = g.p[pi].ldc; }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wm * 64 + 4 * (kl * 4 + r) + i, col = n0 + wn * 64 + 4 * ml;
            *reinterpret_cast<float4*>(dst + (size_t)row * ldd + col) = make_float4(acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]);
        }
    if (want_bias) {                                                    // fold the 8 thread-rows (k mod 8) in a fixed order
        __syncthreads();                                                // the staging LDS is free now
        float* red = reinterpret_cast<float*>(lds);
        *reinterpret_cast<float4*>(&red[(tid / 32) * BM + (tid % 32) * 4]) = make_float4(bsum[0], bsum[1], bsum[2], bsum[3]);
        __syncthreads();
        if (tid < BM) {
            float v = red[tid];
#pragma unroll
            for (int q = 1; q < 8; ++q) v += red[q * BM + tid];
            if (g.splits > 1) g.ws[g.p[pi].bias_off + (size_t)sp * M + m0 + tid] = v;
            else              g.p[pi].bias_out[m0 + tid] = v;
        }
    }
}

// folds the K-range partials of every product (float4 per thread, ranges in ascending order) and of every bias of the group
__global__ __launch_bounds__(256) void sgemm_grouped_reduce_kernel(const GroupedParams g, long long total4, long long total_bias) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total4 + total_bias; e += stride) {
        if (e < total4) {
            long long rem = e; int pi = 0;
            for (int i = 0; i < g.nprob; ++i) {
                const long long n4 = (long long)g.p[i].M * g.p[i].N / 4;
                if (rem < n4) { pi = i; break; }
                rem -= n4;
            }
            const int N = g.p[pi].N, M = g.p[pi].M;
            const long long off = rem * 4;
            const float* part = g.ws + g.p[pi].part_off + off;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4                                                      // (four loads in flight; the additions stay in K-range order)
            for (int s = 0; s < g.splits; ++s) {
                const float4 x = *reinterpret_cast<const float4*>(part + (size_t)s * M * N);
                v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
            }
            const int row = (int)(off / N), col = (int)(off - (long long)row * N);
            *reinterpret_cast<float4*>(g.p[pi].C + (size_t)row * g.p[pi].ldc + col) = v;
        } else {
            long long rem = e - total4; int pi = -1;
            for (int i = 0; i < g.nprob; ++i) {
                if (!g.p[i].bias_out) continue;
                if (rem < g.p[i].M) { pi = i; break; }
                rem -= g.p[i].M;
            }
            if (pi < 0) continue;
            const float* part = g.ws + g.p[pi].bias_off + rem;
            float v = 0.f;
#pragma unroll 4
            for (int s = 0; s < g.splits; ++s) v += part[(size_t)s * g.p[pi].M];
            g.p[pi].bias_out[rem] = v;
        }
    }
}

extern "C" size_t act_sgemm_tn_grouped_workspace(const act_gemm_tn_problem_t* probs, int nprob, int K, int splits) {
    if (!probs || nprob <= 0 || nprob > GG_MAXP || splits <= 1) return 0;
    size_t fl = 0;
    for (int i = 0; i < nprob; ++i) fl += (size_t)probs[i].M * probs[i].N + (probs[i].bias_out ? (size_t)probs[i].M : 0);
    return fl * (size_t)splits * sizeof(float);
}

// default K-range count of a group.  Measured model (benchmarks/grouped_bench.py): two workgroups share a CU, so the launch takes
// ceil(tiles * splits / 512) rounds of a workgroup's K range plus a fixed prologue / epilogue worth ~64 rows; the count that minimises
// rounds * (K / splits + 64) wins (1,792 rows: 7 ranges for the two MLP gradients -- 504 workgroups, one round --, 14 for proj + qkv).
extern "C" int act_sgemm_tn_grouped_splits(const act_gemm_tn_problem_t* probs, int nprob, int K) {
    if (!probs || nprob <= 0 || K <= 0) return 1;
    long long tiles = 0;
    for (int i = 0; i < nprob; ++i) tiles += (long long)(probs[i].M / 128) * (probs[i].N / 128);
    if (tiles <= 0) return 1;
    int best = 1; double best_cost = 1e300;
    const int maxs = K / 128 < 32 ? (K / 128 < 1 ? 1 : K / 128) : 32;
    for (int sp = 1; sp <= maxs; ++sp) {
        int kps = (K + sp - 1) / sp; kps = (kps + 31) / 32 * 32;
        if ((K + kps - 1) / kps != sp) continue;                        // this count collapses to a smaller one after rounding the ranges
        const double rounds = (double)((tiles * sp + 511) / 512);
        const double cost = rounds * (kps + 64.0);
        if (cost < best_cost) { best_cost = cost; best = sp; }
    }
    return best;
}

extern "C" int act_sgemm_tn_grouped_f32(const act_gemm_tn_problem_t* probs, int nprob, int K, int splits, float* workspace, size_t workspace_bytes,
                                        act_stream_t stream) {
    if (!probs) return ACT_E_NULLPTR;
    if (nprob <= 0 || nprob > GG_MAXP || K <= 0 || (K % 16) != 0) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    GroupedParams g{};
    g.nprob = nprob; g.K = K;
    if (splits <= 0) splits = act_sgemm_tn_grouped_splits(probs, nprob, K);
    int kps = (K + splits - 1) / splits; kps = (kps + 31) / 32 * 32; splits = (K + kps - 1) / kps;
    g.k_per_split = kps; g.splits = splits;
    long long off = 0; int tiles = 0; double flops = 0.0, bytes = 0.0;
    long long total4 = 0, total_bias = 0;
    for (int i = 0; i < nprob; ++i) {
        const act_gemm_tn_problem_t& q = probs[i];
        if (!q.A || !q.B || !q.C) return ACT_E_NULLPTR;
        if (q.M <= 0 || q.N <= 0 || (q.M % 128) || (q.N % 128) || (q.lda & 3) || (q.ldb & 3) || (q.ldc & 3) || q.lda < q.M || q.ldb < q.N || q.ldc < q.N) return ACT_E_BADARG;
        if ((reinterpret_cast<uintptr_t>(q.A) | reinterpret_cast<uintptr_t>(q.B) | reinterpret_cast<uintptr_t>(q.C)) & 15) return ACT_E_BADARG;
        GroupedProblem& d = g.p[i];
        d.A = q.A; d.B = q.B; d.C = q.C; d.bias_out = q.bias_out; d.lda = q.lda; d.ldb = q.ldb; d.ldc = q.ldc; d.M = q.M; d.N = q.N;
        d.tile0 = tiles; tiles += (q.M / 128) * (q.N / 128);
        d.part_off = off; off += (long long)splits * q.M * q.N;
        total4 += (long long)q.M * q.N / 4;
        flops += 2.0 * q.M * q.N * (double)K; bytes += 4.0 * ((double)q.M * K + (double)q.N * K + (double)q.M * q.N);
    }
    for (int i = 0; i < nprob; ++i) {
        g.p[i].bias_off = off;
        if (g.p[i].bias_out) { off += (long long)splits * g.p[i].M; total_bias += g.p[i].M; }
    }
    g.total_tiles = tiles;
    if (splits > 1) {
        if (!workspace) return ACT_E_NULLPTR;
        if ((size_t)off * sizeof(float) > workspace_bytes) return ACT_E_BADARG;
    }
    g.ws = workspace;
    ActProfScope ps(KID_GEMM_TN, s, flops, bytes);
    // hand-scheduled main loop when every K range is a multiple of 32 rows and the 32-bit lane offsets of a K tile fit (ACT_GEMM_GROUPED_ASM=0: compiler loop)
    static const int use_asm = [] { const char* e = getenv("ACT_GEMM_GROUPED_ASM"); return e ? atoi(e) : 1; }();
    bool asm_ok = use_asm && (K % 32) == 0;
    for (int i = 0; i < nprob && asm_ok; ++i)
        if ((long long)32 * probs[i].lda * 4 >= (1ll << 31) || (long long)32 * probs[i].ldb * 4 >= (1ll << 31)) asm_ok = false;
    if (asm_ok) hipLaunchKernelGGL(sgemm_tn_grouped_asm_kernel, dim3((unsigned)(tiles * splits)), dim3(256), 0, s, g);
    else        hipLaunchKernelGGL(sgemm_tn_grouped_kernel, dim3((unsigned)(tiles * splits)), dim3(256), 0, s, g);
    ACT_LAUNCH_CHECK();
    if (splits > 1) {
        long long blocks = (total4 + total_bias + 255) / 256; if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(sgemm_grouped_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g, total4, total_bias);
        ACT_LAUNCH_CHECK();
    }
    return 0;
}
