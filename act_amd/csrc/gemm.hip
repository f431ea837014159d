// gemm.hip -- fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32) with fused epilogues.
//
// One kernel family serves every dense product on the ACT hot path:
//   Linear / Conv1d(k=1) forward   C[M,N] = A[M,K] . W[N,K]^T      (a_kmajor=1, b_kmajor=1)
//   input gradient                 dX[M,K'] = dY[M,N'] . W[N',K']   (a_kmajor=1, b_kmajor=0)
//   weight gradient                dW[N',K'] = dY[T,N']^T . X[T,K'] (a_kmajor=0, b_kmajor=0, split-K over T)
// (reference ops: nn.Linear in models/act.py:25-69, Conv1d(k=1) in models/dvae.py:185-215, timm ViT blocks.)
//
// Tiling for 64-wide wavefronts: a 256-thread workgroup (4 waves as 2x2) owns a BM x BN tile, each wave a
// (BM/2)x(BN/2) sub-tile made of 32x32 MFMA accumulators.  Operands are staged through LDS as [k][row]
// (row contiguous, +4 pad) so every MFMA operand fetch is a conflict-free ds_read_b32 of 32 consecutive
// dwords per half-wave; K-contiguous global operands are transposed on the LDS write, row-contiguous ones are
// stored with ds_write_b128.  Global loads for tile t+1 are issued before the MFMAs of tile t (register
// staging, double-buffered LDS, one barrier per K-tile).  Workgroup ids are remapped so each XCD (own L2)
// works on a compact band of tiles.  The epilogue fuses bias, GELU / ReLU (+ saving the pre-activation),
// activation-gradient multiply, per-row scale (DropPath gate) and residual add.
#include "gemm_common.h"
#include <stdlib.h>


// epilogue of sgemm_kernel for one activation (C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)); ragged edges guarded
template <int ACT, int BM, int BN, int TM, int TN, typename Acc>
__device__ __forceinline__ void epilogue32(const GemmParams& p, Acc& acc, int m0, int n0, int wm, int wn, int lane) {
    const int col_l = lane & 31, row_l = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * (BN / 2) + j * 32 + col_l;
            if (col >= p.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + row_l;
                if (row >= p.M) continue;
                float v = acc[i][j][r];
                if (p.partial) {
                    p.partial[((size_t)blockIdx.z * p.M + row) * p.N + col] = v;
                } else {
                    v = epilogue_apply<ACT>(p.epi, v, row, col);
                    float* c = p.C + (size_t)row * p.ldc + col;
                    if (p.epi.accumulate) v += *c;
                    *c = v;
                }
            }
        }
}

// VEC : float4 global loads are legal (alignment, leading dimensions)      FULL: M%BM == N%BN == 0 and every K-range is a
// multiple of BK, so the loaders carry no bounds checks at all.
// PIPE: software-pipelined main loop for low-occupancy launches -- three LDS stages, the LDS store of tile t+1 and the
// workgroup barrier sit in the MIDDLE of tile t's MFMA stream, and the first operand fragments of tile t+1 are fetched under
// the last MFMAs of tile t, so a wave never waits on LDS latency or on the barrier with an empty matrix pipe.
template <int BM, int BN, int BK, bool A_K, bool B_K, bool VEC, bool FULL, bool PIPE>
__global__ __launch_bounds__(256, PIPE ? 3 : GEMM_MIN_WAVES) void sgemm_kernel(const GemmParams p) {
    // LDS row stride: K-major operands are transposed on the store (4 x ds_write_b32 per float4): stride = rows + 2 makes
    // the 32 lanes of a half-wave hit 32 distinct banks ((8*kq + 2*c + row) mod 32); row-major operands are stored with
    // ds_write_b128 and need a 16-byte aligned stride (rows + 4).  Operand reads are conflict-free for any stride.
    constexpr int LDA_S = A_K ? BM + 2 : BM + 4, LDB_S = B_K ? BN + 2 : BN + 4;
    constexpr int TM = BM / 64, TN = BN / 64;          // 32x32 accumulators per wave
    constexpr int NA = BM * BK / 1024, NB = BN * BK / 1024;   // float4 loads per thread per operand per K-tile
    constexpr int KQ = BK / 4;                         // float4 per row of a K-major tile
    constexpr int NSTAGE = PIPE ? 3 : 2;
    __shared__ __attribute__((aligned(16))) float As[NSTAGE][BK * LDA_S];
    __shared__ __attribute__((aligned(16))) float Bs[NSTAGE][BK * LDB_S];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware remap: workgroup b is dispatched to XCD b%8; give each XCD a contiguous band of tiles
    const int nwg = p.tiles_m * p.tiles_n;
    int wg = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = wg & 7, loc = wg >> 3;
        wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    int tile_m, tile_n;
    tile_coords(p, wg, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = blockIdx.z * p.k_per_split;
    const int kend = min(p.K, kbeg + p.k_per_split);
    const int ntiles = (kend - kbeg + BK - 1) / BK;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[NA], rb[NB];

    auto load_a = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int v = tid + 256 * i;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (A_K) {                                   // A stored [M][K]
                const int row = m0 + v / KQ, kk = k0 + (v % KQ) * 4;
                if (FULL) { x = *reinterpret_cast<const float4*>(p.A + (size_t)row * p.lda + kk); }
                else if (row < p.M) {
                    const float* src = p.A + (size_t)row * p.lda + kk;
                    if (VEC) { if (kk < kend) x = *reinterpret_cast<const float4*>(src); }
                    else { if (kk < kend) x.x = src[0]; if (kk + 1 < kend) x.y = src[1]; if (kk + 2 < kend) x.z = src[2]; if (kk + 3 < kend) x.w = src[3]; }
                }
            } else {                                     // A stored [K][M]
                const int kk = k0 + v / (BM / 4), row = m0 + (v % (BM / 4)) * 4;
                if (FULL) { x = *reinterpret_cast<const float4*>(p.A + (size_t)kk * p.lda + row); }
                else if (kk < kend) {
                    const float* src = p.A + (size_t)kk * p.lda + row;
                    if (VEC) { if (row < p.M) x = *reinterpret_cast<const float4*>(src); }
                    else { if (row < p.M) x.x = src[0]; if (row + 1 < p.M) x.y = src[1]; if (row + 2 < p.M) x.z = src[2]; if (row + 3 < p.M) x.w = src[3]; }
                }
            }
            ra[i] = x;
        }
    };
    auto load_b = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int v = tid + 256 * i;
            float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
            if (B_K) {                                   // B stored [N][K]
                const int row = n0 + v / KQ, kk = k0 + (v % KQ) * 4;
                if (FULL) { x = *reinterpret_cast<const float4*>(p.B + (size_t)row * p.ldb + kk); }
                else if (row < p.N) {
                    const float* src = p.B + (size_t)row * p.ldb + kk;
                    if (VEC) { if (kk < kend) x = *reinterpret_cast<const float4*>(src); }
                    else { if (kk < kend) x.x = src[0]; if (kk + 1 < kend) x.y = src[1]; if (kk + 2 < kend) x.z = src[2]; if (kk + 3 < kend) x.w = src[3]; }
                }
            } else {                                     // B stored [K][N]
                const int kk = k0 + v / (BN / 4), row = n0 + (v % (BN / 4)) * 4;
                if (FULL) { x = *reinterpret_cast<const float4*>(p.B + (size_t)kk * p.ldb + row); }
                else if (kk < kend) {
                    const float* src = p.B + (size_t)kk * p.ldb + row;
                    if (VEC) { if (row < p.N) x = *reinterpret_cast<const float4*>(src); }
                    else { if (row < p.N) x.x = src[0]; if (row + 1 < p.N) x.y = src[1]; if (row + 2 < p.N) x.z = src[2]; if (row + 3 < p.N) x.w = src[3]; }
                }
            }
            rb[i] = x;
        }
    };
    auto store_lds = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int v = tid + 256 * i;
            if (A_K) {
                float* d = &As[buf][((v % KQ) * 4) * LDA_S + v / KQ];
                d[0] = ra[i].x; d[LDA_S] = ra[i].y; d[2 * LDA_S] = ra[i].z; d[3 * LDA_S] = ra[i].w;
            } else {
                *reinterpret_cast<float4*>(&As[buf][(v / (BM / 4)) * LDA_S + (v % (BM / 4)) * 4]) = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int v = tid + 256 * i;
            if (B_K) {
                float* d = &Bs[buf][((v % KQ) * 4) * LDB_S + v / KQ];
                d[0] = rb[i].x; d[LDB_S] = rb[i].y; d[2 * LDB_S] = rb[i].z; d[3 * LDB_S] = rb[i].w;
            } else {
                *reinterpret_cast<float4*>(&Bs[buf][(v / (BN / 4)) * LDB_S + (v % (BN / 4)) * 4]) = rb[i];
            }
        }
    };

    const int a_off = wm * (BM / 2) + (lane & 31), b_off = wn * (BN / 2) + (lane & 31), khalf = lane >> 5;
    if constexpr (!PIPE) {
        if (ntiles > 0) {
            load_a(kbeg); load_b(kbeg);
            store_lds(0);
            __syncthreads();
        }
        for (int t = 0; t < ntiles; ++t) {
            const int buf = t & 1;
            if (t + 1 < ntiles) { load_a(kbeg + (t + 1) * BK); load_b(kbeg + (t + 1) * BK); }
            const float* as = As[buf] + khalf * LDA_S + a_off;
            const float* bs = Bs[buf] + khalf * LDB_S + b_off;
            float a[2][TM], b[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[0][i] = as[i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[0][j] = bs[j * 32];
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
                if (kk + 2 < BK) {                      // fetch the next k-pair while this one is in the matrix pipe
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[nxt][i] = as[(kk + 2) * LDA_S + i * 32];
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[nxt][j] = bs[(kk + 2) * LDB_S + j * 32];
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
            }
            if (t + 1 < ntiles) store_lds(buf ^ 1);
            __syncthreads();
        }
    } else {
        float a[2][TM], b[2][TN];
        if (ntiles > 0) {
            load_a(kbeg); load_b(kbeg);
            store_lds(0);
            if (ntiles > 1) { load_a(kbeg + BK); load_b(kbeg + BK); }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < TM; ++i) a[0][i] = As[0][khalf * LDA_S + a_off + i * 32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[0][j] = Bs[0][khalf * LDB_S + b_off + j * 32];
        }
        int cb = 0;                                     // LDS stage of tile t
        for (int t = 0; t < ntiles; ++t) {
            const int nb = cb == 2 ? 0 : cb + 1;
            const float* as = As[cb] + khalf * LDA_S + a_off;
            const float* bs = Bs[cb] + khalf * LDB_S + b_off;
            const float* an = As[nb] + khalf * LDA_S + a_off;
            const float* bn = Bs[nb] + khalf * LDB_S + b_off;
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                const int cur = (kk >> 1) & 1, nxt = cur ^ 1;
                if (kk == BK / 2) {                     // mid-tile: publish tile t+1, start fetching tile t+2, rendezvous
                    if (t + 1 < ntiles) store_lds(nb);
                    if (t + 2 < ntiles) { load_a(kbeg + (t + 2) * BK); load_b(kbeg + (t + 2) * BK); }
                    __syncthreads();
                }
                if (kk + 2 < BK) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[nxt][i] = as[(kk + 2) * LDA_S + i * 32];
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[nxt][j] = bs[(kk + 2) * LDB_S + j * 32];
                } else if (t + 1 < ntiles) {            // first fragments of the next tile, under this tile's last MFMAs
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[nxt][i] = an[i * 32];
#pragma unroll
                    for (int j = 0; j < TN; ++j) b[nxt][j] = bn[j * 32];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            cb = nb;
        }
    }

    // epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    // one uniform branch: launches without activation (the common case) take a lean body, everything else the body with the run-time switch
    // (instantiating all five activations here multiplies the compile time of this file's ~100 kernel variants by four)
    if (p.partial || p.epi.act == ACT_EPI_NONE) epilogue32<ACT_EPI_NONE, BM, BN, TM, TN>(p, acc, m0, n0, wm, wn, lane);
    else                                        epilogue32<-1, BM, BN, TM, TN>(p, acc, m0, n0, wm, wn, lane);
}

// split-K reduction + epilogue (deterministic: fixed summation order over splits).  VEC: four consecutive columns per thread (float4 partial
// loads, one row / column division per four outputs) -- same sums in the same order as the scalar form, element for element.
template <bool VEC>
__global__ __launch_bounds__(256) void sgemm_splitk_reduce(const float* __restrict__ partial, int splits, int M, int N, float* __restrict__ C, int ldc,
                                                           const act_gemm_epilogue_t epi) {
    const long long total = (long long)M * N;
    if constexpr (VEC) {
        const int n4 = N >> 2;
        const long long total4 = (long long)M * n4;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
            const int row = (int)(i / n4), col = (int)(i - (long long)row * n4) * 4;
            const float* p = partial + (size_t)row * N + col;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4                                                      // (four loads in flight; the additions stay in split order)
            for (int s = 0; s < splits; ++s) {
                const float4 x = *reinterpret_cast<const float4*>(p + (size_t)s * total);
                v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
            }
            v = epilogue_apply4(epi, v, row, col);
            float4* c = reinterpret_cast<float4*>(C + (size_t)row * ldc + col);
            if (epi.accumulate) { const float4 o = *c; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *c = v;
        }
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
            float v = 0.f;
#pragma unroll 4
            for (int s = 0; s < splits; ++s) v += partial[(size_t)s * total + i];
            const int row = (int)(i / N), col = (int)(i % N);
            v = epilogue_apply(epi, v, row, col);
            float* c = C + (size_t)row * ldc + col;
            if (epi.accumulate) v += *c;
            *c = v;
        }
    }
}
static void launch_splitk_reduce(const float* partial, int splits, int M, int N, float* C, int ldc, const act_gemm_epilogue_t& epi, hipStream_t s) {
    const bool vec = (N & 3) == 0 && epilogue_is_vec(C, ldc, epi) && (reinterpret_cast<uintptr_t>(partial) & 15) == 0;
    const long long total = vec ? (long long)M * (N >> 2) : (long long)M * N;
    long long g = (total + 255) / 256; if (g > 4096) g = 4096; if (g < 1) g = 1;
    if (vec) hipLaunchKernelGGL(sgemm_splitk_reduce<true>, dim3((unsigned)g), dim3(256), 0, s, partial, splits, M, N, C, ldc, epi);
    else     hipLaunchKernelGGL(sgemm_splitk_reduce<false>, dim3((unsigned)g), dim3(256), 0, s, partial, splits, M, N, C, ldc, epi);
}

// Skinny weight gradients (TN, min(M,N) <= 8, K = tokens): e.g. dW of the 3->128 first conv, of the 512->3 / 5->512 FoldingNet
// convs.  A 64x64 MFMA tile would be >90 % padding and the tile grid cannot fill the chip, so this is a streaming reduction
// instead: the wide operand is read once, coalesced (thread <-> wide column, 4 k-lanes per workgroup), the skinny operand is
// wave-uniform, partial sums per K range go to the split-K workspace and the ordinary deterministic reduce + epilogue finishes.
template <bool SMALL_N>
__global__ __launch_bounds__(256) void sgemm_tn_skinny_kernel(const float* __restrict__ wide, int ldw, int W, const float* __restrict__ skinny,
                                                              int lds, int ns, int K, int rows_per_part, int M, int N,
                                                              float* __restrict__ partial) {
    __shared__ float red[3][64][8];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), klane = threadIdx.x >> 6;
    const int kbeg = blockIdx.y * rows_per_part, kend = min(K, kbeg + rows_per_part);
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    if (col < W) {
#pragma unroll 4
        for (int k = kbeg + klane; k < kend; k += 4) {
            const float w = wide[(size_t)k * ldw + col];
            const float* sk = skinny + (size_t)k * lds;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < ns) acc[j] += w * sk[j];
        }
    }
    if (klane > 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[klane - 1][threadIdx.x & 63][j] = acc[j];
    }
    __syncthreads();
    if (klane == 0 && col < W) {
        float* dst = partial + (size_t)blockIdx.y * M * N;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < ns) {
                const float v = (acc[j] + red[0][threadIdx.x][j]) + (red[1][threadIdx.x][j] + red[2][threadIdx.x][j]);
                if (SMALL_N) dst[(size_t)col * N + j] = v;      // C[m = col][n = j]
                else         dst[(size_t)j * N + col] = v;      // C[m = j][n = col]
            }
    }
}

// The same reduction with 16-byte loads of the wide operand (W % 4 == 0, 16-byte aligned): a row of W floats takes W / 4 threads, the workgroup's
// other threads are k-lanes (8 at W = 128), every wave-level load moves 1 KB instead of 256 B.  Round 4: the dW of the 3 -> 128 first conv
// (wide = dh1 [262144, 128], 134 MB) 166 -> see DESIGN; the launch is also cut into enough K ranges to fill the chip when K is only a few
// thousand rows (the 16-workgroup launches of the position-embedding weight gradients took 117 us for 4 MB).
template <bool SMALL_N>
__global__ __launch_bounds__(256) void sgemm_tn_skinny4_kernel(const float* __restrict__ wide, int ldw, int W, const float* __restrict__ skinny,
                                                               int lds, int ns, int K, int rows_per_part, int M, int N, int tpr_shift,
                                                               float* __restrict__ partial) {
    extern __shared__ float red4[];                                   // [KL - 1][tpr][4 * 8]
    const int tpr = 1 << tpr_shift, KL = 256 >> tpr_shift;
    const int c4 = threadIdx.x & (tpr - 1), klane = threadIdx.x >> tpr_shift;
    const int col = (blockIdx.x * tpr + c4) * 4;
    const int kbeg = blockIdx.y * rows_per_part, kend = min(K, kbeg + rows_per_part);
    float acc[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[q][j] = 0.f;
    if (col < W) {
#pragma unroll 4
        for (int k = kbeg + klane; k < kend; k += KL) {
            const float4 w = *reinterpret_cast<const float4*>(wide + (size_t)k * ldw + col);
            const float* sk = skinny + (size_t)k * lds;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < ns) { const float sv = sk[j]; acc[0][j] += w.x * sv; acc[1][j] += w.y * sv; acc[2][j] += w.z * sv; acc[3][j] += w.w * sv; }
        }
    }
    if (klane > 0) {
        float* r = red4 + ((size_t)(klane - 1) * tpr + c4) * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) r[q * 8 + j] = acc[q][j];
    }
    __syncthreads();
    if (klane == 0 && col < W) {
        float* dst = partial + (size_t)blockIdx.y * M * N;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < ns) {
                    float v = acc[q][j];
                    for (int l = 1; l < KL; ++l) v += red4[((size_t)(l - 1) * tpr + c4) * 32 + q * 8 + j];      // k-lanes in order
                    if (SMALL_N) dst[(size_t)(col + q) * N + j] = v;      // C[m = col + q][n = j]
                    else         dst[(size_t)j * N + col + q] = v;        // C[m = j][n = col + q]
                }
    }
}
// fold of the K-range partials for those launches (up to 1024 ranges, a few hundred outputs): 64 outputs x 4 range-lanes per workgroup, eight loads
// in flight per thread, lanes folded through LDS in a fixed order, then the ordinary epilogue
__global__ __launch_bounds__(256) void skinny_reduce_kernel(const float* __restrict__ partial, int splits, int M, int N, float* __restrict__ C, int ldc,
                                                            const act_gemm_epilogue_t epi) {
    __shared__ float red[3][64];
    const long long total = (long long)M * N;
    const long long i = (long long)blockIdx.x * 64 + (threadIdx.x & 63);
    const int lane = threadIdx.x >> 6;
    float v = 0.f;
    if (i < total) {
#pragma unroll 8
        for (int sp = lane; sp < splits; sp += 4) v += partial[(size_t)sp * total + i];
    }
    if (lane > 0) red[lane - 1][threadIdx.x & 63] = v;
    __syncthreads();
    if (lane == 0 && i < total) {
        v = (v + red[0][threadIdx.x]) + (red[1][threadIdx.x] + red[2][threadIdx.x]);
        const int row = (int)(i / N), col = (int)(i % N);
        v = epilogue_apply(epi, v, row, col);
        float* c = C + (size_t)row * ldc + col;
        if (epi.accumulate) v += *c;
        *c = v;
    }
}

template <int BM, int BN, int BK>
static void launch_variant(const GemmParams& p, int ak, int bk, bool vec, bool full, bool pipe, dim3 grid, hipStream_t s) {
#define L(AK, BKK, V, F, P) hipLaunchKernelGGL((sgemm_kernel<BM, BN, BK, AK, BKK, V, F, P>), grid, dim3(256), 0, s, p)
#define LV(AK, BKK) { if (full && pipe) L(AK, BKK, true, true, true); else if (full) L(AK, BKK, true, true, false); \
                      else if (vec) L(AK, BKK, true, false, false); else L(AK, BKK, false, false, false); }
    if (ak && bk)        LV(true, true)
    else if (ak && !bk)  LV(true, false)
    else if (!ak && !bk) LV(false, false)
    else                 LV(false, true)
#undef LV
#undef L
}

// ---- launch-configuration table: (a_kmajor, b_kmajor, M, N, K) -> (tile id, split-K), filled by the host-side autotuner ----------
#include <mutex>
#include <unordered_map>
namespace {
struct TuneKey { int ak, bk, M, N, K; bool operator==(const TuneKey& o) const { return ak == o.ak && bk == o.bk && M == o.M && N == o.N && K == o.K; } };
struct TuneHash { size_t operator()(const TuneKey& k) const {
    uint64_t h = 1469598103934665603ull;
    for (int v : {k.ak, k.bk, k.M, k.N, k.K}) { h ^= (uint64_t)(uint32_t)v; h *= 1099511628211ull; }
    return (size_t)h; } };
std::mutex g_tune_mu;
std::unordered_map<TuneKey, std::pair<int, int>, TuneHash> g_tune;
}  // namespace
extern "C" int act_gemm_tune_set(int ak, int bk, int M, int N, int K, int tile, int splits) {
    if (tile < 0 || (tile > 21 && (tile < 30 || tile > 36)) || splits < 0) return ACT_E_BADARG;
    std::lock_guard<std::mutex> g(g_tune_mu);
    g_tune[TuneKey{ak != 0, bk != 0, M, N, K}] = {tile, splits};
    return 0;
}
extern "C" int act_gemm_tune_get(int ak, int bk, int M, int N, int K, int* tile, int* splits) {
    std::lock_guard<std::mutex> g(g_tune_mu);
    auto it = g_tune.find(TuneKey{ak != 0, bk != 0, M, N, K});
    if (it == g_tune.end()) return 1;
    if (tile) *tile = it->second.first;
    if (splits) *splits = it->second.second;
    return 0;
}
extern "C" int act_gemm_tune_clear(void) { std::lock_guard<std::mutex> g(g_tune_mu); g_tune.clear(); return 0; }


extern "C" int act_sgemm_ex_f32(int a_kmajor, int b_kmajor, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                                float* C, int ldc, const act_gemm_epilogue_t* epi_in, float* workspace, size_t workspace_bytes,
                                int tile, int force_splits, act_stream_t stream) {
    if (!A || !B || !C) return ACT_E_NULLPTR;
    if (M < 0 || N < 0 || K < 0) return ACT_E_BADARG;
    if (M == 0 || N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    GemmParams p{};
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    if (epi_in) p.epi = *epi_in;
    else { p.epi = act_gemm_epilogue_t{}; p.epi.alpha = 1.0f; }
    if (p.epi.rowscale && p.epi.rows_per_scale <= 0) return ACT_E_BADARG;
    if ((p.epi.act == ACT_EPI_MUL_GELU_GRAD || p.epi.act == ACT_EPI_MUL_RELU_MASK) && !p.epi.aux) return ACT_E_NULLPTR;
    p.epi_vec = epilogue_is_vec(C, ldc, p.epi);

    const int kid = (a_kmajor && b_kmajor) ? KID_GEMM_NT : (a_kmajor ? KID_GEMM_NN : KID_GEMM_TN);
    ActProfScope ps(kid, s, 2.0 * M * N * (double)K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));

    if (!a_kmajor && !b_kmajor && tile == 0 && (M <= 8 || N <= 8) && K >= 2048 && workspace) {
        // wide operand as float4 rows (round 4): W % 4 == 0, 16 .. 64 threads per row, the rest of the workgroup are k-lanes
        static const bool skinny4 = [] { const char* e = getenv("ACT_GEMM_SKINNY4"); return !(e && e[0] == '0'); }();
        const bool small_n = N <= 8;
        const float* wide = small_n ? A : B; const float* skin = small_n ? B : A;
        const int ldw_ = small_n ? lda : ldb, lds_ = small_n ? ldb : lda, W = small_n ? M : N, ns = small_n ? N : M;
        if (skinny4 && W % 4 == 0 && W >= 64 && (ldw_ & 3) == 0 && (reinterpret_cast<uintptr_t>(wide) & 15) == 0) {
            int tpr_shift = 6; while ((4 << tpr_shift) > W && tpr_shift > 4) --tpr_shift;       // 64 / 32 / 16 threads per row
            const int tpr = 1 << tpr_shift, KL = 256 >> tpr_shift, gx = (W + 4 * tpr - 1) / (4 * tpr);
            int nparts = (1024 + gx - 1) / gx;                                                  // ~4 workgroups per CU
            nparts = min(nparts, max(1, K / (8 * KL)));                                         // at least 8 rows per k-lane
            nparts = min(nparts, 1024);
            const size_t fit = workspace_bytes / ((size_t)M * N * sizeof(float));
            if ((size_t)nparts > fit) nparts = (int)fit;
            if (nparts >= 1) {
                const int rpp = ((K + nparts - 1) / nparts + KL - 1) / KL * KL;
                nparts = (K + rpp - 1) / rpp;
                const size_t ldsb = (size_t)(KL - 1) * tpr * 32 * sizeof(float);
                if (small_n) hipLaunchKernelGGL(sgemm_tn_skinny4_kernel<true>, dim3(gx, nparts), dim3(256), ldsb, s, wide, ldw_, W, skin, lds_, ns, K, rpp, M, N,
                                                tpr_shift, workspace);
                else         hipLaunchKernelGGL(sgemm_tn_skinny4_kernel<false>, dim3(gx, nparts), dim3(256), ldsb, s, wide, ldw_, W, skin, lds_, ns, K, rpp, M, N,
                                                tpr_shift, workspace);
                ACT_LAUNCH_CHECK();
                hipLaunchKernelGGL(skinny_reduce_kernel, dim3((unsigned)(((long long)M * N + 63) / 64)), dim3(256), 0, s, workspace, nparts, M, N, C, ldc, p.epi);
                ACT_LAUNCH_CHECK();
                return 0;
            }
        }
        int nparts = K / 1024; if (nparts > 256) nparts = 256; if (nparts < 1) nparts = 1;
        const int rpp = ((K + nparts - 1) / nparts + 3) / 4 * 4;
        nparts = (K + rpp - 1) / rpp;
        if ((size_t)nparts * M * N * sizeof(float) <= workspace_bytes) {
            if (N <= 8) hipLaunchKernelGGL(sgemm_tn_skinny_kernel<true>, dim3((M + 63) / 64, nparts), dim3(256), 0, s, A, lda, M, B, ldb, N, K,
                                           rpp, M, N, workspace);
            else        hipLaunchKernelGGL(sgemm_tn_skinny_kernel<false>, dim3((N + 63) / 64, nparts), dim3(256), 0, s, B, ldb, N, A, lda, M, K,
                                           rpp, M, N, workspace);
            ACT_LAUNCH_CHECK();
            launch_splitk_reduce(workspace, nparts, M, N, C, ldc, p.epi, s);
            ACT_LAUNCH_CHECK();
            return 0;
        }
    }
    // vector path: every float4 is fully in range or fully out, and 16-byte aligned
    const bool vec = ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0 && (lda & 3) == 0 &&
                     (ldb & 3) == 0 && (a_kmajor ? (K & 3) == 0 : (M & 3) == 0) && (b_kmajor ? (K & 3) == 0 : (N & 3) == 0);

    // Tile shape + split-K from a small cost model: every workgroup keeps one wave per SIMD busy, so the time of a launch
    // is ~ ceil(workgroups / 256 CUs) x (work per workgroup) / (efficiency of that tile shape); split-K (deterministic
    // two-pass) is considered when the tile grid alone cannot fill the chip (weight gradients: K = tokens).
    struct Cand { int bm, bn; double eff; };
    const Cand cands[3] = {{128, 128, 1.00}, {128, 64, 0.93}, {64, 64, 0.82}};
    int BM = 128, BN = 128, splits = 1; double best = 1e300;
    for (const Cand& c : cands) {
        const long long nb = (long long)((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn);
        int maxs = 1;
        if (workspace && K >= 1024) { maxs = K / 512; if (maxs > 32) maxs = 32; if (maxs < 1) maxs = 1; }
        for (int sp = 1; sp <= maxs; sp = (sp < 4 ? sp + 1 : sp * 2)) {
            if (sp > 1 && (size_t)sp * M * N * sizeof(float) > workspace_bytes) break;
            const double rounds = (double)((nb * sp + 255) / 256);
            double cost = rounds * ((double)c.bm * c.bn * ((K + sp - 1) / sp + 32)) / c.eff;
            if (sp > 1) cost += 2.0 * (double)M * N * sp / 256.0 * 8.0;      // partial write + reduce traffic (rough)
            if (cost < best) { best = cost; BM = c.bm; BN = c.bn; splits = sp; }
            if (nb * sp >= 1024) break;
        }
    }
    // tiles 33 (128x128), 34 (64x128), 35 (64x64), 36 (128x64): NN / TN kernels with the hand-scheduled main loop (gemm_q_asm_kernel.h); 34..36 NN only;
    // bit-identical to 13 / 14 / 15 / 16
    const bool qa = tile >= 33 && tile <= 36;
    const int qa_tile = tile - 33;
    if (qa) { if (b_kmajor || (tile != 33 && !a_kmajor)) return ACT_E_BADARG; tile -= 20; }
    // tiles 30 (128x128), 31 (128x64), 32 (64x64): NT kernels with the hand-scheduled main loop (gemm_nt_asm_kernel.h); bit-identical to 10 / 11 / 12
    const bool nta = tile >= 30 && tile <= 32;
    const int nta_tile = tile - 30;
    if (nta) { if (!(a_kmajor && b_kmajor)) return ACT_E_BADARG; tile -= 29; }
    // tiles 20 (128x128), 21 (128x64): NT b128 kernels with 32-deep K tiles (full tiles, K per split % 32 == 0); bit-identical to 10 / 11
    const bool nt32 = tile == 20 || tile == 21;
    const int nt32_tile = tile - 20;
    if (nt32) { if (!(a_kmajor && b_kmajor)) return ACT_E_BADARG; tile = tile == 20 ? 1 : 2; }
    // tiles 17 (128x128), 18 (128x64): NT b128 kernels with the software-pipelined main loop (full tiles, K per split % 32 == 0)
    const bool nt16p = tile == 17 || tile == 18;
    const int nt16p_tile = tile == 17 ? 3 : 4;
    if (nt16p) { if (!(a_kmajor && b_kmajor)) return ACT_E_BADARG; tile = tile == 17 ? 1 : 2; }
    // tiles 13 (128x128), 14 (64x128), 15 (64x64), 16 (128x64): quad-fragment kernels of the NN / TN layouts (gemm_q16.hip); 14..16 NN only
    const bool q16 = tile >= 13 && tile <= 16;
    const int q16_tile = tile - 13;
    if (q16) { if (b_kmajor || (tile != 13 && !a_kmajor)) return ACT_E_BADARG; tile = tile == 13 ? 1 : 0; }
    const bool nt16 = tile >= 10 && tile <= 12;         // tiles 10..12 = tiles 1..3, NT-only b128-fragment kernel (gemm_nt16.hip)
    if (nt16) { if (!(a_kmajor && b_kmajor)) return ACT_E_BADARG; tile -= 9; }
    const bool mi16 = tile >= 7 && tile <= 9;           // tiles 7..9 = tiles 1..3 on v_mfma_f32_16x16x4_f32 (gemm16.hip)
    if (mi16) tile -= 6;
    const bool pipe = tile >= 4 && tile <= 6;           // tiles 4..6 = software-pipelined main loop of tiles 1..3
    if (pipe) tile -= 3;
    if (q16) {
        BM = (q16_tile == 1 || q16_tile == 2) ? 64 : 128; BN = q16_tile >= 2 ? 64 : 128;
        splits = force_splits >= 1 ? force_splits : 1;
        if (splits > 1 && (!workspace || (size_t)splits * M * N * sizeof(float) > workspace_bytes)) return ACT_E_BADARG;
    } else if (tile >= 1 && tile <= 3) {                // explicit configuration (autotuner)
        BM = cands[tile - 1].bm; BN = cands[tile - 1].bn;
        splits = force_splits >= 1 ? force_splits : 1;
        if (splits > 1 && (!workspace || (size_t)splits * M * N * sizeof(float) > workspace_bytes)) return ACT_E_BADARG;
    }
    p.tiles_m = (M + BM - 1) / BM; p.tiles_n = (N + BN - 1) / BN;
    static const int group_m_env = [] { const char* e = getenv("ACT_GEMM_GROUP_M"); return e ? atoi(e) : 8; }();
    p.group_m = group_m_env;
    // XCD placement of the NT b128 kernels (experiment knob ACT_GEMM_XCD_ROWS = 1 | 2 | 4: r x 8/r grid of tile blocks; default 0 = row bands)
    static const int xcd_rows_env = [] { const char* e = getenv("ACT_GEMM_XCD_ROWS"); return e ? atoi(e) : 0; }();
    p.xcd_rows = 0;
    if ((xcd_rows_env == 1 || xcd_rows_env == 2 || xcd_rows_env == 4) && p.tiles_m % xcd_rows_env == 0 && p.tiles_n % (8 / xcd_rows_env) == 0)
        p.xcd_rows = xcd_rows_env;
    static const bool fastdiv_env = [] { const char* e = getenv("ACT_GEMM_FASTDIV"); return !(e && e[0] == '0'); }();   // dev A/B knob (gemm_set_tiling)
    if (fastdiv_env) gemm_set_tiling(p);
    const long long nt = (long long)p.tiles_m * p.tiles_n;
    int kps = K;
    if (splits > 1) { kps = (K + splits - 1) / splits; kps = (kps + 31) / 32 * 32; splits = (K + kps - 1) / kps; }
    p.k_per_split = kps;
    p.partial = splits > 1 ? workspace : nullptr;
    if (K == 0) { p.k_per_split = 0; }

    dim3 grid((unsigned)nt, 1, (unsigned)splits);
    const bool full = vec && (M % BM == 0) && (N % BN == 0) && (K % 32 == 0) && (kps % 32 == 0) && K > 0;
    // 16x16x4 kernels also take an M tail when A is K-major (rows = tokens): rows clamped on load, guarded on store
    const bool full_mtail = vec && a_kmajor && (N % BN == 0) && (K % 32 == 0) && (kps % 32 == 0) && K > 0;
    if (qa) {
        if (!(full || full_mtail)) return ACT_E_BADARG;
        if ((long long)(a_kmajor ? BM : 32) * lda * 4 >= (1ll << 31) || (long long)32 * ldb * 4 >= (1ll << 31)) return ACT_E_BADARG;   // 32-bit lane offsets inside a tile
        if (!launch_sgemm_q_asm(p, qa_tile, a_kmajor, grid, s)) return ACT_E_BADARG;
    } else if (nta) {
        if (!(full || full_mtail)) return ACT_E_BADARG;
        if ((long long)BM * lda * 4 >= (1ll << 31) || (long long)BN * ldb * 4 >= (1ll << 31)) return ACT_E_BADARG;     // 32-bit lane offsets inside a tile
        launch_sgemm_nt_asm(p, nta_tile, grid, s);
    } else if (nt32) {
        if (!full) return ACT_E_BADARG;
        launch_sgemm_nt32(p, nt32_tile, grid, s);
    } else if (nt16p) {
        if (!full) return ACT_E_BADARG;
        launch_sgemm_nt16(p, nt16p_tile, grid, s);
    } else if (q16) {
        if (!(full || full_mtail)) return ACT_E_BADARG;
        if (!launch_sgemm_q16(p, q16_tile, a_kmajor, b_kmajor, grid, s)) return ACT_E_BADARG;
    } else if (nt16) {
        if (!(full || full_mtail)) return ACT_E_BADARG;
        launch_sgemm_nt16(p, BM == 128 ? (BN == 128 ? 0 : 1) : 2, grid, s);
    } else if (mi16) {
        if (!(full || full_mtail)) return ACT_E_BADARG;
        launch_sgemm16(p, BM == 128 ? (BN == 128 ? 0 : 1) : 2, a_kmajor, b_kmajor, grid, s);
    } else {
        if (BM == 128 && BN == 128) launch_variant<128, 128, 16>(p, a_kmajor, b_kmajor, vec, full, pipe, grid, s);
        else if (BM == 128)         launch_variant<128, 64, 16>(p, a_kmajor, b_kmajor, vec, full, pipe, grid, s);
        else                        launch_variant<64, 64, 16>(p, a_kmajor, b_kmajor, vec, full, pipe, grid, s);
    }
    ACT_LAUNCH_CHECK();
    if (splits > 1) {
        launch_splitk_reduce(workspace, splits, M, N, C, ldc, p.epi, s);
        ACT_LAUNCH_CHECK();
    }
    return 0;
}

extern "C" int act_sgemm_f32(int a_kmajor, int b_kmajor, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                             float* C, int ldc, const act_gemm_epilogue_t* epi_in, float* workspace, size_t workspace_bytes,
                             act_stream_t stream) {
    int tile = 0, splits = 0;
    if (act_gemm_tune_get(a_kmajor, b_kmajor, M, N, K, &tile, &splits) != 0) { tile = 0; splits = 0; }
    return act_sgemm_ex_f32(a_kmajor, b_kmajor, M, N, K, A, lda, B, ldb, C, ldc, epi_in, workspace, workspace_bytes, tile, splits, stream);
}


// the fused NT launches CAN run on the hand-scheduled main loop (K % 32 == 0; bit-identical: tests/test_gpu_dense.py).  Default OFF: measured in the
// Stage-II step (profiles/r04_fx_asm_ab.txt) conv4 (K = 512, affine + group max) 804 -> 800 us, but conv3 (K = 256, column statistics) 673 -> 723 us and
// conv2 (K = 128) 195 -> 227 us -- with 4-8 K tiles per output tile the launch is prologue / epilogue bound and three 32 KB workgroups per CU overlap
// those better than two 64-72 KB ones.  act_gemm_fx_asm(1) / ACT_GEMM_FX_ASM=1 selects them.
#include <atomic>
// bit 0: the NT launches above; bit 1: the NN launch with the epilogue-side max-pool backward term (measured 574 -> 598 us in the step: also off)
static std::atomic<int> g_fx_asm{[] { const char* e = getenv("ACT_GEMM_FX_ASM"); return e ? atoi(e) : 0; }()};
extern "C" int act_gemm_fx_asm(int on) { return on < 0 ? g_fx_asm.load() : g_fx_asm.exchange(on & 3); }

// ---- GEMM with fused producer / consumer passes (mini-PointNet, see GemmFx in gemm_common.h) ---------------------------------------
extern "C" size_t act_sgemm_fx_tile_stats_floats(int M, int N) { return (size_t)((M + 127) / 128) * 2 * N; }

extern "C" int act_sgemm_fx_f32(int a_kmajor, int b_kmajor, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                                const act_gemm_epilogue_t* epi_in, const act_gemm_fx_t* fx, float* workspace, size_t workspace_bytes,
                                act_stream_t stream) {
    if (!B || !fx) return ACT_E_NULLPTR;
    if (!A && !fx->sa_src) return ACT_E_NULLPTR;                       // (the virtual max-pool-backward operand needs no A)
    if (M <= 0 || N <= 0 || K <= 0) return ACT_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    GemmParams p{};
    p.A = A; p.B = B; p.C = C; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    if (epi_in) p.epi = *epi_in; else p.epi.alpha = 1.0f;
    p.epi_vec = C ? epilogue_is_vec(C, ldc, p.epi) : 0;
    p.fx.a_scale = fx->a_scale; p.fx.a_shift = fx->a_shift; p.fx.b_scale = fx->b_scale; p.fx.b_shift = fx->b_shift;
    p.fx.tile_stats = fx->tile_stats; p.fx.gmax = fx->gmax; p.fx.garg = fx->garg; p.fx.group = fx->group; p.fx.store_c = fx->store_c;
    p.fx.sa_src = fx->sa_src; p.fx.sa_arg = fx->sa_arg; p.fx.ep_src = fx->ep_src; p.fx.ep_arg = fx->ep_arg; p.fx.row_groups = fx->row_groups;
    if (fx->row_groups && !(a_kmajor && b_kmajor && fx->gmax && !fx->store_c && !fx->tile_stats)) return ACT_E_BADARG;   // listed groups: pooled output only
    int scatter = 0;                                                   // max-pool backward generated on load / added in the epilogue
    if (fx->sa_src || fx->sa_arg) {
        if (!fx->sa_src || !fx->sa_arg) return ACT_E_NULLPTR;
        if ((fx->group != 32 && fx->group != 64) || (lda & 3) || ((reinterpret_cast<uintptr_t>(fx->sa_src) | reinterpret_cast<uintptr_t>(fx->sa_arg)) & 15))
            return ACT_E_BADARG;
        scatter |= FX_SCATTER_A;
    }
    if (fx->ep_src || fx->ep_arg) {
        if (!fx->ep_src || !fx->ep_arg) return ACT_E_NULLPTR;
        if ((fx->group != 32 && fx->group != 64) || ((reinterpret_cast<uintptr_t>(fx->ep_src) | reinterpret_cast<uintptr_t>(fx->ep_arg)) & 15)) return ACT_E_BADARG;
        scatter |= FX_SCATTER_EPI;
    }
    static const int group_m_env = [] { const char* e = getenv("ACT_GEMM_GROUP_M"); return e ? atoi(e) : 8; }();
    p.group_m = group_m_env;
    const bool aligned = ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0 && (lda & 3) == 0 && (ldb & 3) == 0;
    if (!aligned) return ACT_E_BADARG;
    if (a_kmajor && b_kmajor) {                                        // forward conv: A-side affine + ReLU, column statistics, group max
        if (scatter) return ACT_E_BADARG;
        int mask = 0;
        if (fx->a_scale) { if (!fx->a_shift || K > 1024) return ACT_E_BADARG; mask |= FX_AFFINE_A; }
        if (fx->tile_stats) mask |= FX_COLSTATS;
        if (fx->gmax) { if (fx->group != 32 && fx->group != 64) return ACT_E_BADARG; mask |= FX_GROUPMAX; if (!fx->store_c) mask |= FX_NOSTORE; }
        if (mask == 0 || (M % 128) || (K % 16) || (N % 64)) return ACT_E_BADARG;
        if (!(mask & FX_NOSTORE) && !C) return ACT_E_NULLPTR;
        const int tile = (N % 128 == 0) ? 0 : 1, BN = tile == 0 ? 128 : 64;
        p.tiles_m = M / 128; p.tiles_n = N / BN; p.k_per_split = K; p.partial = nullptr;
        gemm_set_tiling(p);
        ActProfScope ps(KID_GEMM_NT, s, 2.0 * M * N * (double)K, 4.0 * ((double)M * K + (double)N * K + ((mask & FX_NOSTORE) ? 0.0 : (double)M * N)));
        // hand-scheduled main loop when every K tile is 32 deep (ACT_GEMM_FX_ASM=0: the compiler-scheduled kernels, for A/B runs); same bits either way
        const dim3 fgrid((unsigned)(p.tiles_m * p.tiles_n));
        if (!((g_fx_asm.load() & 1) && !fx->row_groups && launch_sgemm_nt_asm_fx(p, tile, mask, fgrid, s)) && !launch_sgemm_nt16_fx(p, tile, mask, fgrid, s)) return ACT_E_BADARG;
        ACT_LAUNCH_CHECK();
        return 0;
    }
    if (a_kmajor && !b_kmajor) {                                       // input gradient dX = dY . W with dY and / or one term of dX being a max-pool backward
        if (!scatter || !C) return ACT_E_BADARG;
        if ((M % 128) || (N % 128) || (K % 16) || (M % fx->group)) return ACT_E_BADARG;
        if ((scatter & FX_SCATTER_EPI) && ((ldc & 3) || (reinterpret_cast<uintptr_t>(C) & 15))) return ACT_E_BADARG;
        p.tiles_m = M / 128; p.tiles_n = N / 128; p.k_per_split = K; p.partial = nullptr;
        gemm_set_tiling(p);
        ActProfScope ps(KID_GEMM_NN, s, 2.0 * M * N * (double)K, 4.0 * ((double)N * K + (double)M * N));
        const dim3 qgrid((unsigned)(p.tiles_m * p.tiles_n));
        if (!((g_fx_asm.load() & 2) && launch_sgemm_q_asm_fx(p, 1, scatter, qgrid, s)) && !launch_sgemm_q16_fx(p, 1, scatter, qgrid, s)) return ACT_E_BADARG;
        ACT_LAUNCH_CHECK();
        return 0;
    }
    if (!a_kmajor && !b_kmajor) {                                      // weight gradient: B activated on load and / or A = max-pool backward on load
        if (scatter & FX_SCATTER_EPI) return ACT_E_BADARG;
        int mask = scatter;
        if (fx->b_scale || fx->b_shift) { if (!fx->b_scale || !fx->b_shift) return ACT_E_NULLPTR; mask |= FX_AFFINE_B; }
        if (!mask || !C) return ACT_E_NULLPTR;
        if ((M % 128) || (N % 128) || (K % 32) || (scatter && (K % fx->group))) return ACT_E_BADARG;
        p.tiles_m = M / 128; p.tiles_n = N / 128;
        gemm_set_tiling(p);
        const long long nt = (long long)p.tiles_m * p.tiles_n;
        int splits = (int)((768 + nt - 1) / nt);                       // ~3 workgroups per CU; K = rows of the batch, a few hundred thousand
        if (splits > 256) splits = 256;                                // (the 256 x 128 gradient of the second conv is 2 tiles: 64 ranges left half the chip idle)
        while (splits > 1 && (K / splits < 256 || (size_t)splits * M * N * sizeof(float) > workspace_bytes)) --splits;
        int kps = (K + splits - 1) / splits; kps = (kps + 31) / 32 * 32; splits = (K + kps - 1) / kps;
        if (splits > 1 && !workspace) return ACT_E_NULLPTR;
        p.k_per_split = kps; p.partial = splits > 1 ? workspace : nullptr;
        ActProfScope ps(KID_GEMM_TN, s, 2.0 * M * N * (double)K, 4.0 * ((scatter ? 0.0 : (double)M * K) + (double)N * K + (double)M * N));
        if (!launch_sgemm_q16_fx(p, 0, mask, dim3((unsigned)nt, 1, (unsigned)splits), s)) return ACT_E_BADARG;
        ACT_LAUNCH_CHECK();
        if (splits > 1) {
            launch_splitk_reduce(workspace, splits, M, N, C, ldc, p.epi, s);
            ACT_LAUNCH_CHECK();
        }
        return 0;
    }
    return ACT_E_BADARG;
}
