/* act_hip.h -- C ABI of libact_hip.so: the MI355X (gfx950) kernels behind ACT's
 * masked-point-modeling hot path.
 *
 * Every entry point takes plain device pointers + sizes + a hipStream_t (as void*),
 * allocates nothing, never synchronises, and returns 0 or a hipError_t / negative
 * ACT_E* code.  All tensors are dense row-major fp32 unless stated; indices are
 * int32 (FPS, Chamfer) or int64 (kNN), exactly as the reference operator returns them.
 * Citations are file:line in the reference tree (RunpeiDong/ACT @ 2024_08_07).
 */
#ifndef ACT_HIP_H
#define ACT_HIP_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* act_stream_t;            /* hipStream_t */

#define ACT_E_BADARG   (-1)            /* shape / size outside what the kernel supports */
#define ACT_E_NULLPTR  (-2)

/* ---- library ------------------------------------------------------------------------------ */
int         act_version(void);
const char* act_arch(void);             /* "gfx950" */

/* ---- live per-kernel timing (hipEvents on the launch stream; used by bench.py roofline) ---- */
int         act_prof_enable(int on);    /* returns previous state */
int         act_prof_reset(void);
int         act_prof_num_kernels(void);
const char* act_prof_kernel_name(int id);
/* synchronises the recorded events; totals since the last reset */
int         act_prof_read(int id, double* total_ms, long long* launches, double* flops, double* bytes);

/* ---- point operators ---------------------------------------------------------------------- */
/* pointnet2_ops.furthest_point_sample + gather_operation as used by misc.fps
 * (utils/misc.py:39-46; call site models/dvae.py:170).  xyz [B,N,3] -> idx int32 [B,G]
 * (idx[:,0]==0, lowest-index tie-break) and, if centers_out != NULL, centers [B,G,3]. */
size_t act_fps_scratch_floats(int B, int N);   /* 0 for N <= 16384 (the cloud lives in registers / LDS), else B*N running distances */
int act_fps_f32(const float* xyz, int B, int N, int G, int32_t* idx_out, float* centers_out,
                int skip_near_origin, float* scratch /* act_fps_scratch_floats floats, NULL when that is 0 */, act_stream_t stream);
/* Measurement infrastructure (bench.py: group_fps_knn.fps_chain), not a product entry point: latency in microseconds per iteration of the four dependent
   phases of one FPS step -- {distance evaluations, wave arg-max, cross-wave arg-max (LDS + barrier), winner's coordinates} -- and of the whole step, each
   timed as `iters` dependent repetitions in ONE workgroup of the launch configuration act_fps_f32 uses for clouds of N points (N <= 8192).  Synchronises. */
int act_fps_chain_probe(int N, int iters, double* us_per_iteration /* [5] */, int* waves_out, int* points_per_lane_out, act_stream_t stream);

/* knn_cuda.KNN(k).forward fused with Group's gather + centre subtraction
 * (models/dvae.py:159,172-182; DGCNN graph k=4 models/dvae.py:23,68).
 * ref [B,N,3], query [B,Q,3], any K <= N (K <= 64 and N <= 8192: register-resident fast path) -> idx int64 ([B,Q,K], or [B,K,Q] when idx_kq != 0:
 * transpose_mode=False layout), ascending (distance, index).
 * nbr_out  (nullable) [B,Q,K,3] = ref[idx] - query      dist_out (nullable) sqrt distance, idx layout. */
int act_knn_group_f32(const float* ref, const float* query, int B, int N, int Q, int K,
                      int64_t* idx_out, int idx_kq, float* nbr_out, float* dist_out, act_stream_t stream);

/* pointnet2_ops.gather_operation (utils/misc.py:45): feat [B,C,N], idx int32 [B,S] -> out [B,C,S];
 * backward scatter-add: grad_out [B,C,S] -> grad_feat [B,C,N] (zero-filled here, deterministic). */
int act_gather_points_f32(const float* feat, const int32_t* idx, int B, int C, int N, int S, float* out, act_stream_t stream);
int act_gather_points_bwd_f32(const float* grad_out, const int32_t* idx, int B, int C, int N, int S, float* grad_feat,
                              act_stream_t stream);

/* PointcloudScaleAndTranslate (datasets/data_transforms.py:20-34), in place:
 * pc[b,n,:] = pc[b,n,:] * scale[b,:] + shift[b,:]   (mul then add, no FMA). */
int act_scale_translate_f32(float* pc, const float* scale, const float* shift, int B, int N, act_stream_t stream);

/* PointcloudRotate (datasets/data_transforms.py:6-18), in place: pc[b,n,:] = pc[b,n,:] @ R[b] with R[b] row-major [3,3]
 * (the reference builds R = [[c,0,s],[0,1,0],[-s,0,c]] per sample on the host; any per-sample 3x3 is accepted here). */
int act_rotate_points_f32(float* pc, const float* rot, int B, int N, act_stream_t stream);

/* ---- Chamfer distance (extensions/chamfer_dist: chamfer_cuda.cpp:12-39, chamfer.cu:15-229) --- */
/* forward: xyz1 [B,n,3], xyz2 [B,m,3] -> dist1 [B,n], dist2 [B,m] (squared), idx1 int32 [B,n], idx2 int32 [B,m] */
int act_chamfer_fwd_f32(const float* xyz1, const float* xyz2, int B, int n, int m,
                        float* dist1, float* dist2, int32_t* idx1, int32_t* idx2, act_stream_t stream);
/* same with the rounding of the distance selectable: fma_contract = 0 rounds every product and sum (act_chamfer_fwd_f32, the oracle
 * convention); fma_contract = 1 evaluates chamfer.cu:43-57's x2*x2 + y2*y2 + z2*z2 as fma(z2, z2, fma(x2, x2, y2*y2)), the form an
 * FMA-contracting build (nvcc's default) of the reference makes of it -- for comparing against outputs of a real CUDA build: distances agree
 * to 2 ulps, the arg-min index may differ on (near) ties. */
int act_chamfer_fwd_ex_f32(const float* xyz1, const float* xyz2, int B, int n, int m,
                           float* dist1, float* dist2, int32_t* idx1, int32_t* idx2, int fma_contract, act_stream_t stream);
/* backward: deterministic gather formulation of the reference's atomicAdd scatter */
int act_chamfer_bwd_f32(const float* xyz1, const float* xyz2, const int32_t* idx1, const int32_t* idx2,
                        const float* grad_dist1, const float* grad_dist2, int B, int n, int m,
                        float* grad_xyz1, float* grad_xyz2, act_stream_t stream);


/* ---- dense fp32 GEMM on the matrix cores with fused epilogue -------------------------------- */
/* C[M,N] (+)= epilogue( alpha * opA(A)[M,K] . opB(B)[K,N] )
 *   a_kmajor=1: A stored [M][K] (lda>=K)   a_kmajor=0: A stored [K][M] (lda>=M)
 *   b_kmajor=1: B stored [N][K] (ldb>=K)   b_kmajor=0: B stored [K][N] (ldb>=N)
 * nn.Linear / Conv1d(k=1) forward = (1,1) with B = weight [out,in] (models/act.py:25-69, models/dvae.py:185-215);
 * input gradient = (1,0); weight gradient = (0,0) with K = number of rows (split-K through `workspace`).
 * epilogue order: v = alpha*acc; v += bias[col]; activation; v *= rowscale[row / rows_per_scale]; v += res[row / res_row_div, col];
 * if accumulate: v += C[row,col].   (rowscale = DropPath gate/keep per sample, res = residual stream.)
 *   ACT_EPI_GELU          : if aux != NULL the pre-activation is stored to aux[row,col]; v = gelu_erf(v)
 *   ACT_EPI_MUL_GELU_GRAD : v *= gelu'(aux[row,col])        ACT_EPI_MUL_RELU_MASK: v = aux[row,col] > 0 ? v : 0
 * workspace (nullable): scratch for split-K partials, workspace_bytes long; never allocated here. */
enum { ACT_EPI_NONE = 0, ACT_EPI_GELU = 1, ACT_EPI_RELU = 2, ACT_EPI_MUL_GELU_GRAD = 3, ACT_EPI_MUL_RELU_MASK = 4 };
typedef struct {
    float        alpha;            /* 1.0f */
    int          act;              /* ACT_EPI_* */
    int          accumulate;       /* C += result */
    int          rows_per_scale;   /* rows sharing one rowscale entry (tokens per sample) */
    int          ldr, ldaux;       /* leading dimensions of res / aux */
    int          res_row_div;      /* res row = row / res_row_div (0/1: per row; n: one res row per group of n rows) */
    const float* bias;             /* [N] or NULL */
    const float* rowscale;         /* [ceil(M / rows_per_scale)] or NULL */
    const float* res;              /* [M, ldr] or NULL */
    float*       aux;              /* [M, ldaux] or NULL */
} act_gemm_epilogue_t;
int act_sgemm_f32(int a_kmajor, int b_kmajor, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                  float* C, int ldc, const act_gemm_epilogue_t* epilogue, float* workspace, size_t workspace_bytes,
                  act_stream_t stream);
/* same, with an explicit launch configuration: tile 1 = 128x128, 2 = 128x64, 3 = 64x64 workgroup tile, 4..6 = the same tiles with the
 * software-pipelined (3-stage LDS, mid-tile barrier) main loop, 7..9 = the same tiles on v_mfma_f32_16x16x4_f32,
 * 10..12 = NT-only kernels with ds_read_b128 operand fragments (aligned shapes only), 13 = 128x128 / 14 = 64x128 / 15 = 64x64 / 16 = 128x64
 * quad-fragment kernels of the NN (a_kmajor=1, b_kmajor=0) and TN (0,0; 13 only) layouts (17 = 128x128 / 18 = 128x64: NT with the software-pipelined
 * main loop, full tiles only, bit-identical to 10 / 11; 20 = 128x128 / 21 = 128x64: NT with 32-deep K tiles, full tiles only, bit-identical to 10 / 11): ds_read_b128 along the row-contiguous operand, no transpose (0: built-in cost model), splits >= 1 = split-K factor.  Used by the host-side autotuner (act_amd/kernels.py), results are identical up to
 * the fp32 summation order of split-K. */
int act_sgemm_ex_f32(int a_kmajor, int b_kmajor, int M, int N, int K, const float* A, int lda, const float* B, int ldb,
                     float* C, int ldc, const act_gemm_epilogue_t* epilogue, float* workspace, size_t workspace_bytes,
                     int tile, int splits, act_stream_t stream);

/* Several weight-gradient GEMMs of one module in ONE launch (csrc/gemm_grouped.hip): C_p[M_p,N_p] = A_p^T . B_p with A_p stored [K][M_p] (the
 * output gradient dY of a Linear, lda >= M_p), B_p stored [K][N_p] (the Linear's input X), the same K (token rows) for every problem -- the
 * dW = dY^T . X of models/act.py:25-69 for all Linears of a Transformer block -- plus, where bias_out != NULL, db_p[M_p] = column sums of A_p
 * (the bias gradient), accumulated by the workgroups that stage A_p anyway.  M_p, N_p multiples of 128, K a multiple of 16, operands 16-byte
 * aligned, at most 8 problems.  splits = number of K ranges (0: about two workgroups per CU, act_sgemm_tn_grouped_splits); for splits > 1 the
 * partials go through `workspace` (act_sgemm_tn_grouped_workspace bytes) and ONE reduction launch folds them in ascending K order
 * (deterministic).  With the same split factor C_p is bit-identical to act_sgemm_ex_f32(0, 0, ..., tile 13, splits). */
typedef struct {
    const float* A; int lda;
    const float* B; int ldb;
    float* C; int ldc;
    int M, N;
    float* bias_out;               /* [M] or NULL */
} act_gemm_tn_problem_t;
size_t act_sgemm_tn_grouped_workspace(const act_gemm_tn_problem_t* probs, int nprob, int K, int splits);
int act_sgemm_tn_grouped_splits(const act_gemm_tn_problem_t* probs, int nprob, int K);
int act_sgemm_tn_grouped_f32(const act_gemm_tn_problem_t* probs, int nprob, int K, int splits, float* workspace, size_t workspace_bytes,
                             act_stream_t stream);

/* GEMM with the producer / consumer passes of the mini-PointNet fused in (models/dvae.py:201-215: Conv1d -> BatchNorm1d -> ReLU -> Conv1d -> max):
 *  (1,1) forward conv:  a_scale/a_shift [K] (nullable, K <= 1024): A'[r,k] = max(0, A[r,k]*a_scale[k] + a_shift[k]) applied while A is staged --
 *        the BatchNorm + ReLU of the producing layer, whose output tensor then never exists;  tile_stats (nullable): per 128-row tile the
 *        column mean and sum of squared deviations of the stored values, [M/128][2][N] (act_sgemm_fx_tile_stats_floats), input of
 *        act_bn_tiles_finalize_f32 -- no statistics pass over the output;  gmax (nullable) [M/group][N] (+ garg int32, first arg-max): max over
 *        every `group` (32 | 64) consecutive rows, the max-pool over the points of a group;  store_c = 0: C itself is not written.
 *        Needs M % 128 == 0, N % 64 == 0, K % 16 == 0, 16-byte aligned operands.
 *  (0,0) weight gradient: b_scale/b_shift [N]: B'[k,n] = max(0, B[k,n]*b_scale[n] + b_shift[n]) applied while B is staged (M, N % 128 == 0, K % 32 == 0,
 *        deterministic split-K through `workspace`).
 *  backward of the max-pool (torch.max(feature, dim=2), models/dvae.py:211,214: the gradient goes to the arg-max row of every group only):
 *        sa_src [R/group][C] + sa_arg int32 (nullable pair): the A operand is VIRTUAL, A[r][c] = sa_arg[r/group][c] == r % group ? sa_src[r/group][c] : 0,
 *        generated while it is staged (A may be NULL, lda = C) -- for the input gradient (1,0: rows r = M, c = K) and the weight gradient
 *        (0,0: r = K, c = M) of the conv in front of the pool, so the scattered [R][C] gradient tensor never exists;  ep_src / ep_arg
 *        [M/group][N] (nullable pair, (1,0) only): C[r][c] += ep_arg[r/group][c] == r % group ? ep_src[r/group][c] : 0 in the epilogue (the
 *        second path into the tensor in front of the first pool).  group 32 | 64, M % 128 == 0, N % 128 == 0, C % 4 == 0.
 * The epilogue of a fused launch takes alpha / bias / rowscale / res (incl. res_row_div) / accumulate but NO activation (epi->act must be
 * ACT_EPI_NONE, else ACT_E_BADARG): the mini-PointNet applies BatchNorm + ReLU while the NEXT layer stages its operand. */
typedef struct {
    const float *a_scale, *a_shift, *b_scale, *b_shift;
    float*   tile_stats;
    float*   gmax;
    int32_t* garg;
    int      group, store_c;
    const float*   sa_src;
    const int32_t* sa_arg;
    const float*   ep_src;
    const int32_t* ep_arg;
    const int32_t* row_groups;   /* (1,1) with gmax and store_c = 0 only, nullable: the M / group row groups of A are the LISTED groups of a larger
                                  * tensor -- virtual row m is row row_groups[m / group] * group + m % group of A, and the pooled row m / group is
                                  * written to row row_groups[m / group] of gmax / garg (Stage II: the patch embedding is only needed for the
                                  * visible patches, models/act.py:269-275).  A group may be listed twice (padding M to a multiple of 128). */
} act_gemm_fx_t;
size_t act_sgemm_fx_tile_stats_floats(int M, int N);
/* Which fused launches run on the hand-scheduled main loops (csrc/gemm_nt_asm_kernel.h, gemm_q_asm_kernel.h) instead of the compiler-scheduled kernels;
 * bit-identical either way.  Bit 0: the (1,1) launches with K % 32 == 0 (default off: not faster at K <= 512); bit 1: the (1,0) launch with the
 * epilogue-side ep_src / ep_arg term only (default off: 574 -> 598 us at K = 512).  on >= 0 sets the mask (returns the previous one), on < 0 queries; initial value: env
 * ACT_GEMM_FX_ASM (default 0). */
int act_gemm_fx_asm(int on);
int act_sgemm_fx_f32(int a_kmajor, int b_kmajor, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                     const act_gemm_epilogue_t* epilogue, const act_gemm_fx_t* fx, float* workspace, size_t workspace_bytes, act_stream_t stream);
/* BatchNorm (train mode) statistics from those tile partials: mean, rstd, scale = gamma*rstd, shift = beta - mean*scale, running stats updated
 * in place when non-NULL (what act_bn_stats_f32 produces from a pass over the tensor). */
int act_bn_tiles_finalize_f32(const float* tile_stats, int tiles, int rows_per_tile, int C, const float* gamma, const float* beta, float eps,
                              float momentum, float* running_mean, float* running_var, float* mean, float* rstd, float* scale, float* shift,
                              act_stream_t stream);

/* ---- row-wise fused kernels of a Transformer block ------------------------------------------- */
/* xin = x + pos (pos nullable); y = LayerNorm(xin) * gamma + beta  (models/act.py:87-90 with the
 * `x = blk(x + pos)` of :109-112,140-143 fused in).  xin_out / mean / rstd are nullable. D % 4 == 0, D <= 2048. */
int act_layernorm_fwd_f32(const float* x, const float* pos, const float* gamma, const float* beta, float* xin_out,
                          float* y, float* mean, float* rstd, int T, int D, float eps, act_stream_t stream);
/* Frozen-teacher prompt rows (models/dvae.py:485-498,556-566): y[b*P+p,:] = LN(dropout(tok[p,:]) + ppos[p,:]) * gamma + beta,
 * inverted dropout with rate drop_p, keep mask from Philox4x32-10 keyed by (seed, row, column/4); tok, ppos [P,D]; y [B*P,D].
 * seed_dev (nullable): device-resident 64-bit step counter mixed into the key, so a captured hipGraph draws fresh noise per replay. */
int act_prompt_layernorm_fwd_f32(const float* tok, const float* ppos, int B, int P, int D, float drop_p, uint64_t seed,
                                 const uint64_t* seed_dev, const float* gamma, const float* beta, float eps, float* y,
                                 act_stream_t stream);
/* Prompt rows of a trained prompt layer (Stage I): y[b*P+p,:] = dropout(tok[p,:]) + ppos[p,:]; keep mask = `mask` (0/1 floats [B*P, D], nullable)
 * or Philox keyed by (seed, row, column/4) as above; backward: dppos[p,:] = sum_b dy, dtok[p,:] = sum_b dy * keep / (1 - drop_p), fixed order. */
int act_prompt_rows_fwd_f32(const float* tok, const float* ppos, const float* mask, int B, int P, int D, float drop_p, uint64_t seed,
                            float* y, act_stream_t stream);
int act_prompt_rows_bwd_f32(const float* dy, const float* mask, int B, int P, int D, float drop_p, uint64_t seed, float* dtok, float* dppos,
                            act_stream_t stream);
/* dx = dres (nullable, residual-stream gradient) + LayerNorm backward of dy; dgamma/dbeta (nullable) summed over
 * rows in a fixed order through `workspace` (act_layernorm_bwd_workspace bytes). */
size_t act_layernorm_bwd_workspace(int T, int D);
int act_layernorm_bwd_f32(const float* dy, const float* xin, const float* gamma, const float* mean, const float* rstd,
                          const float* dres, float* dx, float* dgamma, float* dbeta, int accumulate_params,
                          float* workspace, size_t workspace_bytes, int T, int D, act_stream_t stream);
/* out[c] (+)= sum_r in[r, c]  (bias gradients); deterministic two-stage. */
size_t act_colsum_workspace(int R, int C);
int act_colsum_f32(const float* in, int R, int C, int ld, float* out, int accumulate, float* workspace,
                   size_t workspace_bytes, act_stream_t stream);

/* Fused multi-head self-attention (models/act.py:57-69): qkv [B,S,3,H,hd] packed as the qkv Linear writes it,
 * out [B,S,H*hd] as the proj Linear reads it, lse [B,H,S] (nullable) = log-sum-exp of the scaled scores.
 * forward: any S (keys streamed through LDS in chunks of 128 with an online softmax), hd in {32,64}.  backward (recomputes P from lse; 128-key x 64-query LDS chunks): any S. */
int act_attention_fwd_f32(const float* qkv, float* out, float* lse, int B, int S, int H, int head_dim, float scale,
                          act_stream_t stream);
/* prefix variant: keys/values = S0 rows of kv0 [B,S0,2,H,hd] followed by the Sq rows of qkv1 [B,Sq,3,H,hd]; queries = qkv1.
 * Used for the prompt-tuned teacher (models/dvae.py:536-576): prompt tokens are keys/values only. */
int act_attention_fwd_prefix_f32(const float* kv0, int S0, const float* qkv1, int Sq, float* out, float* lse, int B,
                                 int H, int head_dim, float scale, act_stream_t stream);
int act_attention_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                          int B, int S, int H, int head_dim, float scale, act_stream_t stream);
/* backward of the prefix variant: dqkv1 [B,Sq,3,H,hd] receives dQ and the dK/dV of the Sq own rows, dkv0 [B,S0,2,H,hd] the
 * dK/dV of the prefix rows (gradients of the learnable prompts of Stage I, models/dvae.py:420-437). */
int act_attention_bwd_prefix_f32(const float* kv0, int S0, const float* qkv1, int Sq, const float* out, const float* dout,
                                 const float* lse, float* dkv0, float* dqkv1, int B, int H, int head_dim, float scale,
                                 act_stream_t stream);

/* Cosine distillation loss (models/act.py:1243-1254; lightly NegativeCosineSimilarity(dim=1, eps=1e-8)):
 * loss = mean over rows of 1 - cos(student_r, teacher_r).  row_loss [R], stats [R,3] are scratch kept for backward. */
int act_cosine_loss_fwd_f32(const float* student, const float* teacher, int R, int D, float eps, float* loss_out,
                            float* row_loss, float* stats, act_stream_t stream);
int act_cosine_loss_bwd_f32(const float* student, const float* teacher, const float* stats, const float* grad_loss,
                            int R, int D, float eps, float* grad_student, act_stream_t stream);

/* Alternative distillation losses (models/act.py:1186-1191,1255): kind 0 = 'l2' (nn.MSELoss, mean), kind 1 = 'smoothl1'
 * (nn.SmoothL1Loss, mean, beta 1) over [R,D]; row_loss [R] scratch. */
int act_regression_loss_fwd_f32(const float* student, const float* teacher, int R, int D, int kind, float* loss_out,
                                float* row_loss, act_stream_t stream);
int act_regression_loss_bwd_f32(const float* student, const float* teacher, const float* grad_loss, int R, int D, int kind,
                                float* grad_student, act_stream_t stream);

/* Classification loss of the finetune path (models/act.py:823-830: nn.CrossEntropyLoss(), mean over rows):
 * logits [R,C], labels int64 [R] -> loss_out[0] = mean_r (logsumexp(logits_r) - logits_r[label_r]); row_buf [3,R] =
 * {logsumexp (kept for backward), row loss, arg-max==label flag}; acc_out (nullable) [1] = fraction of rows whose arg-max
 * (lowest index on ties) equals the label.
 * backward: grad_logits = grad_loss * (softmax(logits) - onehot(label)) / R. */
int act_softmax_xent_fwd_f32(const float* logits, const int64_t* labels, int R, int C, float* loss_out, float* row_buf,
                             float* acc_out, act_stream_t stream);
int act_softmax_xent_bwd_f32(const float* logits, const int64_t* labels, const float* row_buf, const float* grad_loss,
                             int R, int C, float* grad_logits, act_stream_t stream);

/* ---- mini-PointNet / FoldingNet row kernels (models/dvae.py:185-275), rows = points, columns = channels ------ */
/* train-mode BatchNorm1d statistics over all R rows: mean, rstd, scale = gamma*rstd, shift = beta - mean*scale;
 * running stats updated in place when non-NULL (momentum, unbiased variance).  workspace: act_colstats_workspace. */
size_t act_colstats_workspace(int R, int C);
int act_bn_stats_f32(const float* x, int R, int C, const float* gamma, const float* beta, float eps, float momentum,
                     float* running_mean, float* running_var, float* mean, float* rstd, float* scale, float* shift,
                     float* workspace, size_t workspace_bytes, act_stream_t stream);
/* y = relu?(x * scale[c] + shift[c])    (BatchNorm apply, eval or train) */
int act_affine_act_f32(const float* x, const float* scale, const float* shift, int relu, int R, int C, float* y,
                       act_stream_t stream);
/* BatchNorm(+ReLU) backward in train mode: dy w.r.t. the activation output, x = BN input -> dx, dgamma, dbeta */
int act_bn_bwd_f32(const float* x, const float* dy, const float* scale, const float* shift, const float* mean,
                   const float* rstd, int relu, int R, int C, float* dx, float* dgamma, float* dbeta,
                   float* workspace, size_t workspace_bytes, act_stream_t stream);
/* the same backward when dy is known to be zero on whole groups of n consecutive rows: rows r with live[r / n] == 0 are treated as dy = 0 and dy is not
 * read for them (bit-identical to act_bn_bwd_f32 on a dy that holds zeros there: the skipped terms are exact zeros).  live == NULL: act_bn_bwd_f32. */
int act_bn_bwd_groups_f32(const float* x, const float* dy, const float* scale, const float* shift, const float* mean, const float* rstd, int relu,
                          int R, int C, const int32_t* live, int n, float* dx, float* dgamma, float* dbeta, float* workspace,
                          size_t workspace_bytes, act_stream_t stream);
/* SyncBatchNorm building blocks (the statistics are all-reduced across ranks by the host between the calls, tools/runner_pretrain.py:86-88):
 * per-column mean / biased variance of this rank's rows; backward sums of this rank's rows; dx from the global sums over `count` rows. */
int act_col_mean_var_f32(const float* x, int R, int C, float* mean, float* var, float* workspace, size_t workspace_bytes, act_stream_t stream);
int act_bn_bwd_sums_f32(const float* x, const float* dy, const float* scale, const float* shift, const float* mean, const float* rstd,
                        int relu, int R, int C, float* sum_dy, float* sum_dy_xhat, float* workspace, size_t workspace_bytes,
                        act_stream_t stream);
int act_bn_bwd_apply_f32(const float* x, const float* dy, const float* scale, const float* shift, const float* mean, const float* rstd,
                         const float* sum_dy, const float* sum_dy_xhat, float count, int relu, int R, int C, float* dx, act_stream_t stream);
/* torch.max over the n points of each group: in [G*n, C] -> out [G,C], arg int32 [G,C] (first maximum; nullable) */
int act_group_max_f32(const float* in, int G, int n, int C, float* out, int32_t* arg, act_stream_t stream);
int act_group_max_bwd_f32(const float* dout, const int32_t* arg, int G, int n, int C, int accumulate, float* din,
                          act_stream_t stream);
/* The two products of Encoder.backward whose left operand is the gradient of that max (models/dvae.py:209-216, :262-275: the conv in front of
 * torch.max(feature, dim=2)).  dh[g*n + j][c] = (arg[g][c] == j ? dout[g][c] : 0) has one non-zero per (group, channel); these walk the live
 * entries instead of running a dense [G*n, C] GEMM operand (csrc/pool_bwd.hip):
 *   matmul: dx[g*n + j][0:N] = sum over {c : arg[g][c] == j} dout[g][c] * w[c][0:N]        (w [C, N] row-major; N in {256, 512, 1024})
 *   wgrad:  dw[c][0:N]       = sum over g of dout[g][c] * act(x[g*n + arg[g][c]][0:N])     (x [G*n, N]; act = relu(x * scale + shift) per
 *           column when scale/shift are given, identity when both are NULL; N % 64 == 0, C % 128 == 0)
 * n <= 64 and 256 % n == 0; 16-byte aligned operands, leading dimensions multiples of 4.  The wgrad splits the groups over workgroups and folds
 * the partial sums in order (workspace bytes from act_group_max_bwd_wgrad_workspace; a smaller workspace only lowers the split count). */
int act_group_max_bwd_matmul_f32(const float* dout, const int32_t* arg, int G, int n, int C, const float* w, int ldw, int N, float* dx, int lddx,
                                 act_stream_t stream);
/* live[g] = (d[g][0:C] has a non-zero entry).  In Stage II the patch embedding runs on all B*G patches but only the visible ones feed the loss, so
 * 80 % of the rows of its output gradient are exactly zero (models/act.py:269-275); the kernels below skip those groups (the wgrad builds its own list).
 * matmul_live: the n rows of a group with live[g] == 0 are NOT written -- for a consumer that takes the same list (act_bn_bwd_groups_f32). */
int act_group_live_i32(const float* d, int G, int C, int32_t* live, act_stream_t stream);
int act_group_max_bwd_matmul_live_f32(const float* dout, const int32_t* arg, int G, int n, int C, const float* w, int ldw, int N, float* dx, int lddx,
                                      const int32_t* live, act_stream_t stream);
size_t act_group_max_bwd_wgrad_workspace(int G, int n, int C, int N);
int act_group_max_bwd_wgrad_f32(const float* dout, const int32_t* arg, int G, int n, int C, const float* x, int ldx, int N, const float* scale,
                                const float* shift, float* dw, int lddw, float* workspace, size_t workspace_bytes, act_stream_t stream);
/* out[g,c] = sum over the n rows of group g (gradient of a per-group broadcast add) */
int act_group_sum_f32(const float* in, int G, int n, int C, float* out, act_stream_t stream);

/* ---- DGCNN token mixer + dVAE tokenizer glue (models/dvae.py:26-117, 587-588) -------------------------------- */
/* Edge-conv tail.  yz [B*G, ldy] holds Y = Wa.x at column 0 and (zoff >= 0) Z = (Wb-Wa).x at column zoff; idx int64
 * [B,k,G] (KNN transpose_mode=False layout; NULL: no gather, k must be 1).  out[b*G+g, ooff + c] =
 * max_j LeakyReLU(GroupNorm_groups(Y[b, idx[b,j,g], c] + Z[b,g,c])).  stats: scratch [18*B*groups]. */
int act_edge_gn_lrelu_max_f32(const float* yz, int ldy, int zoff, const int64_t* idx, int B, int G, int k, int C,
                              int groups, const float* gamma, const float* beta, float eps, float slope, float* stats,
                              float* out, int ldo, int ooff, act_stream_t stream);
/* backward of act_edge_gn_lrelu_max_f32 (DGCNN backward of Stage I): stats = the mean/rstd the forward left in its
 * stats buffer; dout [B*G, ldd]; dyz [B*G, ldy] receives dY (columns 0..C) and dZ (columns zoff..zoff+C; idx == NULL: k = 1 head,
 * dY only); part [2][B][C] receives the per-sample partial sums of dgamma / dbeta (column-sum them for the parameter
 * gradients); mstat [2][B*groups] scratch.  Deterministic (no atomics). */
int act_edge_gn_lrelu_max_bwd_f32(const float* yz, int ldy, int zoff, const int64_t* idx, int B, int G, int k, int C, int groups,
                                  const float* gamma, const float* beta, const float* stats, float slope,
                                  const float* dout, int ldd, float* dyz, float* part, float* mstat, act_stream_t stream);
/* runtime switch of the graph-layer passes of that backward (A/B and identity tests): 1 (default) = LDS-resident slabs + gather over an inverse
 * adjacency where G <= 128 (round 6), 0 = the global-gather / scatter-image kernels; on < 0 only reads.  Returns the previous value. */
int act_edge_bwd_lds(int on);
/* Tokenizer head: logits = LeakyReLU(GroupNorm(h [B*G, C])); index = argmax_c((logits + gumbel) / tau);
 * out [B*G, D] = codebook[index]  (F.gumbel_softmax(hard=True) + einsum with the codebook, models/dvae.py:587-588).
 * noise [B*G, C] (nullable: Philox4x32-10 keyed by seed); index_out / logits_out nullable. */
int act_gn_gumbel_argmax_gather_f32(const float* h, int B, int G, int C, int groups, const float* gamma, const float* beta,
                                    float eps, float slope, const float* noise, uint64_t seed, const uint64_t* seed_dev, float tau,
                                    const float* codebook, int D, float* stats, int64_t* index_out, float* out,
                                    float* logits_out, act_stream_t stream);
/* Stage-I tokenizer (models/dvae.py:600, 470-476).  Soft gumbel-softmax over rows: y = softmax((logits + G)/tau), G = noise
 * (parity tests) or Philox keyed by (seed,row,c/4) when noise == NULL; C % 4 == 0, C <= 16384.  backward: dlogits = y (dy - <y,dy>)/tau. */
int act_gumbel_softmax_fwd_f32(const float* logits, int R, int C, const float* noise, uint64_t seed, float tau, float* y,
                               act_stream_t stream);
int act_gumbel_softmax_bwd_f32(const float* y, const float* dy, int R, int C, float tau, float* dlogits, act_stream_t stream);
/* klv = KL(mean_g softmax(logits[b,g,:]) || uniform), reduction 'batchmean'; lse [B*G] and qbar [B,C] are kept for backward. */
int act_kl_uniform_fwd_f32(const float* logits, int B, int G, int C, float* lse, float* qbar, float* klv_out, act_stream_t stream);
int act_kl_uniform_bwd_f32(const float* logits, const float* lse, const float* qbar, const float* grad_klv, int B, int G, int C,
                           float* dlogits, act_stream_t stream);

/* ---- OPT-IN: NT GEMM on the bf16 matrix cores with split operands ("bf16x3"), frozen teacher only (csrc/gemm_bf16x3.hip) -------------------------
 * x ~ hi + lo with hi = bf16(x), lo = bf16(x - hi) (round to nearest even): C = epilogue(A_hi.B_hi^T + A_hi.B_lo^T + A_lo.B_hi^T), fp32 accumulation.
 * 16 significand bits per operand: 4e-6 relative per product; NOT used by default (ACT_TEACHER_BF16X3=1 routes the teacher's ViT GEMMs here).
 * act_split_bf16x2_f32: fp32 [R, K] (row stride ldx, K % 4 == 0, 16-byte aligned) -> planes hi / lo [R][K] (uint16 bf16 bit patterns).
 * act_sgemm_nt_bf16x3_f32: planes of A [M][K] and B [N][K] (contiguous rows) -> C [M, ldc] fp32; M, N % 128 == 0, K % 64 == 0
 * (act_sgemm_nt_bf16x3_supported); epilogue: alpha, bias, ACT_EPI_NONE | ACT_EPI_GELU (+ aux), rowscale, res -- no accumulate. */
int act_split_bf16x2_f32(const float* x, int R, int K, int ldx, uint16_t* hi, uint16_t* lo, act_stream_t stream);
int act_sgemm_nt_bf16x3_supported(int M, int N, int K);
int act_sgemm_nt_bf16x3_f32(int M, int N, int K, const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* b_hi, const uint16_t* b_lo,
                            float* C, int ldc, const act_gemm_epilogue_t* epilogue, act_stream_t stream);
/* producers that hand their result on as planes (y nullable: planes only): LayerNorm of x + pos, and the prompt rows' dropout + position + LayerNorm;
 * bit-identical to producing fp32 and splitting it with act_split_bf16x2_f32 */
int act_layernorm_fwd_planes_f32(const float* x, const float* pos, const float* gamma, const float* beta, float* xin_out, float* y,
                                 uint16_t* y_hi, uint16_t* y_lo, float* mean /* nullable */, float* rstd /* nullable */, int T, int D, float eps,
                                 act_stream_t stream);
int act_prompt_layernorm_fwd_planes_f32(const float* tok, const float* ppos, int B, int P, int D, float drop_p, uint64_t seed,
                                        const uint64_t* seed_dev, const float* gamma, const float* beta, float eps, uint16_t* y_hi, uint16_t* y_lo,
                                        act_stream_t stream);
int act_attention_fwd_prefix_planes_f32(const float* kv0, int S0, const float* qkv1, int Sq, float* out, uint16_t* out_hi, uint16_t* out_lo, float* lse,
                                        int B, int H, int head_dim, float scale, act_stream_t stream);
/* the same product with the result ALSO (C != NULL) or ONLY (C == NULL) written as (hi, lo) bf16 planes [M][N]: the A operand of the next split-bf16
 * product comes straight out of this epilogue (teacher MLP: fc1 + GELU -> planes -> fc2) */
int act_sgemm_nt_bf16x3_planes_f32(int M, int N, int K, const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* b_hi, const uint16_t* b_lo,
                                   float* C, int ldc, uint16_t* out_hi, uint16_t* out_lo, const act_gemm_epilogue_t* epilogue, act_stream_t stream);

/* ---- GEMM launch-configuration table (host side) ------------------------------------------------------------------------
 * act_sgemm_f32 (no explicit configuration) first consults this table keyed by (a_kmajor, b_kmajor, M, N, K), then its built-in
 * cost model.  The Python host fills it from the shipped tune file and from first-use timing (act_amd/kernels.py); the composite
 * entry points below therefore launch exactly the configurations the single-GEMM path uses. */
int act_gemm_tune_set(int a_kmajor, int b_kmajor, int M, int N, int K, int tile, int splits);
int act_gemm_tune_get(int a_kmajor, int b_kmajor, int M, int N, int K, int* tile, int* splits);   /* 0 = found, 1 = absent */
int act_gemm_tune_clear(void);

/* x[r,:] * gate[r / rows_per_scale] -> y   (DropPath gate applied to a gradient, utils/transformer_layers.py:105-120) */
int act_scale_rows_f32(const float* x, const float* gate, int T, int D, int rows_per_scale, float* y, act_stream_t stream);
/* eval-mode BatchNorm folded to an affine map: scale = gamma * rsqrt(running_var + eps), shift = beta - running_mean * scale */
int act_bn_eval_affine_f32(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                           int C, float* scale, float* shift, act_stream_t stream);

/* ---- composite entry points: one host call enqueues every kernel of a module -------------------------------------------
 * A Stage-II step is ~500 kernel launches; issued one ctypes call at a time the host needs ~16 ms per step, which is what
 * bounds an 8-rank node sharing one host.  These functions run the launch sequence of a whole module in C: same kernels, same
 * order, same results (bit-identical to calling the single-kernel entry points above one by one).  All buffers are caller-owned;
 * `saved` / `scratch` are single slabs whose sizes the *_floats helpers return; nothing is allocated, nothing synchronises.
 * `side_stream` (nullable): weight-gradient GEMMs and bias column sums are enqueued there (forked after the producing kernel on
 * `stream`, joined back before the function returns) so they share the chip with the latency-bound dX chain. */
/* GEMM-shape collection (host-side autotuning of composites): between begin and end, on the same host thread, the composite
 * entry points launch nothing and record the (a_kmajor, b_kmajor, M, N, K) of each GEMM they would launch.
 * end -> number recorded (the first `max` written to shapes [max][5]). */
int act_composite_collect_begin(void);
int act_composite_collect_end(int* shapes, int max);
/* The only state the library keeps: up to 32 fork / join events (hipEventDisableTiming) per stream that ever produced work for another stream
 * inside a composite call.  act_composite_shutdown destroys them; call it when no composite call is in flight on any host thread and the
 * streams involved are idle (e.g. before hipDeviceReset / at interpreter exit).  Later composite calls simply create new events.  Returns
 * the number of events destroyed. */
int act_composite_shutdown(void);

typedef struct {                     /* parameters of one pre-LN Transformer block (models/act.py:72-90; timm ViT block) */
    const float *norm1_w, *norm1_b, *qkv_w, *qkv_b /* nullable */, *proj_w, *proj_b, *norm2_w, *norm2_b,
                *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} act_block_params_t;
typedef struct {                     /* where the gradients go; NULL struct pointer = frozen block (dX only) */
    float *norm1_w, *norm1_b, *qkv_w, *qkv_b /* nullable */, *proj_w, *proj_b, *norm2_w, *norm2_b, *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} act_block_grads_t;
typedef struct { int B, S, D, heads, hidden; float eps; } act_block_dims_t;

/* forward of blk(x + pos) (models/act.py:87-90 called as :109-112): 7 launches.  x, pos (nullable), out: [B*S, D];
 * gate1 / gate2 (nullable) [B] = DropPath floor(keep+U)/keep of the two residual branches; keep_for_backward = 0 skips the
 * statistics / pre-activation stores.  saved: act_block_saved_floats(dims) floats (activations kept for the backward). */
size_t act_block_saved_floats(const act_block_dims_t* d);
int act_block_fwd_f32(const act_block_dims_t* d, const act_block_params_t* w, const float* x, const float* pos,
                      const float* gate1, const float* gate2, int keep_for_backward, float* saved, float* out,
                      float* workspace, size_t workspace_bytes, act_stream_t stream);
/* backward: dout [B*S, D] -> dx [B*S, D] (gradient of x and of pos) + parameter gradients.  scratch: act_block_bwd_scratch_floats. */
size_t act_block_bwd_scratch_floats(const act_block_dims_t* d);
int act_block_bwd_f32(const act_block_dims_t* d, const act_block_params_t* w, const float* gate1, const float* gate2,
                      const float* saved, const float* dout, float* dx, const act_block_grads_t* grads, float* scratch,
                      float* workspace, size_t workspace_bytes, float* side_workspace, size_t side_workspace_bytes,
                      act_stream_t stream, act_stream_t side_stream);

/* A STACK of `depth` such blocks in one host call -- x = blk_l(x + pos) for l = 0 .. depth-1, the loop of TransformerEncoder.forward /
 * TransformerDecoder.forward (models/act.py:109-112,140-143) -- and its backward.  Pure launch sequencing: the forward is depth calls of
 * act_block_fwd_f32, the backward depth calls of act_block_bwd_f32 in reverse order plus, when dpos != NULL, depth-1 element-wise additions that
 * accumulate the gradient of the shared `pos` input as ((dx_{depth-1} + dx_{depth-2}) + ...) + dx_0, the order in which an autograd engine
 * folds the per-block gradients: results are bit-identical to the per-block calls (tests/test_gpu_composite.py).
 * gate1 / gate2: arrays of depth pointers (entries nullable; a NULL array = no DropPath anywhere).
 * saved: act_block_stack_saved_floats(d, depth, keep_for_backward) floats; scratch: act_block_stack_bwd_scratch_floats(d, depth).
 * grads: array of depth structs (NULL = frozen stack, dX only).  dx = gradient of x; dpos (nullable) = gradient of pos, a separate buffer
 * when depth > 1 (with depth == 1 it is not written: the gradient of pos IS dx). */
typedef struct {
    int depth;
    const act_block_params_t* blocks;          /* [depth] */
    const float* const* gate1;                 /* [depth] or NULL */
    const float* const* gate2;                 /* [depth] or NULL */
} act_block_stack_t;
size_t act_block_stack_saved_floats(const act_block_dims_t* d, int depth, int keep_for_backward);
size_t act_block_stack_bwd_scratch_floats(const act_block_dims_t* d, int depth);
int act_block_stack_fwd_f32(const act_block_dims_t* d, const act_block_stack_t* st, const float* x, const float* pos, int keep_for_backward,
                            float* saved, float* out, float* workspace, size_t workspace_bytes, act_stream_t stream);
/* dpos (nullable): gradient of the shared pos, folded in the order the blocks finish; with depth 1 and no dpos_in nothing is written (it IS dx).
   dpos_in (nullable, needs dpos): the fold of the deeper chunks of the same stack -- the chain continues through it, so a stack differentiated in
   chunks associates exactly like the unchunked call */
int act_block_stack_bwd_f32(const act_block_dims_t* d, const act_block_stack_t* st, const float* saved, const float* dout, float* dx, float* dpos,
                            const float* dpos_in, const act_block_grads_t* grads, float* scratch, float* workspace, size_t workspace_bytes, float* side_workspace,
                            size_t side_workspace_bytes, act_stream_t stream, act_stream_t side_stream);
/* out[i] = a[i] + b[i] (one rounding; out may alias a or b), n % 4 == 0, 16-byte aligned */
int act_add_f32(const float* a, const float* b, float* out, long long n, act_stream_t stream);

/* Block on G patch tokens per cloud with P prompt tokens acting as keys / values only (prompt-tuned frozen Transformer,
 * models/dvae.py:536-576: every layer replaces the prompt rows of its input and the output drops them).  dims: S = G.
 * x, pos [B*G, D]; prompt rows either prm [B*P, D] (= dropout(prompt) + prompt_pos, differentiable path) or n1p [B*P, D]
 * (their LayerNorm, already computed by act_prompt_layernorm_fwd_f32).  Weights are frozen: the backward produces dx (= dpos) and
 * dprm only.  saved: act_prefix_block_saved_floats; scratch: act_prefix_block_bwd_scratch_floats. */
size_t act_prefix_block_saved_floats(const act_block_dims_t* d, int P);
int act_prefix_block_fwd_f32(const act_block_dims_t* d, int P, const act_block_params_t* w, const float* x, const float* pos,
                             const float* prm, const float* n1p, int keep_for_backward, float* saved, float* out,
                             float* workspace, size_t workspace_bytes, act_stream_t stream);
size_t act_prefix_block_bwd_scratch_floats(const act_block_dims_t* d, int P);
int act_prefix_block_bwd_f32(const act_block_dims_t* d, int P, const act_block_params_t* w, const float* prm, const float* saved,
                             const float* dout, float* dx, float* dprm, float* scratch, float* workspace, size_t workspace_bytes,
                             act_stream_t stream);

/* Whole frozen prompt-tuned Transformer of the Stage-II teacher in ONE call (visual_embedding_deep_prompt, models/dvae.py:536-576,
 * inference form): pos = visual_pos_embed(center); x = proj_pre(tokens); depth x { LN(dropout(prompt_i) + prompt_pos_i) (in-kernel
 * Philox keyed by seed_base + 7919 (i+1) and the device-resident step counter), prefix block }; LN; proj_post.  ~125 launches. */
typedef struct {
    int B, P, G, D, heads, hidden, depth, tokens_dims, pos_hidden;
    float eps, drop_p;
    uint64_t seed_base;
    const uint64_t* seed_dev;
    const float *pos_w0, *pos_b0, *pos_w1, *pos_b1, *pre_w, *pre_b, *post_w, *post_b, *norm_w, *norm_b;
    const float* const* prompt_tok;           /* [depth] -> [P, D] */
    const float* const* prompt_pos;           /* [depth] -> [P, D] */
    const act_block_params_t* blocks;         /* [depth] */
} act_prefix_vit_t;
size_t act_prefix_vit_scratch_floats(const act_prefix_vit_t* m);
/* OPT-IN variant (never the default; ACT_TEACHER_BF16X3=1 on the host side): the five Linear products of every block on the split-bf16 kernel
 * (act_sgemm_nt_bf16x3_f32) wherever it takes the shape, everything else as act_prefix_vit_fwd_f32.  w_planes[4 i + {0,1,2,3}] = hi plane of block i's
 * qkv_w [3D][D], proj_w [D][D], fc1_w [hidden][D], fc2_w [D][hidden] (the lo plane follows the hi plane: rows * cols elements further);
 * a_planes: scratch for activation planes, a_planes_elems >= 2 * B*G * hidden + 2 * max(B*G, B*P) * D (the MLP's hidden activation leaves fc1's epilogue
 * as planes and never exists in fp32; the other activations are split by a pass).  Teacher features move by ~7e-6 of their range. */
typedef struct {
    const uint16_t* const* w_planes;          /* [4 * depth] */
    uint16_t* a_planes;
    size_t a_planes_elems;
} act_vit_bf16x3_t;
int act_prefix_vit_fwd_bf16x3_f32(const act_prefix_vit_t* m, const act_vit_bf16x3_t* x3, const float* tokens, const float* center, float* out,
                                  float* scratch, float* workspace, size_t workspace_bytes, act_stream_t stream);
/* ... and the differentiable forward of ONE prefix block (Stage-I prompt tuning) the same way: x3->w_planes[0..3] = this block's four weights; the backward
 * (act_prefix_block_bwd_f32) is unchanged, everything it reads is still written. */
int act_prefix_block_fwd_bf16x3_f32(const act_block_dims_t* d, int P, const act_block_params_t* w, const act_vit_bf16x3_t* x3, const float* x, const float* pos,
                                    const float* prm, int keep_for_backward, float* saved, float* out, float* workspace, size_t workspace_bytes,
                                    act_stream_t stream);
/* ... and its backward: x3->w_planes[0..4] = planes of the TRANSPOSED weights fc2_w^T [hidden][D], fc1_w^T [D][hidden], proj_w^T [D][D], qkv_w^T [D][3D],
 * (qkv_w rows D..3D)^T [D][2D]; the five input-gradient products run on the split-bf16 kernel, LayerNorm / attention backward stay f32. */
int act_prefix_block_bwd_bf16x3_f32(const act_block_dims_t* d, int P, const act_block_params_t* w, const act_vit_bf16x3_t* x3, const float* prm,
                                    const float* saved, const float* dout, float* dx, float* dprm, float* scratch, float* workspace,
                                    size_t workspace_bytes, act_stream_t stream);
int act_prefix_vit_fwd_f32(const act_prefix_vit_t* m, const float* tokens, const float* center, float* out, float* scratch,
                           float* workspace, size_t workspace_bytes, act_stream_t stream);

/* mini-PointNet patch embedding (Encoder, models/dvae.py:185-215) on rows = points: x [BG*n, 3] -> tokens [BG, C].
 * conv 3->128, BN, ReLU, conv 128->256, max over the n points of a group, conv 512->512 on cat(global, local) (the global half
 * evaluated once per group), BN, ReLU, conv 512->C, max.  training: batch statistics + running-stat update, else running stats. */
typedef struct {
    const float *c1_w, *c1_b, *bn1_w, *bn1_b, *c2_w, *c2_b, *c3_w, *c3_b, *bn2_w, *bn2_b, *c4_w, *c4_b;
    float *bn1_mean, *bn1_var, *bn2_mean, *bn2_var;      /* running statistics (updated in place when training) */
} act_pointnet_params_t;
typedef struct { float *c1_w, *c1_b, *bn1_w, *bn1_b, *c2_w, *c2_b, *c3_w, *c3_b, *bn2_w, *bn2_b, *c4_w, *c4_b; } act_pointnet_grads_t;
typedef struct { int BG, n, C; float eps1, eps2, momentum1, momentum2; } act_pointnet_dims_t;
size_t act_pointnet_saved_floats(const act_pointnet_dims_t* d);
int act_pointnet_fwd_f32(const act_pointnet_dims_t* d, const act_pointnet_params_t* w, const float* x, int training,
                         int keep_for_backward, float* saved, float* out, float* workspace, size_t workspace_bytes,
                         act_stream_t stream);
/* The same forward when only some groups' tokens are wanted (Stage II: MaskTransformer keeps the visible patches, models/act.py:269-275): everything in
 * front of the last conv still runs on all groups (its BatchNorm statistics are over all of them), the last conv + max-pool only on the n_groups listed
 * groups (int32 ids, device memory; n_groups * n a multiple of 128 -- repeat an id to pad).  out / the saved arg-max rows of unlisted groups are set to
 * zero; the backward then expects a zero gradient row for them.  groups == NULL: act_pointnet_fwd_f32. */
int act_pointnet_fwd_groups_f32(const act_pointnet_dims_t* d, const act_pointnet_params_t* w, const float* x, int training, int keep_for_backward,
                                float* saved, float* out, const int32_t* groups, int n_groups, float* workspace, size_t workspace_bytes,
                                act_stream_t stream);
size_t act_pointnet_bwd_scratch_floats(const act_pointnet_dims_t* d);
int act_pointnet_bwd_f32(const act_pointnet_dims_t* d, const act_pointnet_params_t* w, const float* x, const float* saved,
                         const float* dout, const act_pointnet_grads_t* grads, float* scratch, float* workspace,
                         size_t workspace_bytes, act_stream_t stream);

/* DGCNN (models/dvae.py:26-117), inference form, everything up to (not including) layer5's GroupNorm: f [B*G, Cin], graph idx
 * int64 [B,k,G] -> h [B*G, Cout].  w_in/b_in = input_trans; stacked[l] = [Wa ; Wb - Wa] of edge-conv layer l ([2*cout_l, cin_l]);
 * gn_w/gn_b[l] = its GroupNorm(4) affine; w5 = layer5 conv [Cout, 2304]. */
typedef struct {
    int B, G, k, Cin, Cout, groups;
    float eps, slope;
    const float *w_in, *b_in, *w5;
    const float* stacked[4];
    const float* gn_w[4];
    const float* gn_b[4];
} act_dgcnn_t;
size_t act_dgcnn_scratch_floats(const act_dgcnn_t* m);
int act_dgcnn_features_f32(const act_dgcnn_t* m, const float* f, const int64_t* idx, float* h, float* scratch, float* workspace,
                           size_t workspace_bytes, act_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ACT_HIP_H */
