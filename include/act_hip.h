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
int act_fps_f32(const float* xyz, int B, int N, int G, int32_t* idx_out, float* centers_out,
                int skip_near_origin, act_stream_t stream);

/* knn_cuda.KNN(k).forward fused with Group's gather + centre subtraction
 * (models/dvae.py:159,172-182; DGCNN graph k=4 models/dvae.py:23,68).
 * ref [B,N,3], query [B,Q,3], K <= 64 -> idx int64 ([B,Q,K], or [B,K,Q] when idx_kq != 0:
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

/* ---- Chamfer distance (extensions/chamfer_dist: chamfer_cuda.cpp:12-39, chamfer.cu:15-229) --- */
/* forward: xyz1 [B,n,3], xyz2 [B,m,3] -> dist1 [B,n], dist2 [B,m] (squared), idx1 int32 [B,n], idx2 int32 [B,m] */
int act_chamfer_fwd_f32(const float* xyz1, const float* xyz2, int B, int n, int m,
                        float* dist1, float* dist2, int32_t* idx1, int32_t* idx2, act_stream_t stream);
/* backward: deterministic gather formulation of the reference's atomicAdd scatter */
int act_chamfer_bwd_f32(const float* xyz1, const float* xyz2, const int32_t* idx1, const int32_t* idx2,
                        const float* grad_dist1, const float* grad_dist2, int B, int n, int m,
                        float* grad_xyz1, float* grad_xyz2, act_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ACT_HIP_H */
