/* Oracle (TEST INFRASTRUCTURE): plain-C restatement of the point operators.
 *
 * Same conventions as oracle/point_ops.py, fast enough for the BASELINE.json
 * full sizes and used as the timed CPU baseline of the FPS+kNN "Group" path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load the shared object built from this file.  Build: oracle/Makefile
 * (gcc -O2 -ffp-contract=off -fopenmp).
 *
 * Reference semantics restated (not copied):
 *   FPS      utils/misc.py:39-46 -> pointnet2_ops.furthest_point_sample
 *            (un-vendored); in-tree restatement
 *            part_segmentation/models/pointnet2_utils.py:60-81
 *   kNN      models/dvae.py:159,172 -> KNN_CUDA 0.2 (un-vendored)
 *   Group    models/dvae.py:161-183
 *   Chamfer  extensions/chamfer_dist/chamfer.cu:15-145
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

static inline float sqdist3(const float *a, const float *b) {
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    float s = xx + yy;
    return s + zz;
}

/* xyz [B,N,3] -> idx int32 [B,G]; tmp may be NULL */
int oracle_fps_f32(const float *xyz, int B, int N, int G, int32_t *idx, int skip_near_origin) {
    if (G <= 0) return 0;
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        const float *p = xyz + (size_t)b * N * 3;
        float *temp = (float *)malloc(sizeof(float) * (size_t)N);
        for (int i = 0; i < N; ++i) temp[i] = 1e10f;
        int old = 0;
        idx[(size_t)b * G] = 0;
        for (int j = 1; j < G; ++j) {
            float best = -1.0f; int besti = 0;
            const float *c = p + (size_t)old * 3;
            for (int k = 0; k < N; ++k) {
                if (skip_near_origin) {
                    float mag = (p[k*3]*p[k*3] + p[k*3+1]*p[k*3+1]) + p[k*3+2]*p[k*3+2];
                    if (mag <= 1e-3f) continue;
                }
                float d = sqdist3(p + (size_t)k * 3, c);
                float d2 = d < temp[k] ? d : temp[k];
                temp[k] = d2;
                if (d2 > best) { best = d2; besti = k; }
            }
            old = besti;
            idx[(size_t)b * G + j] = old;
        }
        free(temp);
    }
    return 0;
}

/* K smallest by (dist, idx) ascending.  ref [B,N,3], query [B,Q,3] ->
 * idx int64 [B,Q,K], dist (sqrt) f32 [B,Q,K] (nullable). */
int oracle_knn_f32(const float *ref, const float *query, int B, int N, int Q, int K,
                   int64_t *idx_out, float *dist_out) {
    if (K > N) return -1;
#pragma omp parallel for schedule(dynamic, 8) collapse(2)
    for (int b = 0; b < B; ++b) {
        for (int q = 0; q < Q; ++q) {
            const float *r = ref + (size_t)b * N * 3;
            const float *c = query + ((size_t)b * Q + q) * 3;
            float bd[256]; int bi[256];       /* K <= 256 */
            int cnt = 0;
            for (int n = 0; n < N; ++n) {
                float d = sqdist3(r + (size_t)n * 3, c);
                if (cnt == K && !(d < bd[K - 1])) continue;   /* ties keep the earlier index */
                int pos = cnt < K ? cnt : K - 1;
                while (pos > 0 && d < bd[pos - 1]) { bd[pos] = bd[pos - 1]; bi[pos] = bi[pos - 1]; --pos; }
                bd[pos] = d; bi[pos] = n;
                if (cnt < K) ++cnt;
            }
            int64_t *o = idx_out + ((size_t)b * Q + q) * K;
            for (int k = 0; k < K; ++k) o[k] = bi[k];
            if (dist_out) {
                float *dd = dist_out + ((size_t)b * Q + q) * K;
                for (int k = 0; k < K; ++k) dd[k] = __builtin_sqrtf(bd[k]);
            }
        }
    }
    return 0;
}

/* Group.forward: xyz [B,N,3] -> center [B,G,3], nbr [B,G,M,3] (centred),
 * fps idx int32 [B,G], knn idx int64 [B,G,M]. */
int oracle_group_f32(const float *xyz, int B, int N, int G, int M,
                     float *center, float *nbr, int32_t *fidx, int64_t *kidx) {
    int rc = oracle_fps_f32(xyz, B, N, G, fidx, 0);
    if (rc) return rc;
    for (int b = 0; b < B; ++b)
        for (int g = 0; g < G; ++g)
            memcpy(center + ((size_t)b * G + g) * 3,
                   xyz + ((size_t)b * N + fidx[(size_t)b * G + g]) * 3, 3 * sizeof(float));
    rc = oracle_knn_f32(xyz, center, B, N, G, M, kidx, 0);
    if (rc) return rc;
#pragma omp parallel for
    for (int b = 0; b < B; ++b)
        for (int g = 0; g < G; ++g) {
            const float *c = center + ((size_t)b * G + g) * 3;
            for (int m = 0; m < M; ++m) {
                const float *p = xyz + ((size_t)b * N + kidx[((size_t)b * G + g) * M + m]) * 3;
                float *o = nbr + (((size_t)b * G + g) * M + m) * 3;
                o[0] = p[0] - c[0]; o[1] = p[1] - c[1]; o[2] = p[2] - c[2];
            }
        }
    return 0;
}

/* the squared distance as an FMA-contracting build (nvcc default) of chamfer.cu:43-57 evaluates x2*x2 + y2*y2 + z2*z2 with x2 = b - a:
 * fma(z2, z2, fma(x2, x2, y2*y2)) -- fmaf() is the correctly rounded fused operation whatever -ffp-contract says */
static inline float sqdist3_fma(const float *b, const float *a) {
    float dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
    float yy = dy * dy;
    return fmaf(dz, dz, fmaf(dx, dx, yy));
}

/* one direction of chamfer.forward: for each point of a [B,n,3] nearest in b [B,m,3] */
static void chamfer_dir_fma(const float *a, int n, const float *bb, int m, int B, float *dist, int32_t *idx) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < B; ++i)
        for (int j = 0; j < n; ++j) {
            const float *p = a + ((size_t)i * n + j) * 3;
            float best = 0.f; int bi = 0;
            for (int k = 0; k < m; ++k) {
                float d = sqdist3_fma(bb + ((size_t)i * m + k) * 3, p);
                if (k == 0 || d < best) { best = d; bi = k; }
            }
            dist[(size_t)i * n + j] = best; idx[(size_t)i * n + j] = bi;
        }
}
int oracle_chamfer_fwd_fma_f32(const float *xyz1, const float *xyz2, int B, int n, int m,
                               float *dist1, float *dist2, int32_t *idx1, int32_t *idx2) {
    chamfer_dir_fma(xyz1, n, xyz2, m, B, dist1, idx1);
    chamfer_dir_fma(xyz2, m, xyz1, n, B, dist2, idx2);
    return 0;
}

static void chamfer_dir(const float *a, int n, const float *bb, int m, int B, float *dist, int32_t *idx) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < B; ++i)
        for (int j = 0; j < n; ++j) {
            const float *p = a + ((size_t)i * n + j) * 3;
            float best = 0.f; int bi = 0;
            for (int k = 0; k < m; ++k) {
                float d = sqdist3(bb + ((size_t)i * m + k) * 3, p);
                if (k == 0 || d < best) { best = d; bi = k; }
            }
            dist[(size_t)i * n + j] = best; idx[(size_t)i * n + j] = bi;
        }
}

int oracle_chamfer_fwd_f32(const float *xyz1, const float *xyz2, int B, int n, int m,
                           float *dist1, float *dist2, int32_t *idx1, int32_t *idx2) {
    chamfer_dir(xyz1, n, xyz2, m, B, dist1, idx1);
    chamfer_dir(xyz2, m, xyz1, n, B, dist2, idx2);
    return 0;
}
