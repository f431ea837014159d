"""Oracle (TEST INFRASTRUCTURE): numpy restatement of the point operators.

Conventions fixed here and followed bit-for-bit by the
HIP kernels in act_amd/csrc/point_ops.hip:

  * squared distance  d = (dx*dx + dy*dy) + dz*dz  evaluated in fp32 with one
    rounding per operation and NO fused multiply-add;
  * FPS: first index 0, running min-distance initialised to 1e10, arg-max
    with lowest-index tie-break   (reference: utils/misc.py:39-46 ->
    pointnet2_ops.furthest_point_sample; in-tree restatement
    part_segmentation/models/pointnet2_utils.py:60-81);
  * kNN: K smallest by (distance, index) ascending, int64 indices
    (reference call sites models/dvae.py:159,172 and :23,68; KNN_CUDA 0.2);
  * Chamfer: nearest neighbour with strict '<' scan => lowest index wins
    (extensions/chamfer_dist/chamfer.cu:33-140), backward :185-198.
"""
import numpy as np

F32 = np.float32


def _sqdist(a, b):
    """a [...,3], b [...,3] broadcastable -> fp32 (dx*dx + dy*dy) + dz*dz."""
    d = (a - b).astype(F32)
    dx, dy, dz = d[..., 0], d[..., 1], d[..., 2]
    return ((dx * dx).astype(F32) + (dy * dy).astype(F32)).astype(F32) + (dz * dz).astype(F32)


def fps_ref(xyz, npoint, skip_near_origin=False):
    """xyz f32 [B,N,3] -> int32 [B,npoint] (utils/misc.py:44 semantics).

    skip_near_origin reproduces the upstream pointnet2_ops quirk (points with
    |p|^2 <= 1e-3 never update / never win); off by default because the
    reference's in-tree CPU restatement has no such skip.
    """
    xyz = np.ascontiguousarray(xyz, dtype=F32)
    B, N, _ = xyz.shape
    idx = np.zeros((B, npoint), dtype=np.int32)
    for b in range(B):
        p = xyz[b]
        temp = np.full((N,), 1e10, dtype=F32)
        if skip_near_origin:
            mag = ((p[:, 0] * p[:, 0]).astype(F32) + (p[:, 1] * p[:, 1]).astype(F32)).astype(F32) \
                + (p[:, 2] * p[:, 2]).astype(F32)
            live = mag > F32(1e-3)
        old = 0
        for j in range(1, npoint):
            d = _sqdist(p, p[old][None, :])
            if skip_near_origin:
                temp = np.where(live, np.minimum(temp, d), temp)
                cand = np.where(live, temp, F32(-1.0))
                old = int(np.argmax(cand)) if live.any() else 0
            else:
                temp = np.minimum(temp, d)
                old = int(np.argmax(temp))      # first occurrence == lowest index
            idx[b, j] = old
    return idx


def gather_ref(xyz, idx):
    """xyz [B,N,C], idx [B,S] -> [B,S,C]  (pointnet2_ops.gather_operation, utils/misc.py:45)."""
    B = xyz.shape[0]
    return xyz[np.arange(B)[:, None], idx.astype(np.int64)]


def knn_ref(ref, query, k):
    """ref f32 [B,N,3], query f32 [B,Q,3] -> (dist f32 [B,Q,k] (sqrt), idx int64 [B,Q,k]).

    KNN(k, transpose_mode=True).forward(ref, query)  (models/dvae.py:172).
    """
    ref = np.ascontiguousarray(ref, dtype=F32)
    query = np.ascontiguousarray(query, dtype=F32)
    d = _sqdist(query[:, :, None, :], ref[:, None, :, :])          # [B,Q,N]
    idx = np.argsort(d, axis=-1, kind="stable")[..., :k].astype(np.int64)
    dk = np.take_along_axis(d, idx, axis=-1)
    return np.sqrt(dk).astype(F32), idx


def group_ref(xyz, num_group, group_size):
    """Group.forward (models/dvae.py:161-183): -> neighborhood [B,G,M,3], center [B,G,3],
    plus the intermediate indices (fps idx int32 [B,G], knn idx int64 [B,G,M])."""
    xyz = np.ascontiguousarray(xyz, dtype=F32)
    fidx = fps_ref(xyz, num_group)
    center = gather_ref(xyz, fidx)
    _, kidx = knn_ref(xyz, center, group_size)
    B = xyz.shape[0]
    nb = xyz[np.arange(B)[:, None, None], kidx]                     # [B,G,M,3]
    nb = (nb - center[:, :, None, :]).astype(F32)
    return nb, center, fidx, kidx


def _chamfer_fwd_fma_c(xyz1, xyz2):
    """the FMA-contracted distance needs a correctly rounded fused multiply-add, which numpy does not have: plain-C fmaf() of the C oracle"""
    import ctypes
    import os
    import subprocess
    d = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(d, "liboracle_point_ops.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(d, "point_ops_c.c")):
        subprocess.check_call(["make", "-C", d, "-s"])
    lib = ctypes.CDLL(so)
    B, n, _ = xyz1.shape
    m = xyz2.shape[1]
    d1 = np.empty((B, n), F32); d2 = np.empty((B, m), F32); i1 = np.empty((B, n), np.int32); i2 = np.empty((B, m), np.int32)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.oracle_chamfer_fwd_fma_f32(P(xyz1), P(xyz2), B, n, m, P(d1), P(d2), P(i1), P(i2))
    return d1, d2, i1, i2


def chamfer_fwd_ref(xyz1, xyz2, fma_contract=False):
    """chamfer.forward (chamfer_cuda.cpp:12-23 / chamfer.cu:15-170):
    -> dist1 [B,n], dist2 [B,m] (squared), idx1 int32 [B,n], idx2 int32 [B,m].
    fma_contract: the distance as an FMA-contracting build of chamfer.cu:43-57 rounds it, fma(z2, z2, fma(x2, x2, y2*y2))."""
    xyz1 = np.ascontiguousarray(xyz1, dtype=F32)
    xyz2 = np.ascontiguousarray(xyz2, dtype=F32)
    if fma_contract:
        return _chamfer_fwd_fma_c(xyz1, xyz2)
    d = _sqdist(xyz1[:, :, None, :], xyz2[:, None, :, :])           # [B,n,m]
    idx1 = np.argmin(d, axis=2).astype(np.int32)                     # first min == strict '<' scan
    idx2 = np.argmin(d, axis=1).astype(np.int32)
    dist1 = np.take_along_axis(d, idx1[:, :, None].astype(np.int64), axis=2)[:, :, 0]
    dist2 = np.take_along_axis(d, idx2[:, None, :].astype(np.int64), axis=1)[:, 0, :]
    return dist1, dist2, idx1, idx2


def chamfer_bwd_ref(xyz1, xyz2, idx1, idx2, g1, g2, dtype=np.float64):
    """chamfer.backward (chamfer.cu:173-229).  Accumulated in float64 here so the
    result is the order-independent exact sum the reference's atomicAdd scatter
    approximates (its fp32 summation order is nondeterministic)."""
    x1 = xyz1.astype(dtype); x2 = xyz2.astype(dtype)
    B, n, _ = x1.shape; m = x2.shape[1]
    gx1 = np.zeros_like(x1); gx2 = np.zeros_like(x2)
    for b in range(B):
        j2 = idx1[b].astype(np.int64)
        t = (2.0 * g1[b].astype(dtype))[:, None] * (x1[b] - x2[b][j2])
        gx1[b] += t
        np.add.at(gx2[b], j2, -t)
        j1 = idx2[b].astype(np.int64)
        t = (2.0 * g2[b].astype(dtype))[:, None] * (x2[b] - x1[b][j1])
        gx2[b] += t
        np.add.at(gx1[b], j1, -t)
    return gx1, gx2


def chamfer_l2_ref(xyz1, xyz2):
    """ChamferDistanceL2.forward (extensions/chamfer_dist/__init__.py:28-44)."""
    d1, d2, _, _ = chamfer_fwd_ref(xyz1, xyz2)
    return F32(d1.mean(dtype=np.float64) + d2.mean(dtype=np.float64))


def chamfer_l1_ref(xyz1, xyz2):
    """ChamferDistanceL1.forward (extensions/chamfer_dist/__init__.py:64-84)."""
    d1, d2, _, _ = chamfer_fwd_ref(xyz1, xyz2)
    return F32((np.sqrt(d1).mean(dtype=np.float64) + np.sqrt(d2).mean(dtype=np.float64)) / 2)


def scale_translate_ref(pc, scale, shift):
    """PointcloudScaleAndTranslate with injected draws (datasets/data_transforms.py:26-34):
    pc [B,N,3] * scale[B,3] + shift[B,3] in fp32 (mul then add, no FMA)."""
    pc = np.asarray(pc, dtype=F32)
    return ((pc * np.asarray(scale, F32)[:, None, :]).astype(F32) + np.asarray(shift, F32)[:, None, :]).astype(F32)


def pc_norm_ref(pc):
    """ShapeNet55Dataset.pc_norm (datasets/ShapeNet55Dataset.py:45-51) per cloud, batched."""
    pc = np.asarray(pc, dtype=F32)
    pc = pc - pc.mean(axis=1, keepdims=True)
    m = np.sqrt((pc ** 2).sum(axis=2)).max(axis=1)
    return (pc / m[:, None, None]).astype(F32)
