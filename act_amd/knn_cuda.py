"""Drop-in for ``knn_cuda.KNN`` (KNN_CUDA 0.2; call sites models/dvae.py:23,68,159,172),
backed by act_knn_group_f32 (include/act_hip.h).  One launch for the whole batch (the upstream
wrapper loops over clouds in Python)."""
import torch
import torch.nn as nn

from . import _C


def knn_group(ref, query, k, want_nbr=False, want_dist=False, idx_kq=False):
    """ref [B,N,3], query [B,Q,3] -> idx int64 ([B,Q,k] or [B,k,Q]), nbr [B,Q,k,3] = ref[idx]-query, dist (sqrt)."""
    for t, n in ((ref, "ref"), (query, "query")):
        if not t.is_cuda or t.dtype != torch.float32:
            raise RuntimeError(f"{n} must be a float32 CUDA tensor")
    ref = ref.contiguous(); query = query.contiguous()
    B, N, D = ref.shape
    Q = query.shape[1]
    if D != 3 or query.shape[2] != 3:
        raise RuntimeError("KNN: only 3-D points are supported on this path")
    shape = (B, k, Q) if idx_kq else (B, Q, k)
    idx = torch.empty(shape, dtype=torch.int64, device=ref.device)
    nbr = torch.empty(B, Q, k, 3, dtype=torch.float32, device=ref.device) if want_nbr else None
    dist = torch.empty(shape, dtype=torch.float32, device=ref.device) if want_dist else None
    _C.check(_C.lib.act_knn_group_f32(_C.ptr(ref), _C.ptr(query), B, N, Q, int(k), _C.ptr(idx), int(idx_kq),
                                      _C.ptr(nbr), _C.ptr(dist), _C.stream()), "act_knn_group_f32")
    return idx, nbr, dist


class KNN(nn.Module):
    def __init__(self, k, transpose_mode=False):
        super().__init__()
        self.k = k
        self._t = transpose_mode

    @torch.no_grad()
    def forward(self, ref, query):
        """transpose_mode=True : ref [B,N,3], query [B,Q,3] -> dist [B,Q,k], idx int64 [B,Q,k]
           transpose_mode=False: ref [B,3,N], query [B,3,Q] -> dist [B,k,Q], idx int64 [B,k,Q]"""
        assert ref.size(0) == query.size(0), "ref.shape={} != query.shape={}".format(ref.shape, query.shape)
        if not self._t:
            ref, query = ref.transpose(1, 2), query.transpose(1, 2)
        idx, _, dist = knn_group(ref, query, self.k, want_dist=True, idx_kq=not self._t)
        return dist, idx
