"""Drop-in for the two pointnet2_ops entry points ACT uses (utils/misc.py:44-45):
``furthest_point_sample`` and ``gather_operation``, backed by act_fps_f32 /
act_gather_points_f32 (include/act_hip.h)."""
import torch

from .. import _C


def _need(t, dtype, name):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")            # upstream: TORCH_CHECK is_cuda
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be a {dtype} tensor")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be a contiguous tensor")


def _scratch(xyz, B, N):
    n = _C.lib.act_fps_scratch_floats(B, N)              # > 0 only for clouds beyond 16,384 points (running distances in HBM)
    return torch.empty(n, dtype=torch.float32, device=xyz.device) if n else None


def _skip(flag):
    """``skip_near_origin=None`` -> the process-wide default: ACT_FPS_SKIP_NEAR_ORIGIN=1 reproduces upstream pointnet2_ops, which
    never selects points with |p|^2 <= 1e-3; the default (0) is the in-tree pure-torch FPS of the reference."""
    return SKIP_NEAR_ORIGIN if flag is None else bool(flag)


SKIP_NEAR_ORIGIN = __import__("os").environ.get("ACT_FPS_SKIP_NEAR_ORIGIN", "0") == "1"


def furthest_point_sample(xyz, npoint, skip_near_origin=None):
    """xyz f32 [B,N,3] -> int32 [B,npoint]; first index 0, lowest-index tie-break.  Non-differentiable.

    ``skip_near_origin=True`` reproduces upstream pointnet2_ops' |p|^2 <= 1e-3 skip."""
    _need(xyz, torch.float32, "xyz")
    B, N, C = xyz.shape
    if C != 3:
        raise RuntimeError("xyz must be [B, N, 3]")
    idx = torch.empty(B, npoint, dtype=torch.int32, device=xyz.device)
    _C.check(_C.lib.act_fps_f32(_C.ptr(xyz), B, N, int(npoint), _C.ptr(idx), None, int(_skip(skip_near_origin)), _C.ptr(_scratch(xyz, B, N)),
                                _C.stream()), "act_fps_f32")
    return idx


def furthest_point_sample_with_centers(xyz, npoint, skip_near_origin=None):
    """fused FPS + gather: -> (idx int32 [B,G], centers f32 [B,G,3]) in one launch."""
    _need(xyz, torch.float32, "xyz")
    B, N, _ = xyz.shape
    idx = torch.empty(B, npoint, dtype=torch.int32, device=xyz.device)
    centers = torch.empty(B, npoint, 3, dtype=torch.float32, device=xyz.device)
    _C.check(_C.lib.act_fps_f32(_C.ptr(xyz), B, N, int(npoint), _C.ptr(idx), _C.ptr(centers), int(_skip(skip_near_origin)),
                                _C.ptr(_scratch(xyz, B, N)), _C.stream()), "act_fps_f32")
    return idx, centers


class GatherOperation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, idx):
        _need(features, torch.float32, "features"); _need(idx, torch.int32, "idx")
        B, C, N = features.shape
        S = idx.shape[1]
        out = torch.empty(B, C, S, dtype=torch.float32, device=features.device)
        _C.check(_C.lib.act_gather_points_f32(_C.ptr(features), _C.ptr(idx), B, C, N, S, _C.ptr(out), _C.stream()),
                 "act_gather_points_f32")
        ctx.save_for_backward(idx)
        ctx.N = N
        ctx.mark_non_differentiable(idx)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (idx,) = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        B, C, S = grad_out.shape
        gf = torch.empty(B, C, ctx.N, dtype=torch.float32, device=grad_out.device)
        _C.check(_C.lib.act_gather_points_bwd_f32(_C.ptr(grad_out), _C.ptr(idx), B, C, ctx.N, S, _C.ptr(gf), _C.stream()),
                 "act_gather_points_bwd_f32")
        return gf, None


gather_operation = GatherOperation.apply
