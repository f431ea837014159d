"""Train-time augmentation on the device (reference: datasets/data_transforms.py:6-34).

PointcloudScaleAndTranslate: per sample, per axis scale ~ U[2/3, 3/2] and shift ~ U[-0.2, 0.2], applied in place
by one HIP launch (act_scale_translate_f32) with draws sampled on the device: no Python loop over the batch and
no per-sample host->device copies (the reference does 2 numpy draws + 2 tiny H2D copies per sample)."""
import torch

from .. import _C


class PointcloudScaleAndTranslate(object):
    def __init__(self, scale_low=2. / 3., scale_high=3. / 2., translate_range=0.2):
        self.scale_low = scale_low
        self.scale_high = scale_high
        self.translate_range = translate_range

    def __call__(self, pc, scale=None, shift=None):
        """pc f32 [B,N,3] CUDA, modified in place and returned.  ``scale``/``shift`` [B,3] inject the draws."""
        B, N, C = pc.shape
        if C != 3 or not pc.is_cuda:
            raise RuntimeError("PointcloudScaleAndTranslate expects a CUDA tensor [B, N, 3]")
        if scale is None:
            scale = torch.empty(B, 3, device=pc.device).uniform_(self.scale_low, self.scale_high)
        if shift is None:
            shift = torch.empty(B, 3, device=pc.device).uniform_(-self.translate_range, self.translate_range)
        scale = scale.to(pc.device, torch.float32).contiguous(); shift = shift.to(pc.device, torch.float32).contiguous()
        _C.check(_C.lib.act_scale_translate_f32(_C.ptr(pc), _C.ptr(scale), _C.ptr(shift), B, N, _C.stream()),
                 "act_scale_translate_f32")
        return pc


class PointcloudRotate(object):
    """Random rotation about the y axis per sample (datasets/data_transforms.py:6-18): pc[i] @ [[c,0,s],[0,1,0],[-s,0,c]],
    angle = 2*pi*U[0,1).  One HIP launch (act_rotate_points_f32); the angles are drawn on the device."""

    def __call__(self, pc, u=None):
        """pc f32 [B,N,3] CUDA, modified in place and returned.  ``u`` [B] in [0,1) injects the draws."""
        B, N, C = pc.shape
        assert C == 3 and pc.is_contiguous() and pc.dtype == torch.float32
        if u is None:
            u = torch.rand(B, device=pc.device, dtype=torch.float64)
        ang = torch.as_tensor(u, dtype=torch.float64, device=pc.device) * (2 * torch.pi)
        c, s = torch.cos(ang), torch.sin(ang)
        z, o = torch.zeros_like(c), torch.ones_like(c)
        rot = torch.stack((c, z, s, z, o, z, -s, z, c), dim=1).to(torch.float32).contiguous()      # [B,9]
        _C.check(_C.lib.act_rotate_points_f32(_C.ptr(pc), _C.ptr(rot), B, N, _C.stream()), "act_rotate_points_f32")
        return pc
