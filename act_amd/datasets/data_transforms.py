"""Train-time augmentation on the device (reference: datasets/data_transforms.py:20-34).

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
