"""``misc.fps`` and small helpers (reference: utils/misc.py:39-46,68-92,277-306)."""
import random

import numpy as np
import torch

from ..pointnet2_ops import pointnet2_utils
from .logger import print_log


def fps(data, number):
    """data [B,N,3] -> the ``number`` farthest-point-sampled points [B,number,3] (utils/misc.py:39-46).
    One fused launch (index selection + gather); differentiable w.r.t. ``data`` like gather_operation."""
    if data.requires_grad:
        fps_idx = pointnet2_utils.furthest_point_sample(data.detach().contiguous(), number)
        return pointnet2_utils.gather_operation(data.transpose(1, 2).contiguous(), fps_idx).transpose(1, 2).contiguous()
    _, centers = pointnet2_utils.furthest_point_sample_with_centers(data.contiguous(), number)
    return centers


def set_random_seed(seed, deterministic=False):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    if deterministic:
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False


def worker_init_fn(worker_id):
    np.random.seed(np.random.get_state()[1][0] + worker_id)


def summary_parameters(model, logger=None):
    print_log("Trainable parameters:", logger=logger)
    for name, param in model.named_parameters():
        if param.requires_grad:
            print_log(f"{name}, {param.size()}", logger=logger)
    print_log("Untrainable parameters:", logger=logger)
    for name, param in model.named_parameters():
        if not param.requires_grad:
            print_log(f"{name}, {param.size()}", logger=logger)
    tot = sum(p.numel() for p in model.parameters())
    trn = sum(p.numel() for p in model.parameters() if p.requires_grad)
    print_log(f"The Model has {tot / 1e6:.2f}M parameters, {trn / 1e6:.2f}M trainable", logger=logger)
