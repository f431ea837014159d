#!/usr/bin/env python3
"""Generates tests/golden/g10_finetune.npz by importing the REFERENCE's PointTransformer (models/act.py:727-910) and
PointcloudRotate (datasets/data_transforms.py:6-18) under the same shims as make_golden.py.  Run in the build container
(needs /root/reference); the .npz is the committed fixture, this script is its provenance.

    python tests/golden/make_golden_finetune.py
"""
import os
import sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import install_shims, save, REF                      # noqa: E402
from fill import fill_module, clouds, TINY_FINETUNE, TINY_FT_LABELS    # noqa: E402


def main():
    os.chdir(REF)
    install_shims()
    import models.act as act                                            # noqa: F401
    from models import build_model_from_cfg
    from easydict import EasyDict
    torch.set_num_threads(8)
    pts = torch.from_numpy(clouds(10, 4, 128))
    label = torch.tensor(TINY_FT_LABELS)
    out = {}
    for ttype in ("full", "linear", "side"):
        cfg = EasyDict(dict(TINY_FINETUNE, transfer_type=ttype))
        model = fill_module(build_model_from_cfg(cfg), f"g10.{ttype}.")
        if ttype == "side":
            with torch.no_grad():
                model.side_alpha.fill_(0.3)
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0                                               # dropout parity is covered oracle<->HIP with replayed draws
        model.train()
        logits = model(pts)
        loss, acc = model.get_loss_acc(logits, label)
        out[f"{ttype}_logits_train"] = logits
        out[f"{ttype}_loss"] = loss
        out[f"{ttype}_acc"] = acc
        if ttype != "linear":
            names = [n for n, p in model.named_parameters() if p.requires_grad]
            loss.backward()
            out[f"{ttype}_grad_names"] = np.array(names)
            out[f"{ttype}_grad_norms"] = np.array([dict(model.named_parameters())[n].grad.norm().item() for n in names])
            out[f"{ttype}_trainable"] = np.array(names)
        model.eval()
        with torch.no_grad():
            out[f"{ttype}_logits_eval"] = model(pts)
        if ttype == "full":
            bn = model.cls_head_finetune[1]
            out["full_head_bn_running_mean"] = bn.running_mean.clone()
            out["full_head_bn_running_var"] = bn.running_var.clone()

    # PointcloudRotate with injected angles
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_dt", f"{REF}/datasets/data_transforms.py")
    dt = importlib.util.module_from_spec(spec); spec.loader.exec_module(dt)
    pc = torch.from_numpy(clouds(11, 3, 64)).clone()
    u = [0.125, 0.61, 0.93]
    seq = iter(u)
    real_u = np.random.uniform
    np.random.uniform = lambda *a, **k: next(seq)
    try:
        rot = dt.PointcloudRotate()(pc.clone())
    finally:
        np.random.uniform = real_u
    out["rotate_u"] = np.array(u)
    out["rotate_out"] = rot
    save("g10_finetune", **out)


if __name__ == "__main__":
    main()
