#!/usr/bin/env python3
"""Generates tests/golden/g11_state_dict.npz: the full state_dict listing (parameter / buffer name -> shape) of the REFERENCE's
ACT_PointDistillation, ACTPromptedDiscreteVAEwithVIT and PointTransformer at the tiny test configs, obtained by importing the
reference's own classes under the shims of make_golden.py.  The product models must reproduce these listings exactly
(checkpoint wire compatibility).  Run in the build container (needs /root/reference).

    python tests/golden/make_golden_statedict.py
"""
import os
import sys
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import install_shims, save, REF, STUB_VIT              # noqa: E402
from fill import TINY_STAGE2, TINY_FINETUNE                             # noqa: E402


def listing(model):
    sd = model.state_dict()
    return np.array(list(sd.keys())), np.array(["x".join(str(d) for d in v.shape) for v in sd.values()])


def main():
    os.chdir(REF)
    install_shims()
    STUB_VIT.update(dim=128, depth=2, heads=2)
    import models.dvae as dvae
    import models.act as act                                             # noqa: F401
    from models import build_model_from_cfg
    from easydict import EasyDict
    out = {}
    cfg = EasyDict(TINY_STAGE2)
    vae = dvae.ACTPromptedDiscreteVAEwithVIT(cfg.dvae_config)
    out["dvae_names"], out["dvae_shapes"] = listing(vae)
    real_load = torch.load
    torch.load = lambda *a, **k: {"base_model": vae.state_dict()}
    try:
        s2 = build_model_from_cfg(cfg)
    finally:
        torch.load = real_load
    out["stage2_names"], out["stage2_shapes"] = listing(s2)
    for ttype in ("full", "linear", "side"):
        ft = build_model_from_cfg(EasyDict(dict(TINY_FINETUNE, transfer_type=ttype)))
        out[f"ft_{ttype}_names"], out[f"ft_{ttype}_shapes"] = listing(ft)
    save("g11_state_dict", **out)


if __name__ == "__main__":
    main()
