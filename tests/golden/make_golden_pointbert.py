#!/usr/bin/env python3
"""Generates tests/golden/g14_pointbert.npz: one forward / backward of the reference's ACT_PointBERT (models/act.py:913-1096, MaskTransformer
:532-725) on a tiny geometry, with every random draw recorded in call order (python `random.random`, `torch.rand`, `torch.randperm`) so that
the oracle and the HIP path can replay them.  Run in the build container (needs /root/reference).

    python tests/golden/make_golden_pointbert.py
"""
import copy
import os
import random
import sys
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import install_shims, save, REF, STUB_VIT                     # noqa: E402
from fill import fill_module, clouds, TINY_STAGE2, TINY_POINTBERT              # noqa: E402


def main():
    os.chdir(REF)
    install_shims()
    STUB_VIT.update(dim=128, depth=2, heads=2)
    import models.dvae as dvae
    from models import build_model_from_cfg
    from easydict import EasyDict
    torch.set_num_threads(8)
    cfg = EasyDict(copy.deepcopy(TINY_POINTBERT))
    tok_model = fill_module(dvae.ACTPromptedDiscreteVAEwithVIT(cfg.dvae_config), "g14.dvae.")
    real_load = torch.load
    torch.load = lambda *a, **k: {"base_model": {"module." + k2: v for k2, v in tok_model.state_dict().items()}}
    try:
        torch.manual_seed(3)
        model = build_model_from_cfg(cfg)
    finally:
        torch.load = real_load
    fill_module(model.transformer_q, "g14.q.")
    with torch.no_grad():
        for pq, pk in zip(model.transformer_q.parameters(), model.transformer_k.parameters()):
            pk.copy_(0.5 * pq)                                        # a key encoder that differs from the query encoder
        model.transformer_q.encoder.load_state_dict(tok_model.encoder.state_dict())    # what _prepare_encoder did before the fill
        queue0 = torch.nn.functional.normalize(torch.from_numpy(np.random.RandomState(14).standard_normal((cfg.transformer_config.cls_dim, cfg.K)).astype(np.float32)), dim=0)
        model.queue.copy_(queue0)
    model.train()
    B, N = 4, 128
    pts = torch.from_numpy(clouds(14, B, N))
    tape = []
    real_rand, real_perm, real_rr = torch.rand, torch.randperm, random.random

    def rec_rand(*a, **k):
        t = real_rand(*a, **k); tape.append(("rand", t.clone())); return t

    def rec_perm(*a, **k):
        t = real_perm(*a, **k); tape.append(("perm", t.clone())); return t

    def rec_rr():
        v = real_rr(); tape.append(("rr", v)); return v
    torch.manual_seed(1414); random.seed(1414)
    torch.rand, torch.randperm, random.random = rec_rand, rec_perm, rec_rr
    try:
        moco, dv, cm = model(pts)
        (moco + dv + cm).backward()
    finally:
        torch.rand, torch.randperm, random.random = real_rand, real_perm, real_rr
    kinds = [k for k, _ in tape]
    assert kinds == ["rr", "rand", "rand", "perm", "rand", "rand", "rr", "rand", "rand", "perm", "rr", "rand", "rand", "perm"], kinds
    lo, hi = cfg.transformer_config.mask_ratio
    v = [t for _, t in tape]
    draws = {"q.ratio": v[0] * (hi - lo) + lo, "q.mask_u": v[1], "q.replace_u": v[2], "q.perm": v[3], "mixup_ratio": v[4], "mixup_u": v[5],
             "mix.ratio": v[6] * (hi - lo) + lo, "mix.mask_u": v[7], "mix.replace_u": v[8], "mix.perm": v[9],
             "k.ratio": v[10] * (hi - lo) + lo, "k.mask_u": v[11], "k.replace_u": v[12], "k.perm": v[13]}
    names = ["transformer_q.blocks.blocks.0.attn.qkv.weight", "transformer_q.encoder.first_conv.0.weight", "transformer_q.mask_token",
             "transformer_q.cls_token", "transformer_q.lm_head.weight", "transformer_q.cls_head.2.weight", "transformer_q.reduce_dim.weight",
             "transformer_q.pos_embed.0.weight"]
    pd = dict(model.named_parameters())
    out = {"draw." + k: (np.float64(x) if isinstance(x, float) else x) for k, x in draws.items()}
    save("g14_pointbert", losses=np.array([moco.item(), dv.item(), cm.item()], dtype=np.float64), grad_names=np.array(names),
         grad_norms=np.array([pd[n].grad.norm().item() for n in names], dtype=np.float64), queue0=queue0, queue1=model.queue,
         queue_ptr=model.queue_ptr, key_norm_after=np.array([pd["transformer_k.blocks.blocks.1.mlp.fc1.weight"].norm().item()]),
         state_dict_keys=np.array(sorted(k for k in model.state_dict() if not k.startswith("dvae."))), **out)


if __name__ == "__main__":
    main()
