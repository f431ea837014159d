#!/usr/bin/env python3
"""Generates tests/golden/g13_cls_loss.npz: the reference's `cls_loss: True` branch of ACT_PointDistillation.forward
(models/act.py:1208-1249 with the shallow hook of VisableOnlyMaskTransformer.forward :293-307) on the tiny Stage-II geometry
(depth 3, register_shallow_hook 1), injected mask / gumbel draws.  Run in the build container (needs /root/reference).

    python tests/golden/make_golden_clsloss.py
"""
import copy
import os
import sys
import numpy as np
import torch
import torch.nn.functional as F
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import install_shims, save, REF, STUB_VIT                     # noqa: E402
from fill import fill_module, clouds, TINY_STAGE2, TINY_B, TINY_N              # noqa: E402


def config():
    cfg = copy.deepcopy(TINY_STAGE2)
    cfg["transformer_config"].update(depth=3, cls_loss=True, register_shallow_hook=1)
    return cfg


def main():
    os.chdir(REF)
    install_shims()
    STUB_VIT.update(dim=128, depth=2, heads=2)
    import models.dvae as dvae
    from models import build_model_from_cfg
    from easydict import EasyDict
    torch.set_num_threads(8)
    cfg = EasyDict(config())
    tok_model = dvae.ACTPromptedDiscreteVAEwithVIT(cfg.dvae_config)
    real_load = torch.load
    torch.load = lambda *a, **k: {"base_model": tok_model.state_dict()}
    try:
        model = build_model_from_cfg(cfg)
    finally:
        torch.load = real_load
    fill_module(model, "g13.")
    model.dvae_tokenizer.prompt_dropout.p = 0.0
    model.train()
    B, N, G = TINY_B, TINY_N, cfg.dvae_config.num_group
    pts = torch.from_numpy(clouds(13, B, N))
    nmask = int(cfg.transformer_config.mask_ratio * G)
    rs = np.random.RandomState(113)
    mask = np.zeros((B, G), dtype=bool)
    for b in range(B):
        mask[b, rs.permutation(G)[:nmask]] = True
    mask_t = torch.from_numpy(mask)
    model.ACT_encoder._mask_center_rand = lambda center, noaug=False: mask_t
    real_gs = F.gumbel_softmax

    def seeded_gumbel(logits, tau=1.0, hard=False, eps=1e-10, dim=-1):
        torch.manual_seed(777)
        return real_gs(logits, tau=tau, hard=hard, dim=dim)
    F.gumbel_softmax = seeded_gumbel
    try:
        loss = model(pts)
        loss.backward()
    finally:
        F.gumbel_softmax = real_gs
    names = ["ACT_encoder.blocks.blocks.0.attn.qkv.weight", "ACT_encoder.blocks.blocks.1.mlp.fc1.weight",
             "ACT_encoder.blocks.blocks.2.mlp.fc2.weight", "ACT_encoder.cls_token", "ACT_encoder.cls_pos", "cls_pos", "ACT_encoder.norm.weight",
             "mask_token", "ACT_decoder.blocks.0.attn.proj.weight", "proj_head.weight", "decoder_pos_embed.0.weight",
             "ACT_encoder.encoder.first_conv.0.weight"]
    pd = dict(model.named_parameters())
    # mask_ratio: 0 on the plain tiny geometry (models/act.py:1175-1178,1238-1240): no decoder, every token regressed
    # (with loss: cosine the reference itself dies there -- `student_feat_global` is unbound at models/act.py:1248 -- so the golden is taken with 'l2')
    cfg0 = EasyDict(copy.deepcopy(TINY_STAGE2)); cfg0.transformer_config.mask_ratio = 0; cfg0.loss = "l2"
    tok0 = dvae.ACTPromptedDiscreteVAEwithVIT(cfg0.dvae_config)
    torch.load = lambda *a, **k: {"base_model": tok0.state_dict()}
    try:
        m0 = build_model_from_cfg(cfg0)
    finally:
        torch.load = real_load
    fill_module(m0, "nm.")
    m0.dvae_tokenizer.prompt_dropout.p = 0.0
    m0.train()
    F.gumbel_softmax = seeded_gumbel
    try:
        l0 = m0(torch.from_numpy(clouds(19, TINY_B, TINY_N)))
        l0.backward()
    finally:
        F.gumbel_softmax = real_gs
    pd0 = dict(m0.named_parameters())
    n0 = ["ACT_encoder.blocks.blocks.1.attn.proj.weight", "proj_head.weight", "ACT_encoder.encoder.second_conv.3.weight"]
    save("g13_cls_loss", nomask_loss=np.array([l0.item()], dtype=np.float64), nomask_grad_names=np.array(n0),
         nomask_grad_norms=np.array([pd0[n].grad.norm().item() for n in n0], dtype=np.float64),
         nomask_state_dict_keys=np.array(sorted(k for k in m0.state_dict() if not k.startswith("dvae_tokenizer."))), mask=mask, loss=np.array([loss.item()], dtype=np.float64), grad_names=np.array(names),
         grad_norms=np.array([pd[n].grad.norm().item() for n in names], dtype=np.float64),
         grad_cls_pos=pd["cls_pos"].grad.clone(),
         state_dict_keys=np.array(sorted(k for k in model.state_dict() if not k.startswith("dvae_tokenizer."))))


if __name__ == "__main__":
    main()
