#!/usr/bin/env python3
"""Generates tests/golden/g16_prompt_variants.npz: the non-default configurations of the reference's ACTPromptedDiscreteVAEwithVIT
(models/dvae.py:513-534): `shallow` = prompts prepended once (use_deep_prompt false), `noprompt` = frozen Transformer without prompts (its
output is computed under no_grad there, :522-524, so nothing upstream of it receives a reconstruction gradient), `novit` = visual_embed_dim
'none' (no image Transformer at all).  Tiny geometry, soft gumbel tau 0.7 seeded with 777 per call, prompt dropout off.

    python tests/golden/make_golden_promptvariants.py
"""
import os
import sys
import numpy as np
import torch
import torch.nn.functional as F
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import install_shims, save, REF, STUB_VIT                     # noqa: E402
from fill import fill_module, clouds, TINY_STAGE2, TINY_B, TINY_N, PROMPT_VARIANTS   # noqa: E402


def main():
    os.chdir(REF)
    install_shims()
    STUB_VIT.update(dim=128, depth=2, heads=2)
    import models.dvae as dvae
    from easydict import EasyDict
    torch.set_num_threads(8)
    pts = torch.from_numpy(clouds(16, TINY_B, TINY_N))
    real_gs = F.gumbel_softmax

    def seeded_gumbel(logits, tau=1.0, hard=False, eps=1e-10, dim=-1):
        torch.manual_seed(777)
        return real_gs(logits, tau=tau, hard=hard, dim=dim)
    out = {}
    for tag, over in PROMPT_VARIANTS.items():
        cfg = dict(TINY_STAGE2["dvae_config"]); cfg.update(over)
        torch.manual_seed(16)
        vae = fill_module(dvae.ACTPromptedDiscreteVAEwithVIT(EasyDict(cfg)), f"g16.{tag}.")
        if hasattr(vae, "prompt_dropout"):
            vae.prompt_dropout.p = 0.0
        vae.train()
        F.gumbel_softmax = seeded_gumbel
        try:
            ret = vae(pts, temperature=0.7, hard=False)
            lr, lk = vae.get_loss(ret, pts)
            (lr + 0.1 * lk).backward()
            with torch.no_grad():
                nb, c = vae.group_divider(pts)
                feat = vae.forward_tokenizer_features(nb, c)
        finally:
            F.gumbel_softmax = real_gs
        names = sorted(n for n, p in vae.named_parameters() if p.grad is not None)
        pd = dict(vae.named_parameters())
        out[f"{tag}.fine"] = ret[3]; out[f"{tag}.logits"] = ret[5]
        out[f"{tag}.loss"] = np.array([lr.item(), lk.item()], dtype=np.float64)
        out[f"{tag}.grad_names"] = np.array(names)
        out[f"{tag}.grad_norms"] = np.array([pd[n].grad.norm().item() for n in names], dtype=np.float64)
        out[f"{tag}.tokenizer_feat"] = feat
        out[f"{tag}.state_dict_keys"] = np.array(sorted(vae.state_dict().keys()))
    save("g16_prompt_variants", **out)


if __name__ == "__main__":
    main()
