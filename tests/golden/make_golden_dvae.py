#!/usr/bin/env python3
"""Generates tests/golden/g15_dvae.npz: the reference's plain Point-BERT tokenizer `DiscreteVAE` (models/dvae.py:278-358, recipe
cfgs/autoencoder/pointbert_dvae.yaml) on a tiny geometry: Stage-I forward with soft gumbel (tau 0.7) + get_loss + gradient norms, and the
frozen `forward_tokenizer_features` (hard gumbel, tau 1).  The gumbel noise is the torch stream seeded with 777 at every call, which the
oracle / HIP tests regenerate.  Run in the build container (needs /root/reference).

    python tests/golden/make_golden_dvae.py
"""
import os
import sys
import numpy as np
import torch
import torch.nn.functional as F
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import install_shims, save, REF                               # noqa: E402
from fill import fill_module, clouds, TINY_DVAE, TINY_B, TINY_N                # noqa: E402


def main():
    os.chdir(REF)
    install_shims()
    import models.dvae as dvae
    from models import build_model_from_cfg
    from easydict import EasyDict
    torch.set_num_threads(8)
    torch.manual_seed(15)
    vae = fill_module(build_model_from_cfg(EasyDict(dict(TINY_DVAE))), "g15.")
    assert type(vae) is dvae.DiscreteVAE
    vae.train()
    pts = torch.from_numpy(clouds(15, TINY_B, TINY_N))
    real_gs = F.gumbel_softmax

    def seeded_gumbel(logits, tau=1.0, hard=False, eps=1e-10, dim=-1):
        torch.manual_seed(777)
        return real_gs(logits, tau=tau, hard=hard, dim=dim)
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
    names = ["encoder.first_conv.0.weight", "encoder.second_conv.3.weight", "dgcnn_1.input_trans.weight", "dgcnn_1.layer5.0.weight", "codebook",
             "dgcnn_2.layer1.0.weight", "dgcnn_2.layer5.1.weight", "decoder.mlp.0.weight", "decoder.final_conv.6.weight"]
    pd = dict(vae.named_parameters())
    save("g15_dvae", coarse=ret[2], fine=ret[3], logits=ret[5], whole_fine=ret[1], whole_coarse=ret[0],
         loss=np.array([lr.item(), lk.item()], dtype=np.float64), grad_names=np.array(names),
         grad_norms=np.array([pd[n].grad.norm().item() for n in names], dtype=np.float64), grad_codebook=pd["codebook"].grad,
         tokenizer_feat=feat, state_dict_keys=np.array(sorted(vae.state_dict().keys())))


if __name__ == "__main__":
    main()
