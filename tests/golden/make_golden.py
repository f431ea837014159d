#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/*.npz by IMPORTING THE REFERENCE'S OWN
PYTHON MODULES from /root/reference (build container only; the reference never travels).

Third-party packages the reference imports but this image lacks are replaced by import
shims so that the reference's *in-tree* code (models/dvae.py, models/act.py,
utils/transformer_layers.py, tools/builder.py, extensions/chamfer_dist/__init__.py,
part_segmentation/models/pointnet2_utils.py, datasets/data_transforms.py) executes
unmodified on CPU.  What each shim stands for:

  pointnet2_ops.furthest_point_sample -> the reference's in-tree pure-torch
        farthest_point_sample (part_segmentation/models/pointnet2_utils.py:60-81) with its
        random start index forced to 0 (torch.randint patched);   gather -> index gather
  knn_cuda.KNN        -> direct-difference brute force, stable ascending sort (our convention,
        NOT a pin of the KNN_CUDA binary); agreement with the in-tree knn_point is recorded
  timm                -> DropPath / trunc_normal_ from the reference's utils/transformer_layers.py;
        create_model -> 12 (or tiny) x the reference's models.act.Block(qkv_bias=True, LN eps 1e-6)
  lightly.loss.NegativeCosineSimilarity -> -F.cosine_similarity(x0,x1,dim=1,eps=1e-8).mean()
  chamfer (CUDA ext)  -> numpy restatement (kernel numerics are NOT pinned by these goldens;
        only the Python-side reductions of extensions/chamfer_dist/__init__.py are)
  easydict / termcolor / mmcv / h5py / tensorboardX -> trivial shims

Run:  python tests/golden/make_golden.py
"""
import os
import sys
import types
import zlib
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
from fill import fill_module, fill_tensor, clouds, TINY_STAGE2, TINY_B, TINY_N  # noqa: E402
from oracle import point_ops as OP  # noqa: E402  (only for the chamfer / knn shims documented above)

STUB_VIT = dict(depth=12, heads=12)


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_shims():
    sys.path.insert(0, REF)

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            for k, v in dict(d or {}, **kw).items():
                setattr(self, k, v)

        def __setattr__(self, k, v):
            if isinstance(v, dict) and not isinstance(v, EasyDict):
                v = EasyDict(v)
            super().__setitem__(k, v)
            super().__setattr__(k, v)
        __setitem__ = __setattr__
    _mod("easydict", EasyDict=EasyDict)
    _mod("termcolor", colored=lambda s, *a, **k: s)
    _mod("mmcv"); _mod("mmcv.utils", collect_env=lambda: {})
    _mod("h5py"); _mod("tensorboardX", SummaryWriter=object)

    # in-tree pieces used by shims
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_pn2_utils", f"{REF}/part_segmentation/models/pointnet2_utils.py")
    pn2 = importlib.util.module_from_spec(spec); spec.loader.exec_module(pn2)
    spec = importlib.util.spec_from_file_location("ref_tl", f"{REF}/utils/transformer_layers.py")
    tl = importlib.util.module_from_spec(spec); spec.loader.exec_module(tl)

    def furthest_point_sample(xyz, npoint):
        real = torch.randint
        torch.randint = lambda lo, hi, size, **kw: torch.zeros(size, dtype=kw.get("dtype", torch.long))
        try:
            return pn2.farthest_point_sample(xyz, npoint).to(torch.int32)
        finally:
            torch.randint = real

    def gather_operation(features, idx):          # [B,C,N], [B,S] -> [B,C,S]
        return torch.gather(features, 2, idx.long().unsqueeze(1).expand(-1, features.shape[1], -1))
    _mod("pointnet2_ops")
    _mod("pointnet2_ops.pointnet2_utils", furthest_point_sample=furthest_point_sample,
         gather_operation=gather_operation)
    sys.modules["pointnet2_ops"].pointnet2_utils = sys.modules["pointnet2_ops.pointnet2_utils"]

    class KNN(nn.Module):
        def __init__(self, k, transpose_mode=False):
            super().__init__(); self.k, self.t = k, transpose_mode

        @torch.no_grad()
        def forward(self, ref, query):
            if not self.t:
                ref, query = ref.transpose(1, 2), query.transpose(1, 2)
            d, i = OP.knn_ref(ref.contiguous().numpy(), query.contiguous().numpy(), self.k)
            d, i = torch.from_numpy(d), torch.from_numpy(i)
            if not self.t:
                d, i = d.transpose(1, 2).contiguous(), i.transpose(1, 2).contiguous()
            return d, i
    _mod("knn_cuda", KNN=KNN)

    class _Vit(nn.Module):
        def __init__(self, dim):
            super().__init__()
            import functools
            from models.act import Block
            self.embed_dim = dim
            self.blocks = nn.Sequential(*[Block(dim, STUB_VIT["heads"], qkv_bias=True,
                                                norm_layer=functools.partial(nn.LayerNorm, eps=1e-6))
                                          for _ in range(STUB_VIT["depth"])])
            self.norm = nn.LayerNorm(dim, eps=1e-6)

    def create_model(name, pretrained=False, **kw):
        return _Vit(STUB_VIT["dim"])
    _mod("timm", create_model=create_model)
    _mod("timm.models"); _mod("timm.models.layers", trunc_normal_=tl.trunc_normal_, DropPath=tl.DropPath)
    _mod("timm.scheduler", CosineLRScheduler=object)

    class NegativeCosineSimilarity(nn.Module):
        def __init__(self, dim=1, eps=1e-8):
            super().__init__(); self.dim, self.eps = dim, eps

        def forward(self, x0, x1):
            return -F.cosine_similarity(x0, x1, self.dim, self.eps).mean()
    _mod("lightly"); _mod("lightly.loss", NegativeCosineSimilarity=NegativeCosineSimilarity)
    sys.modules["lightly"].loss = sys.modules["lightly.loss"]

    def ch_fwd(a, b):
        d1, d2, i1, i2 = OP.chamfer_fwd_ref(a.detach().numpy(), b.detach().numpy())
        return [torch.from_numpy(d1), torch.from_numpy(d2), torch.from_numpy(i1), torch.from_numpy(i2)]

    def ch_bwd(a, b, i1, i2, g1, g2):
        x, y = OP.chamfer_bwd_ref(a.detach().numpy(), b.detach().numpy(), i1.numpy(), i2.numpy(), g1.numpy(), g2.numpy())
        return [torch.from_numpy(x).float(), torch.from_numpy(y).float()]
    _mod("chamfer", forward=ch_fwd, backward=ch_bwd)

    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    return pn2, tl


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB  keys={list(out)}")


def main():
    os.chdir(REF)
    pn2, tl = install_shims()
    STUB_VIT.update(dim=768)
    import models.dvae as dvae
    import models.act as act
    from models import build_model_from_cfg
    from easydict import EasyDict
    torch.set_num_threads(8)

    # ---- G1 group: FPS from the in-tree restatement, kNN convention, knn_point set agreement ----
    pts = torch.from_numpy(clouds(0, 4, 1024))
    fidx = sys.modules["pointnet2_ops.pointnet2_utils"].furthest_point_sample(pts, 64)
    grp = dvae.Group(num_group=64, group_size=32)
    nb, center = grp(pts)
    _, kidx = grp.knn(pts, center)
    ref_sets = dvae.knn_point(32, pts, center)                       # in-tree expansion-form kNN (unsorted)
    agree = sum(int(set(a.tolist()) == set(b.tolist())) for a, b in
                zip(kidx.reshape(-1, 32), ref_sets.reshape(-1, 32)))
    # FPS at larger N / G too (stress geometry subset), still in-tree code
    pts_big = torch.from_numpy(clouds(1, 2, 4096))
    fidx_big = sys.modules["pointnet2_ops.pointnet2_utils"].furthest_point_sample(pts_big, 256)
    save("g1_group", fps_idx=fidx, center=center, knn_idx=kidx, neighborhood=nb,
         knn_point_set_agree=np.array([agree, kidx.shape[0] * kidx.shape[1]]),
         knn_point_sorted=np.sort(ref_sets.numpy(), axis=-1).astype(np.int16), fps_idx_big=fidx_big)

    # ---- G2 mini-PointNet Encoder(128) + TransformerEncoder(128, depth 2, heads 2) ----
    enc = fill_module(dvae.Encoder(128), "g2.enc.")
    tenc = fill_module(act.TransformerEncoder(embed_dim=128, depth=2, num_heads=2, drop_path_rate=0.0), "g2.tenc.")
    pos = fill_tensor("g2.pos", (4, 64, 128), "b")
    enc.train(); tok_train = enc(nb)
    rm1, rv1 = enc.first_conv[1].running_mean.clone(), enc.first_conv[1].running_var.clone()
    enc.eval(); tok_eval = enc(nb)
    out = tenc(tok_train, pos)
    save("g2_encoder", tok_train=tok_train, tok_eval=tok_eval, out=out, bn1_running_mean=rm1, bn1_running_var=rv1)

    # ---- G3 Block(384, 6): fwd, input grad, weight-grad norms (act.Block and the stand-alone copy) ----
    blk = fill_module(act.Block(384, 6), "g3.blk.")
    x = fill_tensor("g3.x", (2, 14, 384), "code").requires_grad_(True)
    y = blk(x)
    w = fill_tensor("g3.w", (2, 14, 384), "code")
    (y * w).sum().backward()
    gn = {n: p.grad.norm().item() for n, p in blk.named_parameters()}
    blk2 = tl.Block(384, 6); blk2.load_state_dict(blk.state_dict())
    y2 = blk2(x.detach())
    save("g3_block", y=y, dx=x.grad, y_standalone=y2, grad_names=np.array(list(gn.keys())),
         grad_norms=np.array(list(gn.values()), dtype=np.float64))
    # teacher-style block (qkv bias, eps 1e-6), S=128 exercised at reduced width
    import functools
    blk_t = fill_module(act.Block(128, 2, qkv_bias=True, norm_layer=functools.partial(nn.LayerNorm, eps=1e-6)), "g3.blkt.")
    xt = fill_tensor("g3.xt", (2, 128, 128), "code")
    save("g3_block_teacher", y=blk_t(xt))

    # ---- G4 / G8 tiny Stage-II: loss, grads, 2 AdamW steps through the reference's param groups ----
    STUB_VIT.update(dim=128, depth=2, heads=2)
    real_load = torch.load
    cfg = EasyDict(TINY_STAGE2)
    tok_model = dvae.ACTPromptedDiscreteVAEwithVIT(cfg.dvae_config)
    torch.load = lambda *a, **k: {"base_model": tok_model.state_dict()}
    try:
        model = build_model_from_cfg(cfg)
    finally:
        torch.load = real_load
    fill_module(model, "g4.")
    model.dvae_tokenizer.prompt_dropout.p = 0.0
    model.train()
    B, N, G = TINY_B, TINY_N, cfg.dvae_config.num_group
    tpts = torch.from_numpy(clouds(4, B, N))
    nmask = int(cfg.transformer_config.mask_ratio * G)
    rs = np.random.RandomState(44)
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
    loss = model(tpts)
    loss.backward()
    names = ["ACT_encoder.blocks.blocks.0.attn.qkv.weight", "ACT_encoder.encoder.first_conv.0.weight", "mask_token",
             "ACT_encoder.encoder.second_conv.3.weight", "ACT_decoder.blocks.1.mlp.fc2.weight", "proj_head.weight",
             "ACT_encoder.pos_embed.0.weight", "decoder_pos_embed.2.bias", "ACT_encoder.cls_token",
             "ACT_encoder.norm.weight"]
    pd = dict(model.named_parameters())
    gnorm = np.array([pd[n].grad.norm().item() for n in names], dtype=np.float64)
    g_qkv = pd[names[0]].grad.clone()
    teacher = model.dvae_tokenizer.forward_tokenizer_features(*model.group_divider(tpts))
    # G8: reference optimizer param groups (tools/builder.py:38-55), 2 steps
    import importlib.util as ilu
    spec = ilu.spec_from_file_location("ref_builder", f"{REF}/tools/builder.py")   # bypass tools/__init__ (torchvision)
    builder = ilu.module_from_spec(spec); spec.loader.exec_module(builder)
    wrap = types.SimpleNamespace(module=model)
    ocfg = EasyDict(optimizer=dict(type="AdamW", kwargs=dict(lr=1e-3, weight_decay=0.05)),
                    scheduler=dict(type="function", kwargs={}))
    optimizer, _ = builder.build_opti_sche(wrap, ocfg)
    n_nodecay = len(optimizer.param_groups[0]["params"]); n_decay = len(optimizer.param_groups[1]["params"])
    losses = [loss.item()]
    optimizer.step(); model.zero_grad()
    loss2 = model(tpts); loss2.backward(); optimizer.step(); model.zero_grad()
    losses.append(loss2.item())
    after = np.array([pd[n].detach().norm().item() for n in names[:3]], dtype=np.float64)
    F.gumbel_softmax = real_gs
    save("g4_stage2", pts=tpts, mask=mask, loss=np.array(losses, dtype=np.float64), grad_names=np.array(names),
         grad_norms=gnorm, grad_qkv0=g_qkv, teacher_feat=teacher, n_param_groups=np.array([n_nodecay, n_decay]),
         norms_after_2_steps=after)

    # ---- G7 tiny Stage-I forward (soft gumbel, tau=0.7) + get_loss ----
    vae = dvae.ACTPromptedDiscreteVAEwithVIT(cfg.dvae_config)
    fill_module(vae, "g7.")
    vae.prompt_dropout.p = 0.0
    vae.train()
    F.gumbel_softmax = seeded_gumbel
    ret = vae(tpts, temperature=0.7, hard=False)
    lr, lk = vae.get_loss(ret, tpts)
    (lr + 0.1 * lk).backward()
    vn = ["encoder.first_conv.0.weight", "dgcnn_1.layer5.0.weight", "codebook", "deep_prompt_tokens", "proj_pre.weight",
          "dgcnn_2.layer1.0.weight", "decoder.mlp.0.weight", "decoder.final_conv.6.weight", "visual_pos_embed.0.weight"]
    vpd = dict(vae.named_parameters())
    F.gumbel_softmax = real_gs
    save("g7_stage1", coarse=ret[2], fine=ret[3], logits=ret[5], whole_fine=ret[1],
         loss=np.array([lr.item(), lk.item()], dtype=np.float64), grad_names=np.array(vn),
         grad_norms=np.array([vpd[n].grad.norm().item() for n in vn], dtype=np.float64))

    # ---- G5 chamfer reductions (Python side of extensions/chamfer_dist/__init__.py) ----
    from extensions.chamfer_dist import ChamferDistanceL1, ChamferDistanceL2
    xa = fill_tensor("g5.x", (4, 64, 3), "code"); ya = fill_tensor("g5.y", (4, 128, 3), "code")
    save("g5_chamfer", l1=ChamferDistanceL1()(xa, ya), l2=ChamferDistanceL2()(xa, ya))

    # ---- G6 cosine loss loop (models/act.py:1243-1254) is covered by G4; direct vector too ----
    s = fill_tensor("g6.s", (4, 51, 384), "code"); t = fill_tensor("g6.t", (4, 51, 384), "code")
    lf = sys.modules["lightly.loss"].NegativeCosineSimilarity()
    acc = torch.zeros(1)
    for b in range(4):
        acc += (1 + lf(s[b], t[b]).mean())
    save("g6_cosine", loss=(acc.mean() / 4))

    # ---- G9 augmentation with injected draws (datasets/data_transforms.py:20-34) ----
    spec = __import__("importlib.util").util.spec_from_file_location("ref_dt", f"{REF}/datasets/data_transforms.py")
    dt = __import__("importlib.util").util.module_from_spec(spec); spec.loader.exec_module(dt)
    pc = torch.from_numpy(clouds(9, 2, 128)).clone()
    rs = np.random.RandomState(9)
    draws = [rs.uniform(2. / 3., 3. / 2., 3), rs.uniform(-0.2, 0.2, 3), rs.uniform(2. / 3., 3. / 2., 3), rs.uniform(-0.2, 0.2, 3)]
    seq = iter(draws)
    real_u = np.random.uniform
    np.random.uniform = lambda low, high, size: next(seq)
    try:
        outp = dt.PointcloudScaleAndTranslate()(pc.clone())
    finally:
        np.random.uniform = real_u
    save("g9_augment", scale=np.stack(draws[0::2]), shift=np.stack(draws[1::2]), out=outp)


if __name__ == "__main__":
    main()
