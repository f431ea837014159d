"""Oracle (TEST INFRASTRUCTURE): pure-PyTorch CPU restatement of the ACT Stage-II /
Stage-I models with injectable randomness.  state_dict keys equal the reference's.

Reference files restated (file:line in /root/reference):
  Group                               models/dvae.py:154-183
  VisableOnlyMaskTransformer          models/act.py:148-309
  ACTPromptedDiscreteVAEwithVIT       models/dvae.py:360-615
  ACT_PointDistillation               models/act.py:1099-1258
  ChamferDistanceL1/L2                extensions/chamfer_dist/__init__.py:13-84
  PointTransformer (finetune / inference) models/act.py:727-910
"""
import math
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import point_ops as P
from .layers import (Draws, Encoder, DGCNN, Decoder, TransformerEncoder, TransformerDecoder, BlockList,
                     cosine_distill_loss, pairwise_distill_loss, trunc_normal_, knn_graph_ref)


class edict(dict):
    """minimal EasyDict stand-in (the reference uses easydict, utils/config.py:2)."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            self[k] = v

    def __setitem__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, edict):
            v = edict(v)
        super().__setitem__(k, v)

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    __setattr__ = __setitem__


class Group(nn.Module):
    def __init__(self, num_group, group_size):
        super().__init__()
        self.num_group, self.group_size = num_group, group_size

    def forward(self, xyz):
        nb, center, _, _ = P.group_ref(xyz.detach().cpu().numpy(), self.num_group, self.group_size)
        return torch.from_numpy(nb), torch.from_numpy(center)


class _ChamferFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz1, xyz2):
        d1, d2, i1, i2 = P.chamfer_fwd_ref(xyz1.detach().numpy(), xyz2.detach().numpy())
        if xyz1.dtype == torch.float64:      # gradcheck path: distances in float64
            a, b = xyz1.detach().numpy(), xyz2.detach().numpy()
            B = a.shape[0]
            d1 = ((a - b[np.arange(B)[:, None], i1]) ** 2).sum(-1)
            d2 = ((b - a[np.arange(B)[:, None], i2]) ** 2).sum(-1)
        ctx.save_for_backward(xyz1, xyz2, torch.from_numpy(i1), torch.from_numpy(i2))
        return torch.from_numpy(d1).to(xyz1.dtype), torch.from_numpy(d2).to(xyz1.dtype)

    @staticmethod
    def backward(ctx, g1, g2):
        xyz1, xyz2, i1, i2 = ctx.saved_tensors
        gx1, gx2 = P.chamfer_bwd_ref(xyz1.numpy(), xyz2.numpy(), i1.numpy(), i2.numpy(), g1.numpy(), g2.numpy())
        return torch.from_numpy(gx1).to(xyz1.dtype), torch.from_numpy(gx2).to(xyz2.dtype)


def chamfer_l1(xyz1, xyz2):
    d1, d2 = _ChamferFn.apply(xyz1, xyz2)
    return (torch.mean(torch.sqrt(d1)) + torch.mean(torch.sqrt(d2))) / 2


def chamfer_l2(xyz1, xyz2):
    d1, d2 = _ChamferFn.apply(xyz1, xyz2)
    return torch.mean(d1) + torch.mean(d2)


def rand_mask(B, G, num_mask, generator=None):
    """exactly num_mask ones per row (models/act.py:244-267 semantics, torch RNG)."""
    r = torch.rand(B, G, generator=generator)
    order = r.argsort(dim=1)
    mask = torch.zeros(B, G, dtype=torch.bool)
    mask.scatter_(1, order[:, :num_mask], True)
    return mask


def block_mask(center, num_mask, seed_index):
    """the num_mask centres nearest to centre[seed_index[b]] per cloud (models/act.py:215-242, random.randint injected)."""
    B, G, _ = center.shape
    mask = torch.zeros(B, G, dtype=torch.bool)
    for b in range(B):
        d = torch.norm(center[b, int(seed_index[b])].reshape(1, 3) - center[b], p=2, dim=-1)
        mask[b, torch.argsort(d, descending=False)[:num_mask]] = True
    return mask


class VisableOnlyMaskTransformer(nn.Module):
    def __init__(self, config):
        super().__init__()
        tc = config.transformer_config
        self.mask_type = tc.get("mask_type", "rand") if hasattr(tc, "get") else "rand"
        self.mask_ratio, self.embed_dim, self.cls_dim = tc.mask_ratio, tc.embed_dim, tc.cls_dim
        self.depth, self.num_heads = tc.depth, tc.num_heads
        self.encoder_dims = config.dvae_config.encoder_dims
        self.encoder = Encoder(self.encoder_dims)
        self.reduce_dim = nn.Linear(self.encoder_dims, self.embed_dim) if self.encoder_dims != self.embed_dim \
            else nn.Identity()
        self.cls_token = nn.Parameter(torch.randn(1, 1, self.embed_dim))
        self.cls_pos = nn.Parameter(torch.randn(1, 1, self.embed_dim))
        self.pos_embed = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, self.embed_dim))
        dpr = [x.item() for x in torch.linspace(0, tc.drop_path_rate, self.depth)]
        self.blocks = TransformerEncoder(self.embed_dim, self.depth, self.num_heads, dpr, tag="enc")
        self.norm = nn.LayerNorm(self.embed_dim)
        self.lm_head = nn.Linear(self.embed_dim, config.dvae_config.num_tokens)
        self.cls_head = nn.Sequential(nn.Linear(self.embed_dim, self.cls_dim), nn.GELU(),
                                      nn.Linear(self.cls_dim, self.cls_dim))
        trunc_normal_(self.cls_token); trunc_normal_(self.cls_pos)
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, (nn.Linear, nn.Conv1d)):
            trunc_normal_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0); nn.init.constant_(m.weight, 1.0)

    def forward(self, neighborhood, center, draws, only_cls_tokens=False, noaug=False, register_shallow_hook=-1):
        B, G, _ = center.shape
        if noaug or self.mask_ratio == 0:
            mask = torch.zeros(B, G, dtype=torch.bool)
        else:
            if self.mask_type == "block":
                seed = draws.get("mask_seed", lambda: torch.randint(0, G, (B,)))
                mask = block_mask(center, int(self.mask_ratio * G), seed)
            else:
                mask = draws.get("mask", lambda: rand_mask(B, G, int(self.mask_ratio * G)))
        tok = self.reduce_dim(self.encoder(neighborhood))
        C = tok.shape[-1]
        x_vis = tok[~mask].reshape(B, -1, C)
        pos = self.pos_embed(center[~mask].reshape(B, -1, 3))
        x_vis = torch.cat((self.cls_token.expand(B, -1, -1), x_vis), dim=1)
        pos = torch.cat((self.cls_pos.expand(B, -1, -1), pos), dim=1)
        shallow = None
        if register_shallow_hook > 0:              # models/act.py:293-297: keep the output of block `hook` (before the final norm)
            for idx, blk in enumerate(self.blocks.blocks):
                x_vis = blk(x_vis + pos, draws)
                if idx == register_shallow_hook:
                    shallow = x_vis
        else:
            x_vis = self.blocks(x_vis, pos, draws)
        x_vis = self.norm(x_vis)
        if only_cls_tokens:
            return self.cls_head(x_vis[:, 0])
        if register_shallow_hook > 0:
            return x_vis[:, 1:], x_vis[:, 0], shallow[:, 1:], mask
        return x_vis[:, 1:], mask


class DiscreteVAE(nn.Module):
    """The plain Point-BERT tokenizer (models/dvae.py:278-358): Group -> mini-PointNet -> DGCNN -> gumbel-softmax over the codebook ->
    DGCNN -> FoldingNet.  ACTPromptedDiscreteVAEwithVIT (:360-615) is this plus the prompt-tuned image Transformer between the codebook
    lookup and dgcnn_2."""

    def __init__(self, config):
        super().__init__()
        c = config
        self.group_size, self.num_group = c.group_size, c.num_group
        self.encoder_dims, self.tokens_dims = c.encoder_dims, c.tokens_dims
        self.decoder_dims, self.num_tokens = c.decoder_dims, c.num_tokens
        self.group_divider = Group(self.num_group, self.group_size)
        self.encoder = Encoder(self.encoder_dims)
        self.dgcnn_1 = DGCNN(self.encoder_dims, self.num_tokens)
        self.codebook = nn.Parameter(torch.randn(self.num_tokens, self.tokens_dims))
        self.dgcnn_2 = DGCNN(self.tokens_dims, self.decoder_dims)
        self.decoder = Decoder(self.decoder_dims, self.group_size)

    def visual_embedding(self, x, center, draws):
        return x

    def _gumbel(self, logits, tau, hard, draws):
        g = draws.get("gumbel", lambda: -torch.empty_like(logits).exponential_().log())
        y = (logits + g) / tau
        if hard:
            index = y.argmax(dim=-1)
            return self.codebook[index]                 # == einsum(one_hot, codebook) (models/dvae.py:587-588)
        return torch.einsum("bgn,nc->bgc", y.softmax(dim=-1), self.codebook)

    def forward_tokenizer_features(self, neighborhood, center, draws, return_global=True):
        idx = knn_graph_ref(center, 4)
        logits = self.dgcnn_1(self.encoder(neighborhood), center, idx)
        sampled = self._gumbel(logits, 1.0, True, draws)
        feature = self.visual_embedding(sampled, center, draws)
        if return_global:
            feature = self.dgcnn_2(feature, center, idx)
        return feature

    def forward(self, inp, draws, temperature=1.0, hard=False):
        neighborhood, center = self.group_divider(inp)
        idx = knn_graph_ref(center, 4)
        logits = self.dgcnn_1(self.encoder(neighborhood), center, idx)
        sampled = self._gumbel(logits, temperature, hard, draws)
        sampled = self.visual_embedding(sampled, center, draws)
        feature = self.dgcnn_2(sampled, center, idx)
        coarse, fine = self.decoder(feature)
        with torch.no_grad():
            whole_fine = (fine + center.unsqueeze(2)).reshape(inp.size(0), -1, 3)
            whole_coarse = (coarse + center.unsqueeze(2)).reshape(inp.size(0), -1, 3)
        return whole_coarse, whole_fine, coarse, fine, neighborhood, logits

    def get_loss(self, ret, gt=None):
        _, _, coarse, fine, group_gt, logits = ret
        bs, g = coarse.shape[:2]
        coarse = coarse.reshape(bs * g, -1, 3).contiguous()
        fine = fine.reshape(bs * g, -1, 3).contiguous()
        group_gt = group_gt.reshape(bs * g, -1, 3).contiguous()
        loss_recon = chamfer_l1(coarse, group_gt) + chamfer_l1(fine, group_gt)
        mean_softmax = F.softmax(logits, dim=-1).mean(dim=1)
        log_qy = torch.log(mean_softmax)
        log_uniform = torch.log(torch.tensor([1.0 / self.num_tokens]))
        loss_klv = F.kl_div(log_qy, log_uniform.expand(log_qy.size(0), log_qy.size(1)), None, None, "batchmean",
                            log_target=True)
        return loss_recon, loss_klv


class ACTPromptedDiscreteVAEwithVIT(DiscreteVAE):
    """models/dvae.py:360-615.  Configurations (:513-534): deep prompts (the ACT recipe), shallow prompts (`use_deep_prompt` false), no prompts
    (`num_prompt_token` 0; with a frozen Transformer its output is then computed under no_grad, :522-524), no Transformer (`visual_embed_dim`
    'none')."""

    def __init__(self, config):
        super().__init__(config)
        c = config
        self.visual_embed_dim, self.num_prompt_token = c.visual_embed_dim, c.num_prompt_token
        self.use_deep_prompt = bool(c.get("use_deep_prompt", True))
        self.freeze_visual_embed = bool(c.get("freeze_visual_embed", True))
        self.prompt_p = 0.1
        if self.visual_embed_dim == "none":
            self.visual_embed = None
            return
        self.visual_embed_depth = int(c.get("visual_embed_depth", 12))
        vit_heads = int(c.get("visual_embed_heads", 12))
        D = self.visual_embed_dim
        # visual_embed = Sequential(blocks, norm) of a timm ViT (models/dvae.py:405-410): LN eps 1e-6, qkv bias
        vit = BlockList(D, self.visual_embed_depth, vit_heads, 0.0, qkv_bias=True, eps=1e-6, tag="vit")
        self.visual_embed = nn.Sequential(vit.blocks, nn.LayerNorm(D, eps=1e-6))
        self.proj_pre = nn.Linear(self.tokens_dims, D)
        self.visual_pos_embed = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, D))
        self.proj_post = nn.Linear(D, self.tokens_dims)
        Pn = self.num_prompt_token
        if Pn > 0:
            self.visual_prompt_token = nn.Parameter(torch.zeros(1, Pn, D))
            self.visual_prompt_pos = nn.Parameter(torch.randn(1, Pn, D))
            trunc_normal_(self.visual_prompt_token); trunc_normal_(self.visual_prompt_pos)
            if self.use_deep_prompt:
                self.deep_prompt_tokens = nn.Parameter(torch.zeros(self.visual_embed_depth - 1, Pn, D))
                self.deep_prompt_pos = nn.Parameter(torch.randn(self.visual_embed_depth - 1, Pn, D))
                trunc_normal_(self.deep_prompt_tokens); trunc_normal_(self.deep_prompt_pos)
        else:
            self.visual_prompt_token = None
        if self.freeze_visual_embed:
            for p in self.visual_embed.parameters():
                p.requires_grad = False

    # -- pieces -----------------------------------------------------------------------------
    def _prompt_dropout(self, t, draws, key):
        if not self.training or self.prompt_p == 0:
            return t
        keep = draws.get(key, lambda: (torch.rand_like(t) >= self.prompt_p).to(t.dtype))
        return t * keep / (1.0 - self.prompt_p)

    def _plain_stack(self, h, pos, draws):
        """forward_visual_feature (models/dvae.py:500-511): `x = blk(x + pos)` for every block, then the final norm."""
        for blk in self.visual_embed[0]:
            h = blk(h + pos, draws)
        return self.visual_embed[1](h)

    def visual_embedding(self, x, center, draws):
        if self.visual_embed is None:
            return x
        B, Pn = x.shape[0], self.num_prompt_token
        pos = self.visual_pos_embed(center)
        f = self.proj_pre(x)
        if not self.use_deep_prompt:                                   # models/dvae.py:517-534
            if self.visual_prompt_token is None:
                if self.freeze_visual_embed:
                    with torch.no_grad():
                        h = self._plain_stack(f, pos, draws)
                else:
                    h = self._plain_stack(f, pos, draws)
            else:                                                      # incorporate_prompt (:485-498) once, prompt outputs flow through
                prm = self._prompt_dropout(self.visual_prompt_token.expand(B, -1, -1), draws, "prompt.0")
                h = self._plain_stack(torch.cat((prm, f), dim=1), torch.cat((self.visual_prompt_pos.expand(B, -1, -1), pos), dim=1), draws)[:, Pn:]
            return self.proj_post(h)
        # visual_embedding_deep_prompt (models/dvae.py:536-576) + incorporate_prompt (:485-498)
        prm = self._prompt_dropout(self.visual_prompt_token.expand(B, -1, -1), draws, "prompt.0")
        h = torch.cat((prm, f), dim=1)
        pos = torch.cat((self.visual_prompt_pos.expand(B, -1, -1), pos), dim=1)
        blocks = self.visual_embed[0]
        for i in range(self.visual_embed_depth):
            if i > 0:
                prm = self._prompt_dropout(self.deep_prompt_tokens[i - 1].expand(B, -1, -1), draws, f"prompt.{i}")
                h = torch.cat((prm, h[:, Pn:]), dim=1)
                pos = torch.cat((self.deep_prompt_pos[i - 1].expand(B, -1, -1), pos[:, Pn:]), dim=1)
            h = blocks[i](h + pos, draws)
        h = self.visual_embed[1](h)[:, Pn:]
        return self.proj_post(h)


class ACT_PointDistillation(nn.Module):
    def __init__(self, config):
        super().__init__()
        tc = config.transformer_config
        self.mask_ratio, self.embed_dim = tc.mask_ratio, tc.embed_dim
        self.loss_type = config.get("loss", "cosine") if hasattr(config, "get") else "cosine"
        self.cls_loss = bool(tc.get("cls_loss", False)) if hasattr(tc, "get") else False
        self.register_shallow_hook = int(tc.get("register_shallow_hook", -1)) if hasattr(tc, "get") else -1
        self.ACT_encoder = VisableOnlyMaskTransformer(config)
        if self.cls_loss:                          # models/act.py:1120-1122: the model's own position of the cls token for the shallow pass
            self.cls_pos = nn.Parameter(torch.randn(1, 1, self.embed_dim))
            trunc_normal_(self.cls_pos)
        self.dvae_tokenizer = ACTPromptedDiscreteVAEwithVIT(config.dvae_config)
        for p in self.dvae_tokenizer.parameters():
            p.requires_grad = False
        self.group_divider = Group(config.dvae_config.num_group, config.dvae_config.group_size)
        self.proj_head = nn.Linear(self.embed_dim, config.dvae_config.tokens_dims)
        if self.mask_ratio > 0:                    # models/act.py:1158-1178: no decoder at all when nothing is masked
            self.mask_token = nn.Parameter(torch.zeros(1, 1, self.embed_dim))
            self.decoder_pos_embed = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, self.embed_dim))
            dpr = [x.item() for x in torch.linspace(0, tc.drop_path_rate, tc.decoder_depth)]
            self.ACT_decoder = TransformerDecoder(self.embed_dim, tc.decoder_depth, tc.decoder_num_heads, dpr)
            for m in self.ACT_decoder.modules():
                if isinstance(m, nn.Linear):
                    nn.init.xavier_uniform_(m.weight)
                    if m.bias is not None:
                        nn.init.constant_(m.bias, 0)
            trunc_normal_(self.mask_token)

    def _loss(self, student, teacher, num_mask=1):  # models/act.py:1186-1195,1243-1256
        if self.loss_type in ("ntxent", "barlow"):
            return pairwise_distill_loss(student, teacher, self.loss_type, num_mask)
        if self.loss_type == "l2":
            return F.mse_loss(student, teacher)
        if self.loss_type == "smoothl1":
            return F.smooth_l1_loss(student, teacher)
        return cosine_distill_loss(student, teacher)

    def forward(self, pts, draws=None, noaug=False):
        draws = draws if draws is not None else Draws()
        neighborhood, center = self.group_divider(pts)
        if noaug:
            with torch.no_grad():
                return self.ACT_encoder(neighborhood, center, draws, only_cls_tokens=True, noaug=True)
        if self.cls_loss:                          # models/act.py:1208-1213
            x_vis, x_cls, x_shallow, mask = self.ACT_encoder(neighborhood, center, draws, register_shallow_hook=self.register_shallow_hook)
        else:
            x_vis, mask = self.ACT_encoder(neighborhood, center, draws)
        B, _, C = x_vis.shape
        with torch.no_grad():
            teacher = self.dvae_tokenizer.forward_tokenizer_features(neighborhood, center, draws)
        if self.mask_ratio == 0:                    # models/act.py:1238-1240: no decoder, every token regressed
            return self._loss(self.proj_head(x_vis), teacher)
        pos_vis = self.decoder_pos_embed(center[~mask]).reshape(B, -1, C)
        pos_msk = self.decoder_pos_embed(center[mask]).reshape(B, -1, C)
        num_mask = pos_msk.shape[1]
        x_full = torch.cat([x_vis, self.mask_token.expand(B, num_mask, -1)], dim=1)
        pos_full = torch.cat([pos_vis, pos_msk], dim=1)
        student = self.proj_head(self.ACT_decoder(x_full, pos_full, num_mask, draws))
        teacher = teacher[mask].reshape(B, -1, student.shape[-1])
        loss = self._loss(student, teacher, num_mask)
        if self.cls_loss and self.loss_type in ("ntxent", "barlow"):                # models/act.py:1252-1253: the global term, same loss
            x_sh = torch.cat([x_cls.unsqueeze(1), x_shallow, self.mask_token.expand(B, num_mask, -1)], dim=1)
            pos_sh = torch.cat([self.cls_pos.expand(B, -1, -1), pos_full], dim=1)
            student_g = self.proj_head(self.ACT_decoder(x_sh, pos_sh, num_mask, draws, tag="dec_shallow"))
            loss = loss + self._loss(student_g, teacher, num_mask)
        if self.cls_loss and self.loss_type == "cosine":                          # second decoder pass on [cls, shallow visible tokens, mask tokens] (models/act.py:1231-1236,1248-1249)
            x_sh = torch.cat([x_cls.unsqueeze(1), x_shallow, self.mask_token.expand(B, num_mask, -1)], dim=1)
            pos_sh = torch.cat([self.cls_pos.expand(B, -1, -1), pos_full], dim=1)
            student_g = self.proj_head(self.ACT_decoder(x_sh, pos_sh, num_mask, draws, tag="dec_shallow"))
            loss = loss + cosine_distill_loss(student_g, teacher)
        return loss


class PointTransformer(nn.Module):
    """Finetune / inference classifier (models/act.py:727-910): Group -> mini-PointNet tokens -> [cls; tokens] through the
    encoder blocks -> LN -> cat(cls, max over tokens) -> head.  Dropout masks of the mlp-3 head come from ``draws``
    (keys ``head.drop1`` / ``head.drop2``: keep masks, scaled by 1/(1-p))."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.embed_dim, self.depth, self.cls_dim = config.embed_dim, config.depth, config.cls_dim
        self.num_heads, self.encoder_dims = config.num_heads, config.encoder_dims
        self.group_divider = Group(config.num_group, config.group_size)
        self.encoder = Encoder(self.encoder_dims)
        self.reduce_dim = nn.Linear(self.encoder_dims, self.embed_dim) if self.encoder_dims != self.embed_dim \
            else nn.Identity()
        self.cls_token = nn.Parameter(torch.zeros(1, 1, self.embed_dim))
        self.cls_pos = nn.Parameter(torch.randn(1, 1, self.embed_dim))
        self.pos_embed = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, self.embed_dim))
        dpr = [x.item() for x in torch.linspace(0, config.drop_path_rate, self.depth)]
        self.blocks = TransformerEncoder(self.embed_dim, self.depth, self.num_heads, dpr, tag="enc")
        self.norm = nn.LayerNorm(self.embed_dim)
        if config.transfer_type == 'linear':
            self.cls_head_finetune = nn.Sequential(nn.Linear(self.embed_dim * 2, self.cls_dim))
        else:
            self.cls_head_finetune = nn.Sequential(
                nn.Linear(self.embed_dim * 2, 256), nn.BatchNorm1d(256), nn.ReLU(inplace=True), nn.Dropout(0.5),
                nn.Linear(256, 256), nn.BatchNorm1d(256), nn.ReLU(inplace=True), nn.Dropout(0.5),
                nn.Linear(256, self.cls_dim))
        if config.transfer_type == 'side':
            self.side_alpha = nn.Parameter(torch.Tensor([0.0]))
            self.side = Encoder(self.embed_dim)
            self.side_projection = nn.Linear(self.embed_dim, self.embed_dim, bias=False)
        else:
            self.side = None
        trunc_normal_(self.cls_token); trunc_normal_(self.cls_pos)
        t = config.transfer_type                       # parameter-efficient transfer: freeze by name (models/act.py:797-808)
        if t != 'full':
            for name, p in self.named_parameters():
                if t in ('mlp-3', 'linear'):
                    keep = 'cls' in name
                elif t == 'side':
                    keep = 'side' in name or 'cls' in name
                elif t == 'bit-fit':
                    keep = 'bias' in name or 'cls' in name
                else:
                    keep = True
                if not keep:
                    p.requires_grad = False

    def _head(self, f, draws):
        h = self.cls_head_finetune
        if len(h) == 1:
            return h[0](f)
        x = f
        for i, m in enumerate(h):
            if isinstance(m, nn.Dropout):
                if self.training and m.p > 0:
                    key = "head.drop1" if i == 3 else "head.drop2"
                    keep = draws.get(key, lambda: (torch.rand_like(x) >= m.p).to(x.dtype))
                    x = x * keep / (1.0 - m.p)
            else:
                x = m(x)
        return x

    def forward(self, pts, draws=None):
        draws = draws if draws is not None else Draws()
        neighborhood, center = self.group_divider(pts)
        tok = self.reduce_dim(self.encoder(neighborhood))
        B = tok.shape[0]
        if self.side is not None:
            side = self.side_projection(self.side(neighborhood))
        x = torch.cat((self.cls_token.expand(B, -1, -1), tok), dim=1)
        pos = torch.cat((self.cls_pos.expand(B, -1, -1), self.pos_embed(center)), dim=1)
        x = self.norm(self.blocks(x, pos, draws))
        if self.side is not None:
            a = torch.sigmoid(self.side_alpha)
            side = a * x[:, 1:] + (1 - a) * side
            f = torch.cat([x[:, 0], side.max(1)[0]], dim=-1)
        else:
            f = torch.cat([x[:, 0], x[:, 1:].max(1)[0]], dim=-1)
        return self._head(f, draws)

    def get_loss_acc(self, ret, gt):
        loss = F.cross_entropy(ret, gt.long())
        acc = (ret.argmax(-1) == gt).sum() / float(gt.size(0))
        return loss, acc * 100


def param_groups(model, weight_decay):
    """tools/builder.py:38-51 add_weight_decay."""
    decay, no_decay = [], []
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if len(p.shape) == 1 or name.endswith(".bias") or "token" in name:
            no_decay.append(p)
        else:
            decay.append(p)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


# ---- ACT_PointBERT: the Point-BERT style alternative recipe (models/act.py:532-725 MaskTransformer, :913-1096 ACT_PointBERT) -----------
class MaskTransformer(nn.Module):
    """models/act.py:532-725.  Random draws by key (``tag`` distinguishes the three passes of one ACT_PointBERT.forward):
    ``<tag>.ratio`` (python float in [lo, hi)), ``<tag>.mask_u`` [B,G] uniforms (mask = u < ratio), ``<tag>.replace_u`` [B,G] uniforms,
    ``<tag>.perm`` permutation of B*G; block masking: ``<tag>.seed`` [B] indices + ``<tag>.ratios`` [B]."""

    def __init__(self, config):
        super().__init__()
        tc = config.transformer_config
        self.mask_ratio, self.mask_type = tc.mask_ratio, tc.mask_type
        self.embed_dim, self.depth, self.cls_dim = tc.embed_dim, tc.depth, tc.cls_dim
        self.replace_pob, self.num_heads, self.encoder_dims = tc.replace_pob, tc.num_heads, tc.encoder_dims
        self.encoder = Encoder(self.encoder_dims)
        self.reduce_dim = nn.Linear(self.encoder_dims, self.embed_dim)
        self.cls_token = nn.Parameter(torch.randn(1, 1, self.embed_dim))
        self.mask_token = nn.Parameter(torch.randn(1, 1, self.embed_dim))
        self.cls_pos = nn.Parameter(torch.randn(1, 1, self.embed_dim))
        self.pos_embed = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, self.embed_dim))
        dpr = [x.item() for x in torch.linspace(0, tc.drop_path_rate, self.depth)]
        self.blocks = TransformerEncoder(self.embed_dim, self.depth, self.num_heads, dpr, tag="bert")
        self.norm = nn.LayerNorm(self.embed_dim)
        self.num_tokens = config.dvae_config.num_tokens
        self.lm_head = nn.Linear(self.embed_dim, self.num_tokens)
        self.cls_head = nn.Sequential(nn.Linear(self.embed_dim, self.cls_dim), nn.GELU(), nn.Linear(self.cls_dim, self.cls_dim))
        for t in (self.cls_token, self.cls_pos, self.mask_token):
            trunc_normal_(t)
        self.apply(VisableOnlyMaskTransformer._init_weights)

    def _mask(self, center, noaug, draws, tag):
        B, G, _ = center.shape
        if noaug or self.mask_ratio[1] == 0:
            return torch.zeros(B, G, dtype=torch.bool)
        lo, hi = self.mask_ratio
        if self.mask_type == 'rand':                # :648-659
            import random
            ratio = draws.get(f"{tag}.ratio", lambda: random.random() * (hi - lo) + lo)
            return draws.get(f"{tag}.mask_u", lambda: torch.rand(B, G)) < float(ratio)
        seed = draws.get(f"{tag}.seed", lambda: torch.randint(0, G, (B,)))                  # :620-646
        ratios = draws.get(f"{tag}.ratios", lambda: lo + (hi - lo) * torch.rand(B))
        out = torch.zeros(B, G, dtype=torch.bool)
        for b in range(B):
            d = torch.norm(center[b, int(seed[b])].reshape(1, 3) - center[b], p=2, dim=-1)
            idx = torch.argsort(d, dim=-1, descending=False)
            out[b, idx[:int(float(ratios[b]) * G)]] = True
        return out

    def _random_replace(self, tok, mask, noaug, draws, tag):                             # :661-689
        if noaug or self.replace_pob == 0:
            return tok, mask
        B, G, C = tok.shape
        rep = (draws.get(f"{tag}.replace_u", lambda: torch.rand(B, G)) < self.replace_pob) & ~mask
        overall = rep | mask
        perm = draws.get(f"{tag}.perm", lambda: torch.randperm(B * G))
        shuffled = tok.detach().reshape(B * G, C)[perm].reshape(B, G, C)
        w = rep.unsqueeze(-1).to(tok.dtype)
        return tok * (1 - w) + shuffled * w, overall

    def forward(self, neighborhood, center, draws, return_all_tokens=False, only_cls_tokens=False, noaug=False, tag="q"):
        mask = self._mask(center, noaug, draws, tag)
        tok = self.reduce_dim(self.encoder(neighborhood))
        tok, overall = self._random_replace(tok, mask.clone(), noaug, draws, tag)
        B, G, _ = tok.shape
        w = mask.unsqueeze(-1).to(tok.dtype)
        tok = tok * (1 - w) + self.mask_token.expand(B, G, -1) * w
        x = torch.cat((self.cls_token.expand(B, -1, -1), tok), dim=1)
        pos = torch.cat((self.cls_pos.expand(B, -1, -1), self.pos_embed(center)), dim=1)
        x = self.norm(self.blocks(x, pos, draws))
        if only_cls_tokens:
            return self.cls_head(x[:, 0])
        logits = self.lm_head(x[:, 1:])
        if return_all_tokens:
            return self.cls_head(x[:, 0]), logits
        return self.cls_head(x[:, 0]), logits[~overall], logits[overall], overall


class ACT_PointBERT(nn.Module):
    """models/act.py:913-1096: MoCo-style query / momentum-key MaskTransformers, dVAE token prediction, cut-mix contrast.
    forward -> (moco_loss, dvae_loss, cutmix_loss).  Draw keys: ``mixup_ratio`` [B], ``mixup_u`` [B,G], plus the MaskTransformer keys
    under tags ``q`` / ``mix`` / ``k``."""

    def __init__(self, config):
        super().__init__()
        self.m, self.T, self.K = config.m, config.T, config.K
        tc = config.transformer_config
        self.moco_loss, self.dvae_loss, self.cutmix_loss = tc.moco_loss, tc.dvae_loss, tc.cutmix_loss
        self.return_all_tokens = tc.return_all_tokens
        self.transformer_q = MaskTransformer(config)
        self.transformer_k = MaskTransformer(config)
        for pq, pk in zip(self.transformer_q.parameters(), self.transformer_k.parameters()):
            pk.data.copy_(pq.data); pk.requires_grad = False
        self.dvae = ACTPromptedDiscreteVAEwithVIT(config.dvae_config)
        for p in self.dvae.parameters():
            p.requires_grad = False
        self.group_divider = Group(config.dvae_config.num_group, config.dvae_config.group_size)
        self.register_buffer("queue", F.normalize(torch.randn(tc.cls_dim, self.K), dim=0))
        self.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))

    def forward_tokenizer(self, neighborhood, center):
        d = self.dvae
        return d.dgcnn_1(d.encoder(neighborhood), center, knn_graph_ref(center, 4)).argmax(-1).long()   # models/dvae.py:578-582

    def forward(self, pts, draws=None, noaug=False):
        draws = draws if draws is not None else Draws()
        neighborhood, center = self.group_divider(pts)
        if noaug:
            with torch.no_grad():
                return self.transformer_q(neighborhood, center, draws, only_cls_tokens=True, noaug=True)
        B, G = center.shape[:2]
        with torch.no_grad():
            label = self.forward_tokenizer(neighborhood, center)
        rat = self.return_all_tokens
        q_out = self.transformer_q(neighborhood, center, draws, return_all_tokens=rat, tag="q")
        q_cls = F.normalize(q_out[0], dim=1)
        ratio = draws.get("mixup_ratio", lambda: torch.rand(B))                                   # _mixup_pc :1015-1032
        mm = (draws.get("mixup_u", lambda: torch.rand(B, G)) < ratio.unsqueeze(-1)).to(neighborhood.dtype)
        mix_nb = neighborhood * mm[..., None, None] + neighborhood.flip(0) * (1 - mm[..., None, None])
        mix_c = center * mm.unsqueeze(-1) + center.flip(0) * (1 - mm.unsqueeze(-1))
        mix_label = (label * mm + label.flip(0) * (1 - mm)).long()
        m_out = self.transformer_q(mix_nb, mix_c, draws, return_all_tokens=rat, tag="mix")
        m_cls = F.normalize(m_out[0], dim=1)
        with torch.no_grad():
            for pq, pk in zip(self.transformer_q.parameters(), self.transformer_k.parameters()):
                pk.data = pk.data * self.m + pq.data * (1. - self.m)
            k_cls = F.normalize(self.transformer_k(neighborhood, center, draws, only_cls_tokens=True, tag="k"), dim=1)
        zero = torch.tensor(0.)
        queue = self.queue.clone().detach()
        moco = zero
        if self.moco_loss:
            lg = torch.cat([(q_cls * k_cls).sum(1, keepdim=True), q_cls @ queue], dim=1) / self.T
            moco = F.cross_entropy(lg, torch.zeros(B, dtype=torch.long))
        dv = zero
        if self.dvae_loss:
            if rat:
                dv = F.cross_entropy(q_out[1].reshape(-1, q_out[1].size(-1)), label.reshape(-1)) + \
                    F.cross_entropy(m_out[1].reshape(-1, m_out[1].size(-1)), mix_label.reshape(-1))
            else:
                dv = F.cross_entropy(q_out[2], label[q_out[3]]) + F.cross_entropy(m_out[2], mix_label[m_out[3]])
        cm = zero
        if self.cutmix_loss:
            lg = torch.cat([m_cls @ k_cls.t(), m_cls @ queue], dim=1) / self.T
            lab = torch.arange(B, dtype=torch.long)
            cm = (ratio * F.cross_entropy(lg, lab, reduction='none') + (1 - ratio) * F.cross_entropy(lg, lab.flip(0), reduction='none')).mean()
        with torch.no_grad():                                                                      # _dequeue_and_enqueue :991-1005
            ptr = int(self.queue_ptr)
            assert self.K % B == 0
            self.queue[:, ptr:ptr + B] = k_cls.T
            self.queue_ptr[0] = (ptr + B) % self.K
        return moco, dv, cm
