"""ACT_PointBERT: the Point-BERT style alternative pretraining recipe of the reference (models/act.py:532-725 ``MaskTransformer``,
:913-1096 ``ACT_PointBERT``): a query MaskTransformer and its momentum (key) copy, the frozen dVAE as token labeller, three losses
(MoCo contrast against a feature queue, masked dVAE-token prediction, cut-mix contrast).  Same class names, constructor keys,
``state_dict`` keys and ``forward`` contract -- a 3-tuple of losses, which ``tools/runner_pretrain.py:140-142`` sums.

The heavy parts run on the HIP kernels of the hot path (Group, mini-PointNet, Transformer blocks, LayerNorm, Linear, cross-entropy).
The recipe-specific glue on [B, 1+K] contrast logits and the boolean token selection of ``return_all_tokens: False`` are small torch
ops, as in the reference (that selection has a data-dependent size, i.e. one host synchronisation per step -- the reference's behaviour).
The reference ships no YAML for this model; the keys are the ones its constructor reads."""
import random

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import kernels as K
from ..utils.logger import print_log
from .act import TransformerEncoder, trunc_normal_
from .build import MODELS
from .dvae import ACTPromptedDiscreteVAEwithVIT, Encoder, Group


def _draw(draws, key, make):
    return draws.get(key, make) if draws is not None else make()


class MaskTransformer(nn.Module):
    """models/act.py:532-725.  ``draws`` keys (parity tests; ``tag`` names the pass): ``<tag>.ratio``, ``<tag>.mask_u`` [B,G],
    ``<tag>.replace_u`` [B,G], ``<tag>.perm`` [B*G]; block masking: ``<tag>.seed`` [B], ``<tag>.ratios`` [B]."""

    def __init__(self, config, **kwargs):
        super().__init__()
        self.config = config
        tc = config.transformer_config
        self.mask_ratio, self.mask_type = tc.mask_ratio, tc.mask_type
        self.embed_dim, self.depth, self.drop_path_rate = tc.embed_dim, tc.depth, tc.drop_path_rate
        self.cls_dim, self.replace_pob, self.num_heads = tc.cls_dim, tc.replace_pob, tc.num_heads
        print_log(f'[Transformer args] {tc}', logger='dVAE BERT')
        self.encoder_dims = tc.encoder_dims
        self.encoder = Encoder(encoder_channel=self.encoder_dims)
        self.reduce_dim = nn.Linear(self.encoder_dims, self.embed_dim)
        self.cls_token = nn.Parameter(torch.randn(1, 1, self.embed_dim))
        self.mask_token = nn.Parameter(torch.randn(1, 1, self.embed_dim))
        self.cls_pos = nn.Parameter(torch.randn(1, 1, self.embed_dim))
        self.pos_embed = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, self.embed_dim))
        dpr = [x.item() for x in torch.linspace(0, self.drop_path_rate, self.depth)]
        self.blocks = TransformerEncoder(embed_dim=self.embed_dim, depth=self.depth, drop_path_rate=dpr, num_heads=self.num_heads)
        self.norm = nn.LayerNorm(self.embed_dim)
        self.num_tokens = config.dvae_config.num_tokens
        self.lm_head = nn.Linear(self.embed_dim, self.num_tokens)
        self.cls_head = nn.Sequential(nn.Linear(self.embed_dim, self.cls_dim), nn.GELU(), nn.Linear(self.cls_dim, self.cls_dim))
        for t in (self.cls_token, self.cls_pos, self.mask_token):
            trunc_normal_(t, std=.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, (nn.Linear, nn.Conv1d)):
            trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def _prepare_encoder(self, dvae_ckpt):
        """the mini-PointNet starts from the Stage-I tokenizer's (models/act.py:603-609)"""
        ckpt = torch.load(dvae_ckpt, map_location='cpu')
        base = {k.replace("module.", ""): v for k, v in ckpt['base_model'].items()}
        self.encoder.load_state_dict({k.replace("encoder.", ""): v for k, v in base.items() if 'encoder' in k}, strict=True)
        print_log(f'[Encoder] Successful Loading the ckpt for encoder from {dvae_ckpt}', logger='dVAE BERT')

    def _mask(self, center, noaug, draws, tag):
        B, G, _ = center.shape
        dev = center.device
        if noaug or self.mask_ratio[1] == 0:
            return torch.zeros(B, G, dtype=torch.bool, device=dev)
        lo, hi = self.mask_ratio
        if self.mask_type == 'rand':                                     # :648-659: one ratio per batch, Bernoulli mask
            ratio = _draw(draws, f"{tag}.ratio", lambda: random.random() * (hi - lo) + lo)
            u = _draw(draws, f"{tag}.mask_u", lambda: torch.rand(B, G, device=dev))
            return u.to(dev) < float(ratio)
        # 'block' (:611-646): the int(ratio_b * G) centres nearest to a random seed centre, per cloud; on the device, no Python loop
        seed = _draw(draws, f"{tag}.seed", lambda: torch.randint(0, G, (B,), device=dev)).to(dev).long()
        ratios = _draw(draws, f"{tag}.ratios", lambda: lo + (hi - lo) * torch.rand(B, device=dev)).to(dev)
        ref = center[torch.arange(B, device=dev), seed]
        rank = torch.argsort(torch.argsort(torch.linalg.vector_norm(ref.unsqueeze(1) - center, dim=-1), dim=-1, stable=True), dim=-1)
        return rank < (ratios.double() * G).long().unsqueeze(1)

    def _random_replace(self, tok, mask, noaug, draws, tag):
        """:661-689: a fraction ``replace_pob`` of the unmasked tokens is replaced by tokens of random other groups of the batch"""
        if noaug or self.replace_pob == 0:
            return tok, mask
        B, G, C = tok.shape
        dev = tok.device
        rep = (_draw(draws, f"{tag}.replace_u", lambda: torch.rand(B, G, device=dev)).to(dev) < self.replace_pob) & ~mask
        perm = _draw(draws, f"{tag}.perm", lambda: torch.randperm(B * G, device=dev)).to(dev)
        shuffled = tok.detach().reshape(B * G, C)[perm].reshape(B, G, C)
        w = rep.unsqueeze(-1).to(tok.dtype)
        return tok * (1 - w) + shuffled * w, rep | mask

    def forward(self, neighborhood, center, return_all_tokens=False, only_cls_tokens=False, noaug=False, draws=None, tag="q"):
        mask = self._mask(center, noaug, draws, tag)                       # B G
        tok = self.encoder(neighborhood)
        tok = K.linear(tok, self.reduce_dim.weight, self.reduce_dim.bias)
        tok, overall = self._random_replace(tok, mask.clone(), noaug, draws, tag)
        B, G, _ = tok.shape
        w = mask.unsqueeze(-1).to(tok.dtype)
        tok = tok * (1 - w) + self.mask_token.expand(B, G, -1) * w
        pe = self.pos_embed
        pos = K.mlp(center, pe[0].weight, pe[0].bias, pe[2].weight, pe[2].bias)
        x = torch.cat((self.cls_token.expand(B, -1, -1), tok), dim=1)
        pos = torch.cat((self.cls_pos.expand(B, -1, -1), pos), dim=1)
        x = self.blocks(x, pos, draws, tag=f"bert.{tag}")
        x = K.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps)
        ch = self.cls_head
        cls = K.mlp(x[:, 0].contiguous(), ch[0].weight, ch[0].bias, ch[2].weight, ch[2].bias)
        if only_cls_tokens:
            return cls
        logits = K.linear(x[:, 1:].contiguous(), self.lm_head.weight, self.lm_head.bias)
        if return_all_tokens:
            return cls, logits
        return cls, logits[~overall], logits[overall], overall           # data-dependent sizes, as in the reference


@MODELS.register_module()
class ACT_PointBERT(nn.Module):
    """models/act.py:913-1096.  forward(pts) -> (moco_loss, dvae_loss, cutmix_loss); forward(pts, noaug=True) -> cls feature."""

    def __init__(self, config):
        super().__init__()
        print_log('[ACT] build dVAE_BERT ...', logger='ACT')
        self.config = config
        self.m, self.T, self.K = config.m, config.T, config.K
        tc = config.transformer_config
        self.moco_loss, self.dvae_loss, self.cutmix_loss = tc.moco_loss, tc.dvae_loss, tc.cutmix_loss
        self.return_all_tokens = tc.return_all_tokens
        ckpt = config.dvae_config.get("ckpt", None)
        has_ckpt = bool(ckpt) and str(ckpt).lower() not in ("none", "random", "")
        self.transformer_q = MaskTransformer(config)
        if has_ckpt:
            self.transformer_q._prepare_encoder(ckpt)
        self.transformer_k = MaskTransformer(config)
        for pq, pk in zip(self.transformer_q.parameters(), self.transformer_k.parameters()):
            pk.data.copy_(pq.data)
            pk.requires_grad = False                                      # momentum copy: never updated by gradient
        self.dvae = ACTPromptedDiscreteVAEwithVIT(config.dvae_config)
        if has_ckpt:
            blob = torch.load(ckpt, map_location='cpu')
            self.dvae.load_state_dict({k.replace("module.", ""): v for k, v in blob['base_model'].items()}, strict=True)
            print_log(f'[dVAE] Successful Loading the ckpt for dvae from {ckpt}', logger='ACT')
        else:
            import warnings
            warnings.warn("ACT_PointBERT: dvae_config.ckpt is 'none' -- the dVAE token labeller is RANDOMLY INITIALISED (tests / benchmarks only)",
                          stacklevel=2)
        for p in self.dvae.parameters():
            p.requires_grad = False
        self.group_size, self.num_group = config.dvae_config.group_size, config.dvae_config.num_group
        self.group_divider = Group(num_group=self.num_group, group_size=self.group_size,
                                   skip_near_origin=config.dvae_config.get("fps_skip_near_origin", None))
        self.register_buffer("queue", F.normalize(torch.randn(self.transformer_q.cls_dim, self.K), dim=0))
        self.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
        self.build_loss_func()

    def build_loss_func(self):
        self.loss_ce = nn.CrossEntropyLoss()                               # kept for interface parity; the token loss runs the HIP kernel
        self.loss_ce_batch = nn.CrossEntropyLoss(reduction='none')

    @torch.no_grad()
    def _momentum_update_key_encoder(self):
        qs = [p.data for p in self.transformer_q.parameters()]
        ks = [p.data for p in self.transformer_k.parameters()]
        torch._foreach_mul_(ks, self.m)                                    # two fused launches instead of two per parameter
        torch._foreach_add_(ks, qs, alpha=1. - self.m)

    @torch.no_grad()
    def _dequeue_and_enqueue(self, keys):
        B = keys.shape[0]
        assert self.K % B == 0
        # the pointer lives on the device: advance it there (the reference's int(self.queue_ptr) is a host sync per step)
        idx = (self.queue_ptr + torch.arange(B, device=keys.device)) % self.K
        self.queue.index_copy_(1, idx, keys.T.contiguous())
        self.queue_ptr.add_(B).remainder_(self.K)

    def forward_eval(self, pts):
        with torch.no_grad():
            neighborhood, center = self.group_divider(pts)
            return self.transformer_q(neighborhood, center, only_cls_tokens=True, noaug=True)

    def _mixup_pc(self, neighborhood, center, dvae_label, draws):
        """:1007-1032: per cloud a ratio; groups are kept with that probability, otherwise taken from the batch-flipped cloud"""
        B, G = center.shape[:2]
        dev = center.device
        ratio = _draw(draws, "mixup_ratio", lambda: torch.rand(B, device=dev)).to(dev)
        mm = (_draw(draws, "mixup_u", lambda: torch.rand(B, G, device=dev)).to(dev) < ratio.unsqueeze(-1)).to(neighborhood.dtype)
        nb = neighborhood * mm[..., None, None] + neighborhood.flip(0) * (1 - mm[..., None, None])
        c = center * mm.unsqueeze(-1) + center.flip(0) * (1 - mm.unsqueeze(-1))
        lab = (dvae_label * mm + dvae_label.flip(0) * (1 - mm)).long()
        return ratio, nb.contiguous(), c.contiguous(), lab

    def forward(self, pts, noaug=False, draws=None, **kwargs):
        if noaug:
            return self.forward_eval(pts)
        neighborhood, center = self.group_divider(pts)
        B = center.shape[0]
        dev = pts.device
        with torch.no_grad():
            dvae_label = self.dvae.forward_tokenizer(neighborhood, center)
        rat = self.return_all_tokens
        q_out = self.transformer_q(neighborhood, center, return_all_tokens=rat, draws=draws, tag="q")
        q_cls = F.normalize(q_out[0], dim=1)
        ratio, mix_nb, mix_c, mix_label = self._mixup_pc(neighborhood, center, dvae_label, draws)
        m_out = self.transformer_q(mix_nb, mix_c, return_all_tokens=rat, draws=draws, tag="mix")
        m_cls = F.normalize(m_out[0], dim=1)
        with torch.no_grad():
            self._momentum_update_key_encoder()
            k_cls = F.normalize(self.transformer_k(neighborhood, center, only_cls_tokens=True, draws=draws, tag="k"), dim=1)
        queue = self.queue.clone().detach()
        zero = torch.zeros((), device=dev)
        moco = zero
        if self.moco_loss:
            lg = torch.cat([(q_cls * k_cls).sum(1, keepdim=True), q_cls @ queue], dim=1) / self.T
            moco, _ = K.softmax_xent(lg.contiguous(), torch.zeros(B, dtype=torch.long, device=dev))
        dv = zero
        if self.dvae_loss:
            if rat:
                dv = K.softmax_xent(q_out[1].reshape(-1, q_out[1].size(-1)), dvae_label.reshape(-1))[0] + \
                    K.softmax_xent(m_out[1].reshape(-1, m_out[1].size(-1)), mix_label.reshape(-1))[0]
            else:
                dv = K.softmax_xent(q_out[2], dvae_label[q_out[3]])[0] + K.softmax_xent(m_out[2], mix_label[m_out[3]])[0]
        cm = zero
        if self.cutmix_loss:
            lg = torch.cat([m_cls @ k_cls.t(), m_cls @ queue], dim=1) / self.T
            lab = torch.arange(B, dtype=torch.long, device=dev)
            cm = (ratio * self.loss_ce_batch(lg, lab) + (1 - ratio) * self.loss_ce_batch(lg, lab.flip(0))).mean()
        self._dequeue_and_enqueue(k_cls)
        return moco, dv, cm
