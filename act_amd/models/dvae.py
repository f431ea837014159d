"""Point-patch tokenizer and the frozen cross-modal teacher (reference: models/dvae.py).

Same class names, constructor arguments, forward signatures and ``state_dict`` keys as the reference
(``Group``, ``Encoder``, ``DGCNN``, ``Decoder``, ``ACTPromptedDiscreteVAEwithVIT``); the arithmetic runs in
the HIP kernels of libact_hip.so via act_amd.kernels / act_amd.knn_cuda / act_amd.pointnet2_ops.
GPU only: a CPU tensor raises (there is no fallback path).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import kernels as K
from .. import composite as CP
from ..knn_cuda import KNN, knn_group
from ..pointnet2_ops import pointnet2_utils
from ..extensions.chamfer_dist import ChamferDistanceL1, ChamferDistanceL2
from ..utils.draws import Draws
from .build import MODELS

knn = KNN(k=4, transpose_mode=False)          # module-level DGCNN graph operator, as in models/dvae.py:23


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


class Group(nn.Module):
    """FPS centres -> kNN neighbourhoods -> centred patches (models/dvae.py:154-183), two launches."""

    def __init__(self, num_group, group_size, skip_near_origin=None):
        super().__init__()
        self.num_group = num_group
        self.group_size = group_size
        # None: process default (ACT_FPS_SKIP_NEAR_ORIGIN); True reproduces upstream pointnet2_ops, which never selects a point with
        # |p|^2 <= 1e-3 (config key ``fps_skip_near_origin`` of the model / dvae_config sections, absent from the reference YAML)
        self.skip_near_origin = skip_near_origin
        self.knn = KNN(k=self.group_size, transpose_mode=True)

    def forward(self, xyz):
        """xyz [B,N,3] -> neighborhood [B,G,M,3] (centred), center [B,G,3]"""
        xyz = xyz.contiguous()
        with torch.no_grad():
            _, center = pointnet2_utils.furthest_point_sample_with_centers(xyz, self.num_group, self.skip_near_origin)
            _, neighborhood, _ = knn_group(xyz, center, self.group_size, want_nbr=True)
        return neighborhood, center


def _w2d(conv):
    """[out, in, 1(,1)] conv weight viewed as the [out, in] matrix of the equivalent Linear."""
    w = conv.weight
    return w.view(w.shape[0], w.shape[1])


class Encoder(nn.Module):
    """mini-PointNet patch embedding (models/dvae.py:185-215) on rows [B*G*n, C].

    second_conv[0] acts on cat(global.expand, local): W[:, :256] . global is identical for the n points of a
    group, so it is computed once per group and broadcast (exact in real arithmetic, 25% fewer FLOPs)."""

    def __init__(self, encoder_channel):
        super().__init__()
        self.encoder_channel = encoder_channel
        self.first_conv = nn.Sequential(nn.Conv1d(3, 128, 1), nn.BatchNorm1d(128), nn.ReLU(inplace=True),
                                        nn.Conv1d(128, 256, 1))
        self.second_conv = nn.Sequential(nn.Conv1d(512, 512, 1), nn.BatchNorm1d(512), nn.ReLU(inplace=True),
                                         nn.Conv1d(512, self.encoder_channel, 1))

    def forward(self, point_groups, need=None):
        """``need`` [B, k] (optional, no reference counterpart): the only groups per cloud whose tokens the caller reads -- MaskTransformer keeps the
        visible patches (models/act.py:269-275).  The last conv + max-pool then run on those groups alone and the other tokens come back as ZEROS; every
        layer in front still sees all groups (BatchNorm statistics are over all of them), so the wanted tokens and all gradients are unchanged."""
        if CP.ENABLED and not any(isinstance(m, nn.SyncBatchNorm) for m in (self.first_conv[1], self.second_conv[1])):
            return CP.pointnet_forward(self, point_groups, need)        # one host call per direction (csrc/composite.hip)
        bs, g, n, _ = point_groups.shape
        x = point_groups.reshape(bs * g * n, 3)
        c1, bn1, _, c2 = self.first_conv
        c3, bn2, _, c4 = self.second_conv
        h = K.linear(x, _w2d(c1), c1.bias)
        h = K.batch_norm_act(h, bn1, self.training, relu=True)
        h = K.linear(h, _w2d(c2), c2.bias)                               # [R,256]
        fg = K.group_max(h, n)                                           # [BG,256]
        w3 = _w2d(c3)
        gw = K.linear(fg, w3[:, :256], c3.bias)                          # per-group half of the 512->512 conv
        h = K.linear_group_add(h, w3[:, 256:], gw, n)                    # + per-point half, broadcast add in the epilogue
        h = K.batch_norm_act(h, bn2, self.training, relu=True)
        h = K.linear(h, _w2d(c4), c4.bias)
        return K.group_max(h, n).reshape(bs, g, self.encoder_channel)


class DGCNN(nn.Module):
    """4 x (kNN(k=4) graph feature -> 1x1 conv -> GroupNorm(4) -> LeakyReLU(0.2) -> max_k) + 1x1 conv head
    (models/dvae.py:26-117).

    conv(cat(x_j - x_i, x_i)) = Wa x_j + (Wb - Wa) x_i, so each edge-conv layer is ONE GEMM over the B*G points
    (not over the B*G*k edges) followed by a gather: 4x fewer FLOPs than the reference formulation."""

    def __init__(self, encoder_channel, output_channel):
        super().__init__()
        self.input_trans = nn.Conv1d(encoder_channel, 128, 1)

        def layer(cin, cout):
            return nn.Sequential(nn.Conv2d(cin, cout, kernel_size=1, bias=False), nn.GroupNorm(4, cout),
                                 nn.LeakyReLU(negative_slope=0.2))
        self.layer1 = layer(256, 256)
        self.layer2 = layer(512, 512)
        self.layer3 = layer(1024, 512)
        self.layer4 = layer(1024, 1024)
        self.layer5 = nn.Sequential(nn.Conv1d(2304, output_channel, kernel_size=1, bias=False),
                                    nn.GroupNorm(4, output_channel), nn.LeakyReLU(negative_slope=0.2))

    @staticmethod
    def graph_index(coor):
        """coor [B,G,3] -> idx int64 [B,k=4,G] (KNN(k=4, transpose_mode=False) layout)."""
        idx, _, _ = knn_group(coor.contiguous(), coor.contiguous(), 4, idx_kq=True)
        return idx

    @staticmethod
    def _stacked_weight(conv):
        w = _w2d(conv)
        cin = w.shape[1] // 2
        wa, wb = w[:, :cin], w[:, cin:]
        if torch.is_grad_enabled() and w.requires_grad:
            return torch.cat((wa, wb - wa), dim=0)                        # rows: [Wa ; Wb - Wa]
        # frozen / no-grad use (Stage-II teacher): the stacked matrix only changes when the weight does
        cache = conv.__dict__.get("_act_stacked")
        key = (conv.weight.data_ptr(), conv.weight._version, conv.weight.device)
        if cache is None or cache[0] != key:
            with torch.no_grad():
                cache = (key, torch.cat((wa, wb - wa), dim=0))
            conv.__dict__["_act_stacked"] = cache
        return cache[1]

    @staticmethod
    def _edge_layer(f, idx, layer, B, G, out=None, ooff=0):
        conv, gn, _ = layer
        cout = conv.weight.shape[0]
        yz = K.linear(f, DGCNN._stacked_weight(conv), None)               # [BG, 2*Cout] = [Y | Z]
        if not torch.is_grad_enabled():                                   # frozen teacher (Stage II): written into the cat buffer
            return K.edge_gn_lrelu_max(yz, cout, idx, B, G, idx.shape[1], cout, gn, out=out, ooff=ooff)
        return K.edge_gn_lrelu_max_train(yz, cout, idx, B, G, idx.shape[1], cout, gn)     # Stage I: same kernels + HIP backward

    def features(self, f, coor, idx=None):
        """everything up to (not including) layer5's GroupNorm: -> pre-norm head output rows [B*G, C']"""
        B, G, Cin = f.shape
        if idx is None:
            with torch.no_grad():
                idx = self.graph_index(coor)
        if CP.ENABLED and not torch.is_grad_enabled():                     # frozen teacher: the whole stack in one host call
            stacked = [self._stacked_weight(l[0]) for l in (self.layer1, self.layer2, self.layer3, self.layer4)]
            return CP.dgcnn_features(self, f, idx, stacked)
        x = K.linear(f.reshape(B * G, Cin), _w2d(self.input_trans), self.input_trans.bias)
        fused = not torch.is_grad_enabled()
        cat = torch.empty(B * G, 2304, dtype=torch.float32, device=f.device) if fused else None
        feats, off = [], 0
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            cout = layer[0].weight.shape[0]
            x = self._edge_layer(x, idx, layer, B, G, out=cat, ooff=off)
            if fused:
                x = cat[:, off:off + cout]                                # written in place: no torch.cat of the 4 outputs
            feats.append(x)
            off += cout
        h_in = cat if fused else torch.cat(feats, dim=1)
        return K.linear(h_in, _w2d(self.layer5[0]), None)                 # [BG, C']

    def forward(self, f, coor, idx=None):
        """f [B,G,C], coor [B,G,3] -> [B,G,C']"""
        B, G, _ = f.shape
        h = self.features(f, coor, idx)
        gn = self.layer5[1]
        if not torch.is_grad_enabled():
            return K.edge_gn_lrelu_max(h, -1, None, B, G, 1, h.shape[1], gn).view(B, G, -1)
        return K.edge_gn_lrelu_max_train(h, -1, None, B, G, 1, h.shape[1], gn).view(B, G, -1)


class Decoder(nn.Module):
    """FoldingNet decoder (models/dvae.py:217-275)."""

    def __init__(self, encoder_channel, num_fine):
        super().__init__()
        self.num_fine = num_fine
        self.grid_size = 2
        self.num_coarse = self.num_fine // 4
        assert num_fine % 4 == 0
        self.mlp = nn.Sequential(nn.Linear(encoder_channel, 1024), nn.ReLU(inplace=True), nn.Linear(1024, 1024),
                                 nn.ReLU(inplace=True), nn.Linear(1024, 3 * self.num_coarse))
        self.final_conv = nn.Sequential(nn.Conv1d(encoder_channel + 3 + 2, 512, 1), nn.BatchNorm1d(512), nn.ReLU(inplace=True),
                                        nn.Conv1d(512, 512, 1), nn.BatchNorm1d(512), nn.ReLU(inplace=True), nn.Conv1d(512, 3, 1))
        lin = torch.linspace(-0.05, 0.05, steps=self.grid_size, dtype=torch.float)
        a = lin.view(1, self.grid_size).expand(self.grid_size, self.grid_size).reshape(1, -1)
        b = lin.view(self.grid_size, 1).expand(self.grid_size, self.grid_size).reshape(1, -1)
        # non-persistent buffer: follows .to(device) (a CPU attribute would cost a blocking H2D copy per step), not in the state_dict
        self.register_buffer("folding_seed", torch.cat([a, b], dim=0).view(1, 2, self.grid_size ** 2), persistent=False)

    def forward(self, feature_global):
        """feature_global [B,G,C] -> coarse [B,G,M/4,3], fine [B,G,M,3]"""
        bs, g, c = feature_global.shape
        fgl = feature_global.reshape(bs * g, c)
        m = self.mlp
        coarse = K.mlp_relu3(fgl, m[0].weight, m[0].bias, m[2].weight, m[2].bias, m[4].weight, m[4].bias).reshape(bs * g, self.num_coarse, 3)
        S = self.grid_size ** 2
        rep = coarse.unsqueeze(2).expand(-1, -1, S, -1).reshape(bs * g * self.num_fine, 3)      # rows (group, point)
        seed = self.folding_seed.to(fgl.device).view(2, S).t()                                   # [S,2]
        seed = seed.unsqueeze(0).expand(bs * g * self.num_coarse, -1, -1).reshape(bs * g * self.num_fine, 2)
        c1, bn1, _, c2, bn2, _, c3 = self.final_conv
        w1 = _w2d(c1)
        # conv over cat(feature_global, seed, point): the feature_global part is constant per group
        gw = K.linear(fgl, w1[:, :c], c1.bias)                                                  # [BG,512]
        h = K.linear_group_add(torch.cat((seed, rep), dim=1), w1[:, c:], gw, self.num_fine)
        h = K.batch_norm_act(h, bn1, self.training, relu=True)
        h = K.batch_norm_act(K.linear(h, _w2d(c2), c2.bias), bn2, self.training, relu=True)
        fine = K.linear(h, _w2d(c3), c3.bias) + rep
        return coarse.reshape(bs, g, self.num_coarse, 3), fine.reshape(bs, g, self.num_fine, 3)


class _FrozenBlock(nn.Module):
    """parameter container with the timm ViT block key names (norm1, attn.qkv/proj, norm2, mlp.fc1/fc2)."""

    def __init__(self, dim, heads, qkv_bias=True, eps=1e-6):
        super().__init__()
        self.num_heads, self.eps = heads, eps
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = nn.Module()
        self.attn.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn.proj = nn.Linear(dim, dim)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = nn.Module()
        self.mlp.fc1 = nn.Linear(dim, dim * 4)
        self.mlp.fc2 = nn.Linear(dim * 4, dim)

    def forward(self, x, pos=None, gate1=None, gate2=None, train_w=False):
        a, m = self.attn, self.mlp
        return K.BlockFn.apply(x, pos, gate1, gate2, self.norm1.weight, self.norm1.bias, a.qkv.weight, a.qkv.bias,
                               a.proj.weight, a.proj.bias, self.norm2.weight, self.norm2.bias, m.fc1.weight, m.fc1.bias,
                               m.fc2.weight, m.fc2.bias, self.num_heads, self.eps, train_w)


_VIT_GEOMETRY = {            # timm names -> (depth, heads); width must equal config.visual_embed_dim
    "vit_base_patch16_384": (12, 12), "vit_base_patch16_224": (12, 12), "vit_small_patch16_384": (12, 6),
    "vit_small_patch16_224": (12, 6), "deit_base_distilled_patch16_384": (12, 12),
}


@MODELS.register_module()
class DiscreteVAE(nn.Module):
    """The plain Point-BERT tokenizer / autoencoder (models/dvae.py:278-358, recipe cfgs/autoencoder/pointbert_dvae.yaml):
    Group -> mini-PointNet -> DGCNN -> gumbel-softmax over the codebook -> DGCNN -> FoldingNet; everything trainable.
    ``forward(inp, temperature, hard)`` -> the 6-tuple, ``get_loss(ret, gt)`` -> (loss_recon, loss_klv)."""

    def __init__(self, config, **kwargs):
        super().__init__()
        self.group_size = config.group_size
        self.num_group = config.num_group
        self.encoder_dims = config.encoder_dims
        self.tokens_dims = config.tokens_dims
        self.decoder_dims = config.decoder_dims
        self.num_tokens = config.num_tokens
        self.group_divider = Group(num_group=self.num_group, group_size=self.group_size, skip_near_origin=config.get("fps_skip_near_origin", None))
        self.encoder = Encoder(encoder_channel=self.encoder_dims)
        self.dgcnn_1 = DGCNN(encoder_channel=self.encoder_dims, output_channel=self.num_tokens)
        self.codebook = nn.Parameter(torch.randn(self.num_tokens, self.tokens_dims))
        self.dgcnn_2 = DGCNN(encoder_channel=self.tokens_dims, output_channel=self.decoder_dims)
        self.decoder = Decoder(encoder_channel=self.decoder_dims, num_fine=self.group_size)
        self.build_loss_func()

    def build_loss_func(self):
        self.loss_func_cdl1 = ChamferDistanceL1()
        self.loss_func_cdl2 = ChamferDistanceL2()

    # ---- losses (Stage I) ------------------------------------------------------------------------
    def recon_loss(self, ret, gt):
        whole_coarse, whole_fine, coarse, fine, group_gt, _ = ret
        bs, g, _, _ = coarse.shape
        coarse = coarse.reshape(bs * g, -1, 3).contiguous()
        fine = fine.reshape(bs * g, -1, 3).contiguous()
        group_gt = group_gt.reshape(bs * g, -1, 3).contiguous()
        return self.loss_func_cdl1(coarse, group_gt) + self.loss_func_cdl1(fine, group_gt)

    def get_loss(self, ret, gt):
        loss_recon = self.recon_loss(ret, gt)
        loss_klv = K.kl_to_uniform(ret[-1])        # KL(mean_g softmax(logits) || uniform), 'batchmean' (models/dvae.py:470-476)
        return loss_recon, loss_klv

    def visual_embedding(self, input, center, draws=None, rng=None):
        return input                               # the plain tokenizer feeds the codebook vectors straight to dgcnn_2 (:340, :350)

    def _rng(self, device):
        """(base seed, device-resident step counter) of the in-kernel Philox draws of the frozen-teacher path.  The counter lives
        in HBM and is bumped by a device op per teacher forward, so a captured hipGraph of that forward draws fresh noise on
        every replay and eager / replayed executions produce the same sequence."""
        st = self.__dict__.get("_rng_state")
        if st is None or st[1].device != device:
            st = (int(torch.randint(0, 2 ** 62, (1,)).item()), torch.zeros(1, dtype=torch.int64, device=device))
            self.__dict__["_rng_state"] = st
        return st

    # ---- tokenizer ----------------------------------------------------------------------------------
    def _gumbel_codes(self, logits, tau, hard, draws):
        g = None
        if draws is not None and (draws.has("gumbel") or draws.record):
            g = draws.get("gumbel", lambda: -torch.empty_like(logits).exponential_().log())
        if hard:
            if g is None:
                g = -torch.empty_like(logits).exponential_().log()
            index = ((logits + g) / tau).argmax(dim=-1)    # one-hot x codebook == row gather (models/dvae.py:587-588)
            codes = F.embedding(index, self.codebook)
            if torch.is_grad_enabled() and logits.requires_grad:
                # straight-through estimator of F.gumbel_softmax(hard=True): y = y_hard - y_soft.detach() + y_soft, so the value is the
                # codebook row while the gradient reaches the logits through y_soft (and the codebook only through the selected rows)
                soft = K.gumbel_softmax(logits, tau, noise=g)
                through = K.linear(soft, self.codebook.detach().t().contiguous(), None)
                codes = codes + (through - through.detach())
            return codes
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if g is None else 0       # host RNG (no device sync)
        y = K.gumbel_softmax(logits, tau, noise=g, seed=seed)                       # soft one-hot [B,G,N], noise from Philox in-kernel
        return K.linear(y, self.codebook.t().contiguous(), None)

    def forward_tokenizer(self, neighborhood, center):
        gt_logits = self.dgcnn_1(self.encoder(neighborhood), center)
        return gt_logits.argmax(-1).long()

    def forward_tokenizer_features(self, neighborhood, center, return_global=True, draws=None):
        with torch.no_grad():
            idx = DGCNN.graph_index(center)                # the k=4 graph is identical for all 8 edge-conv layers
        if not torch.is_grad_enabled():
            # frozen teacher: head GroupNorm + LeakyReLU + hard gumbel + codebook lookup in one pass over the logits
            B, G, _ = center.shape
            h = self.dgcnn_1.features(self.encoder(neighborhood), center, idx)
            noise = None
            if draws is not None and (draws.has("gumbel") or draws.record):
                noise = draws.get("gumbel", lambda: -torch.empty(B, G, h.shape[1], device=h.device).exponential_().log())
            rng = self._rng(h.device)
            sampled, _, _ = K.gn_gumbel_argmax_gather(h, B, G, self.dgcnn_1.layer5[1], self.codebook, noise=noise,
                                                      seed=rng[0] ^ 0x5bd1e995, seed_dev=rng[1])
            feature = self.visual_embedding(sampled, center, draws, rng)
            if return_global:
                feature = self.dgcnn_2(feature, center, idx)
            rng[1].add_(1)                                 # next teacher forward draws a fresh Philox stream
            return feature
        logits = self.dgcnn_1(self.encoder(neighborhood), center, idx)
        sampled = self._gumbel_codes(logits, 1.0, True, draws)
        feature = self.visual_embedding(sampled, center, draws)
        if return_global:
            feature = self.dgcnn_2(feature, center, idx)
        return feature

    def forward(self, inp, temperature=1., hard=False, draws=None, **kwargs):
        neighborhood, center = self.group_divider(inp)
        with torch.no_grad():
            idx = DGCNN.graph_index(center)
        logits = self.dgcnn_1(self.encoder(neighborhood), center, idx)
        sampled = self._gumbel_codes(logits, temperature, hard, draws)
        sampled = self.visual_embedding(sampled, center, draws)
        feature = self.dgcnn_2(sampled, center, idx)
        coarse, fine = self.decoder(feature)
        with torch.no_grad():
            whole_fine = (fine + center.unsqueeze(2)).reshape(inp.size(0), -1, 3)
            whole_coarse = (coarse + center.unsqueeze(2)).reshape(inp.size(0), -1, 3)
        assert fine.size(2) == self.group_size
        return (whole_coarse, whole_fine, coarse, fine, neighborhood, logits)


@MODELS.register_module()
class ACTPromptedDiscreteVAEwithVIT(DiscreteVAE):
    """Stage-I autoencoder / Stage-II frozen teacher (models/dvae.py:360-615): DiscreteVAE with the prompt-tuned image Transformer
    between the codebook lookup and dgcnn_2.

    The pretrained image Transformer is represented by its ``blocks`` + ``norm`` (what the reference keeps,
    :405-410) with timm's key names; weights come from the dVAE checkpoint (``ckpt``), never from the network.
    Optional config keys (absent from the reference YAML): ``visual_embed_depth`` / ``visual_embed_heads``
    override the geometry implied by ``visual_embed_type``."""

    def __init__(self, config, **kwargs):
        super().__init__(config)
        self.visual_embed_type = config.visual_embed_type
        self.visual_embed_dim = config.visual_embed_dim
        self.freeze_visual_embed = config.freeze_visual_embed
        self.num_prompt_token = config.num_prompt_token
        self.use_deep_prompt = config.use_deep_prompt
        if self.use_deep_prompt and (self.visual_embed_dim == 'none' or self.num_prompt_token <= 0):
            raise ValueError("use_deep_prompt needs an image Transformer and num_prompt_token > 0 (the reference fails on this combination too)")
        self.build_visual_embedding(config)

    def build_visual_embedding(self, config):
        """models/dvae.py:390-444.  Configurations: deep prompts (the ACT recipe, `use_deep_prompt`), shallow prompts (prepended once), no
        prompts (`num_prompt_token` 0), no image Transformer (`visual_embed_dim: none`)."""
        if self.visual_embed_dim == 'none':
            self.visual_embed = None
            return
        depth, heads = _VIT_GEOMETRY.get(self.visual_embed_type, (12, 12))
        depth = int(config.get("visual_embed_depth", depth))
        heads = int(config.get("visual_embed_heads", heads))
        D = self.visual_embed_dim
        blocks = nn.Sequential(*[_FrozenBlock(D, heads) for _ in range(depth)])
        self.visual_embed = nn.Sequential(blocks, nn.LayerNorm(D, eps=1e-6))
        self.visual_embed_depth = depth
        self.proj_pre = nn.Linear(self.tokens_dims, D)
        self.visual_pos_embed = nn.Sequential(nn.Linear(3, 128), nn.GELU(), nn.Linear(128, D))
        self.proj_post = nn.Linear(D, self.tokens_dims)
        Pn = self.num_prompt_token
        if Pn > 0:
            self.visual_prompt_proj = nn.Identity()
            self.prompt_dropout = nn.Dropout(0.1)
            self.visual_prompt_token = nn.Parameter(torch.zeros(1, Pn, D))
            self.visual_prompt_pos = nn.Parameter(torch.randn(1, Pn, D))
            trunc_normal_(self.visual_prompt_token, std=.02)
            trunc_normal_(self.visual_prompt_pos, std=.02)
            if self.use_deep_prompt:
                self.deep_prompt_tokens = nn.Parameter(torch.zeros(depth - 1, Pn, D))
                self.deep_prompt_pos = nn.Parameter(torch.randn(depth - 1, Pn, D))
                trunc_normal_(self.deep_prompt_tokens, std=.02)
                trunc_normal_(self.deep_prompt_pos, std=.02)
        else:
            self.visual_prompt_token = None
        if self.freeze_visual_embed:
            for param in self.visual_embed.parameters():
                param.requires_grad = False

    # ---- prompt-tuned frozen Transformer ---------------------------------------------------------
    def _drop(self, t, draws, key):
        p = self.prompt_dropout.p
        if not self.training or p == 0:
            return t
        if draws is not None and draws.has(key):
            return t * draws.get(key, None).to(t.dtype) / (1.0 - p)
        if draws is not None and draws.record:
            keep = draws.get(key, lambda: (torch.rand_like(t) >= p).to(t.dtype))
            return t * keep / (1.0 - p)
        return F.dropout(t, p, True)

    def _prompt_rows(self, tok, ppos, B, draws, key):
        """dropout(tok) + ppos for every cloud, [B*Pn, D], one HIP launch per direction (K.prompt_rows): injected / recorded keep masks
        when the parity tests ask for them, in-kernel Philox otherwise."""
        p = self.prompt_dropout.p if self.training else 0.0
        mask = None
        if p > 0 and draws is not None and (draws.has(key) or draws.record):
            shape = (B,) + tuple(tok.shape)
            mask = draws.get(key, lambda: (torch.rand(shape, device=tok.device) >= p).to(tok.dtype))
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if (p > 0 and mask is None) else 0       # host RNG: no device sync
        return K.prompt_rows(tok, ppos, B, p, seed, mask)

    def visual_embedding(self, input, center, draws=None, rng=None):
        if self.visual_embed is None:
            return input
        if self.use_deep_prompt:
            return self.visual_embedding_deep_prompt(input, center, draws=draws, rng=rng)
        return self._visual_embedding_shallow(input, center, draws)

    def _visual_embedding_shallow(self, input, center, draws=None):
        """models/dvae.py:517-534: the Transformer applied as `x = blk(x + pos)` (forward_visual_feature :500-511), either on the patch
        tokens alone -- under no_grad when it is frozen (:522-524), so nothing upstream receives a gradient through it -- or behind prompts
        that are prepended ONCE (incorporate_prompt :485-498) and whose outputs flow through all blocks before being dropped."""
        B = input.shape[0]
        Pn = self.num_prompt_token
        vp = self.visual_pos_embed
        pos = K.mlp(center, vp[0].weight, vp[0].bias, vp[2].weight, vp[2].bias)
        hidden = K.linear(input, self.proj_pre.weight, self.proj_pre.bias)
        train_w = not self.freeze_visual_embed
        nrm = self.visual_embed[1]

        def stack(h, p):
            for blk in self.visual_embed[0]:
                h = blk(h, p, None, None, train_w)
            return h
        if self.visual_prompt_token is None:
            if self.freeze_visual_embed:
                with torch.no_grad():
                    hidden = stack(hidden, pos)
            else:
                hidden = stack(hidden, pos)
        else:
            hidden = torch.cat((self._drop(self.visual_prompt_token.expand(B, -1, -1), draws, "prompt.0"), hidden), dim=1)
            pos = torch.cat((self.visual_prompt_pos.expand(B, -1, -1), pos), dim=1)
            hidden = stack(hidden, pos)[:, Pn:].contiguous()
        return K.linear(K.layer_norm(hidden, nrm.weight, nrm.bias, nrm.eps), self.proj_post.weight, self.proj_post.bias)

    def _visual_embedding_prefix(self, input, center, draws=None, rng=None):
        """Inference form of visual_embedding_deep_prompt.  Every layer REPLACES the prompt rows of its input
        (models/dvae.py:556-566) and the final output drops them (:574), so prompt tokens only ever act as keys/values:
        queries, projection and MLP are evaluated for the G patch tokens only (58 % of the FLOPs, identical outputs)."""
        B, G, _ = input.shape
        Pn, D = self.num_prompt_token, self.visual_embed_dim
        drop_p = self.prompt_dropout.p if self.training else 0.0
        fused = draws is None or not (draws.record or any(draws.has(f"prompt.{i}") for i in range(self.visual_embed_depth)))
        base, ctr = rng if rng is not None else self._rng(input.device)
        if fused and CP.ENABLED:                                         # the whole 12-layer stack in one host call (~125 launches)
            return CP.prefix_vit_forward(self, input, center, drop_p, base, ctr)
        vp = self.visual_pos_embed
        pos = K.mlp(center, vp[0].weight, vp[0].bias, vp[2].weight, vp[2].bias).reshape(B * G, D)
        x = K.linear(input, self.proj_pre.weight, self.proj_pre.bias).reshape(B * G, D)
        blocks = self.visual_embed[0]
        for i in range(self.visual_embed_depth):
            tok = self.visual_prompt_token[0] if i == 0 else self.deep_prompt_tokens[i - 1]
            ppos = self.visual_prompt_pos[0] if i == 0 else self.deep_prompt_pos[i - 1]
            blk = blocks[i]
            a, m = blk.attn, blk.mlp
            prm = n1p = None
            if fused:       # dropout (in-kernel Philox) + prompt position + LayerNorm in one launch
                n1p = K.prompt_layernorm(tok, ppos, B, drop_p, (base + 7919 * (i + 1)) & (2 ** 62 - 1), blk.norm1.weight, blk.norm1.bias,
                                         blk.eps, seed_dev=ctr)
            else:           # injected / recorded dropout masks (parity tests)
                prm = (self._drop(tok.unsqueeze(0).expand(B, -1, -1), draws, f"prompt.{i}") + ppos).reshape(B * Pn, D)
            x = K.block_forward_prefix(x, pos, prm, B, Pn, G, blk.norm1.weight, blk.norm1.bias, a.qkv.weight, a.qkv.bias,
                                       a.proj.weight, a.proj.bias, blk.norm2.weight, blk.norm2.bias, m.fc1.weight, m.fc1.bias,
                                       m.fc2.weight, m.fc2.bias, blk.num_heads, blk.eps, n1p=n1p)
        nrm = self.visual_embed[1]
        feature = K.layer_norm(x, nrm.weight, nrm.bias, nrm.eps)
        return K.linear(feature, self.proj_post.weight, self.proj_post.bias).reshape(B, G, -1)

    def _visual_embedding_prefix_train(self, input, center, draws=None):
        """Stage-I training form of the same restructuring (frozen block weights, learnable prompts / projections): gradients
        reach the prompts through their keys/values only, exactly as in the reference graph."""
        B, G, _ = input.shape
        Pn, D = self.num_prompt_token, self.visual_embed_dim
        vp = self.visual_pos_embed
        pos = K.mlp(center, vp[0].weight, vp[0].bias, vp[2].weight, vp[2].bias).reshape(B * G, D)
        x = K.linear(input, self.proj_pre.weight, self.proj_pre.bias).reshape(B * G, D)
        blocks = self.visual_embed[0]
        for i in range(self.visual_embed_depth):
            tok = self.visual_prompt_token[0] if i == 0 else self.deep_prompt_tokens[i - 1]
            ppos = self.visual_prompt_pos[0] if i == 0 else self.deep_prompt_pos[i - 1]
            prm = self._prompt_rows(tok, ppos, B, draws, f"prompt.{i}")
            blk = blocks[i]
            a, m = blk.attn, blk.mlp
            x = K.PrefixBlockFn.apply(x, pos, prm, B, Pn, G, blk.norm1.weight, blk.norm1.bias, a.qkv.weight, a.qkv.bias,
                                      a.proj.weight, a.proj.bias, blk.norm2.weight, blk.norm2.bias, m.fc1.weight, m.fc1.bias,
                                      m.fc2.weight, m.fc2.bias, blk.num_heads, blk.eps)
        nrm = self.visual_embed[1]
        feature = K.layer_norm(x, nrm.weight, nrm.bias, nrm.eps)
        return K.linear(feature, self.proj_post.weight, self.proj_post.bias).reshape(B, G, -1)

    def visual_embedding_deep_prompt(self, input, center, permute_feature=False, draws=None, rng=None):
        """models/dvae.py:536-576 (+ incorporate_prompt :485-498): prompts replaced (not appended) at layers 1..L-1."""
        if not torch.is_grad_enabled():
            return self._visual_embedding_prefix(input, center, draws, rng)
        if self.freeze_visual_embed:
            return self._visual_embedding_prefix_train(input, center, draws)
        B, G, _ = input.shape
        Pn = self.num_prompt_token
        vp = self.visual_pos_embed
        pos_tok = K.mlp(center, vp[0].weight, vp[0].bias, vp[2].weight, vp[2].bias)                 # [B,G,D]
        feature = K.linear(input, self.proj_pre.weight, self.proj_pre.bias)
        train_w = not self.freeze_visual_embed
        hidden = torch.cat((self._drop(self.visual_prompt_token.expand(B, -1, -1), draws, "prompt.0"), feature), dim=1)
        pos = torch.cat((self.visual_prompt_pos.expand(B, -1, -1), pos_tok), dim=1)
        blocks = self.visual_embed[0]
        for i in range(self.visual_embed_depth):
            if i > 0:
                prm = self._drop(self.deep_prompt_tokens[i - 1].expand(B, -1, -1), draws, f"prompt.{i}")
                hidden = torch.cat((prm, hidden[:, Pn:, :]), dim=1)
                pos = torch.cat((self.deep_prompt_pos[i - 1].expand(B, -1, -1), pos_tok), dim=1)
            hidden = blocks[i](hidden, pos, None, None, train_w)
        nrm = self.visual_embed[1]
        feature = K.layer_norm(hidden[:, Pn:].contiguous(), nrm.weight, nrm.bias, nrm.eps)
        return K.linear(feature, self.proj_post.weight, self.proj_post.bias)
