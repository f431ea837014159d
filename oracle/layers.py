"""Oracle (TEST INFRASTRUCTURE): pure-PyTorch CPU restatement of the dense layers
on the ACT hot path.  Attribute names equal the reference's so a reference
``state_dict`` loads strictly; arithmetic is plain fp32 torch ops.

Reference files restated (file:line in /root/reference):
  Mlp / Attention / Block             models/act.py:25-90  (== utils/transformer_layers.py:138-232)
  DropPath                            utils/transformer_layers.py:105-120 (timm DropPath)
  TransformerEncoder / Decoder        models/act.py:93-145
  mini-PointNet Encoder               models/dvae.py:185-215
  DGCNN                               models/dvae.py:26-117
  FoldingNet Decoder                  models/dvae.py:217-275
  NegativeCosineSimilarity            lightly 1.2.28 (un-vendored): -cosine_similarity(x0,x1,dim=1,eps=1e-8).mean()
"""
import math
import torch
import torch.nn as nn
import torch.nn.functional as F

from .point_ops import knn_ref


class Draws:
    """Keyed table of random draws.  ``get(key, make)`` returns the stored tensor
    for ``key`` or calls ``make()`` (and stores it when recording).  Lets a test
    run the oracle once, then replay the identical draws into the HIP path."""

    def __init__(self, table=None, record=False):
        self.table = dict(table) if table else {}
        self.record = record

    def get(self, key, make):
        if key in self.table:
            return self.table[key]
        t = make()
        if self.record:
            self.table[key] = t
        return t


def drop_path(x, p, training, draws, key):
    """per-sample stochastic depth: x / keep * floor(keep + U[0,1))."""
    if p == 0.0 or not training:
        return x
    keep = 1.0 - p
    B = x.shape[0]
    u = draws.get(key, lambda: torch.rand(B, dtype=x.dtype, device=x.device))
    gate = torch.floor(keep + u).view((B,) + (1,) * (x.dim() - 1))
    return x.div(keep) * gate


class Mlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))          # exact erf GELU; dropout p=0


class Attention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias=False):
        super().__init__()
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, S, C = x.shape
        h = self.num_heads
        qkv = self.qkv(x).view(B, S, 3, h, C // h).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        a = torch.matmul(q, k.transpose(-2, -1)) * self.scale     # scale AFTER q@k^T (act.py:62)
        a = torch.softmax(a, dim=-1)
        o = torch.matmul(a, v).transpose(1, 2).reshape(B, S, C)
        return self.proj(o)


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, drop_path=0.0, eps=1e-5, tag=""):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads, qkv_bias)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))
        self.dp = float(drop_path)
        self.tag = tag

    def forward(self, x, draws, tag=None):
        tag = self.tag if tag is None else tag
        x = x + drop_path(self.attn(self.norm1(x)), self.dp, self.training, draws, tag + ".attn")
        x = x + drop_path(self.mlp(self.norm2(x)), self.dp, self.training, draws, tag + ".mlp")
        return x


class BlockList(nn.Module):
    """holds ``blocks`` so keys read  <name>.blocks.{i}.*"""

    def __init__(self, dim, depth, heads, dpr, qkv_bias=False, eps=1e-5, tag=""):
        super().__init__()
        self.blocks = nn.ModuleList([
            Block(dim, heads, 4.0, qkv_bias, dpr[i] if isinstance(dpr, (list, tuple)) else dpr, eps, f"{tag}.{i}")
            for i in range(depth)])


class TransformerEncoder(BlockList):
    def forward(self, x, pos, draws):
        for blk in self.blocks:                    # pos added before EVERY block (act.py:109-112)
            x = blk(x + pos, draws)
        return x


class TransformerDecoder(BlockList):
    def __init__(self, dim, depth, heads, dpr, tag="dec"):
        super().__init__(dim, depth, heads, dpr, tag=tag)
        self.norm = nn.LayerNorm(dim)

    def forward(self, x, pos, return_token_num, draws, tag="dec"):
        for i, blk in enumerate(self.blocks):
            x = blk(x + pos, draws, f"{tag}.{i}")
        return self.norm(x[:, -return_token_num:])


class Encoder(nn.Module):
    """mini-PointNet (models/dvae.py:185-215)."""

    def __init__(self, encoder_channel):
        super().__init__()
        self.encoder_channel = encoder_channel
        self.first_conv = nn.Sequential(nn.Conv1d(3, 128, 1), nn.BatchNorm1d(128), nn.ReLU(inplace=True),
                                        nn.Conv1d(128, 256, 1))
        self.second_conv = nn.Sequential(nn.Conv1d(512, 512, 1), nn.BatchNorm1d(512), nn.ReLU(inplace=True),
                                         nn.Conv1d(512, encoder_channel, 1))

    def forward(self, point_groups):
        bs, g, n, _ = point_groups.shape
        x = point_groups.reshape(bs * g, n, 3).transpose(2, 1)
        f = self.first_conv(x)                                      # BG 256 n
        fg = f.max(dim=2, keepdim=True)[0]
        f = torch.cat([fg.expand(-1, -1, n), f], dim=1)             # BG 512 n
        f = self.second_conv(f)
        return f.max(dim=2)[0].reshape(bs, g, self.encoder_channel)


def knn_graph_ref(coor, k=4):
    """KNN(k=4, transpose_mode=False)(coor, coor) of models/dvae.py:23,68 on [B,G,3] -> idx int64 [B,k,G]."""
    _, idx = knn_ref(coor.detach().cpu().numpy(), coor.detach().cpu().numpy(), k)
    return torch.from_numpy(idx).permute(0, 2, 1).contiguous().to(coor.device)


class DGCNN(nn.Module):
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
    def graph_feature(x, idx):
        """x [B,C,G], idx [B,k,G] -> [B,2C,G,k] = cat(nbr - x, x)  (models/dvae.py:60-79)."""
        B, C, G = x.shape
        k = idx.shape[1]
        xt = x.transpose(2, 1)                                       # B G C
        nb = xt[torch.arange(B, device=x.device).view(B, 1, 1), idx]  # B k G C
        nb = nb.permute(0, 3, 2, 1)                                  # B C G k
        xq = x.unsqueeze(-1).expand(-1, -1, -1, k)
        return torch.cat((nb - xq, xq), dim=1)

    def forward(self, f, coor, idx=None):
        if idx is None:
            idx = knn_graph_ref(coor, 4)
        f = self.input_trans(f.transpose(1, 2))                      # B 128 G
        feats = []
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            f = layer(self.graph_feature(f, idx)).max(dim=-1)[0]
            feats.append(f)
        f = self.layer5(torch.cat(feats, dim=1))                     # B C' G
        return f.transpose(-1, -2)


class Decoder(nn.Module):
    """FoldingNet decoder (models/dvae.py:217-275)."""

    def __init__(self, encoder_channel, num_fine):
        super().__init__()
        self.num_fine = num_fine
        self.grid_size = 2
        self.num_coarse = num_fine // 4
        self.mlp = nn.Sequential(nn.Linear(encoder_channel, 1024), nn.ReLU(inplace=True),
                                 nn.Linear(1024, 1024), nn.ReLU(inplace=True),
                                 nn.Linear(1024, 3 * self.num_coarse))
        self.final_conv = nn.Sequential(nn.Conv1d(encoder_channel + 3 + 2, 512, 1), nn.BatchNorm1d(512),
                                        nn.ReLU(inplace=True), nn.Conv1d(512, 512, 1), nn.BatchNorm1d(512),
                                        nn.ReLU(inplace=True), nn.Conv1d(512, 3, 1))
        lin = torch.linspace(-0.05, 0.05, steps=2, dtype=torch.float)
        a = lin.view(1, 2).expand(2, 2).reshape(1, -1)
        b = lin.view(2, 1).expand(2, 2).reshape(1, -1)
        self.folding_seed = torch.cat([a, b], dim=0).view(1, 2, 4)

    def forward(self, feature_global):
        bs, g, c = feature_global.shape
        fgl = feature_global.reshape(bs * g, c)
        coarse = self.mlp(fgl).reshape(bs * g, self.num_coarse, 3)
        rep = coarse.unsqueeze(2).expand(-1, -1, 4, -1).reshape(bs * g, self.num_fine, 3).transpose(2, 1)
        seed = self.folding_seed.unsqueeze(2).expand(bs * g, -1, self.num_coarse, -1)
        seed = seed.reshape(bs * g, -1, self.num_fine).to(fgl.device)
        feat = torch.cat([fgl.unsqueeze(2).expand(-1, -1, self.num_fine), seed, rep], dim=1)
        fine = self.final_conv(feat) + rep
        fine = fine.reshape(bs, g, 3, self.num_fine).transpose(-1, -2)
        return coarse.reshape(bs, g, self.num_coarse, 3), fine


def cosine_distill_loss(student, teacher):
    """models/act.py:1243-1254 with loss='cosine': mean over (b, token) of 1 - cos."""
    B = student.shape[0]
    loss = student.new_zeros(1)
    for b in range(B):
        loss = loss + (1 + (-F.cosine_similarity(student[b], teacher[b], dim=1, eps=1e-8)).mean())
    return loss.mean() / B


def ntxent_pair_loss(out0, out1, temperature=0.07):
    """lightly 1.2.28 (requirements.txt:14; NOT installed in this image -> restated from its published source, parity unpinned) loss.NTXentLoss
    (temperature, memory_bank_size=0).forward(out0, out1) on two [n, C] views: rows L2-normalised, the 2n x 2n cosine matrix / temperature with
    its diagonal removed, cross-entropy (mean over the 2n anchors) towards the other view of the same row."""
    n = out0.shape[0]
    z = torch.cat([F.normalize(out0, dim=1), F.normalize(out1, dim=1)], dim=0)
    logits = z @ z.t() / temperature
    logits = logits[~torch.eye(2 * n, dtype=torch.bool)].view(2 * n, -1)
    labels = torch.arange(n)
    labels = torch.cat([labels + n - 1, labels])
    return F.cross_entropy(logits, labels)


def barlow_pair_loss(z_a, z_b, lambda_param=5e-3):
    """lightly 1.2.28 loss.BarlowTwinsLoss(lambda_param).forward(z_a, z_b) on two [n, D] views (same provenance note as ntxent_pair_loss): both
    standardised along the batch (unbiased std), c = z_a^T z_b / n, sum of (c - I)^2 with the off-diagonal terms weighted by lambda."""
    n, d = z_a.shape
    za = (z_a - z_a.mean(0)) / z_a.std(0)
    zb = (z_b - z_b.mean(0)) / z_b.std(0)
    c = za.t() @ zb / n
    c_diff = (c - torch.eye(d)).pow(2)
    off = ~torch.eye(d, dtype=torch.bool)
    c_diff = torch.where(off, c_diff * lambda_param, c_diff)
    return c_diff.sum()


def pairwise_distill_loss(student, teacher, kind, num_mask):
    """models/act.py:1243-1254 for the losses that are neither 'cosine' nor 'l1' / 'l2' named: per cloud loss_func(student[b], teacher[b]) / num_mask,
    summed over the batch, / batch size.  kind: 'ntxent' (temperature 0.07, :1193) or 'barlow' (lambda 5e-3, :1195)."""
    B = student.shape[0]
    fn = ntxent_pair_loss if kind == "ntxent" else barlow_pair_loss
    loss = student.new_zeros(1)
    for b in range(B):
        loss = loss + fn(student[b], teacher[b]) / num_mask
    return loss.mean() / B


def trunc_normal_(t, std=0.02):
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)
