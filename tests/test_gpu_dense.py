"""GPU parity (through the C ABI): fp32 MFMA GEMM + fused epilogues, LayerNorm fwd/bwd, fused attention fwd/bwd,
the fused Transformer block and the cosine loss -- against float64 CPU math, the CPU oracle and the goldens.
Tolerance: 1e-4 (north_star) relative to max(1, |ref|max) unless stated."""
import numpy as np
import os
import pytest
import torch

from tests.conftest import golden
from tests.golden.fill import fill_module, fill_tensor

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def K():
    assert torch.cuda.is_available()
    import act_amd.kernels as K
    return K


def _rel(a, ref):
    a = a.detach().double().cpu(); ref = ref.detach().double().cpu()
    assert a.shape == ref.shape, (a.shape, ref.shape)
    return ((a - ref).abs().max() / max(1.0, ref.abs().max())).item()


def _rnd(name, *shape):
    return fill_tensor(name, shape, "code")


@pytest.mark.parametrize("M,N,K_", [(256, 384, 384), (1792, 1152, 384), (100, 70, 33), (1, 5, 3), (257, 130, 19), (4096, 768, 3072),
                                    (8192, 128, 3), (64, 64, 16), (300, 512, 8)])
def test_gemm_layouts(K, M, N, K_):
    a = _rnd(f"ga{M}{K_}", M, K_); b = _rnd(f"gb{N}{K_}", N, K_)
    ref = a.double() @ b.double().t()
    scale = max(1.0, ref.abs().max().item())
    ad, bd = a.cuda(), b.cuda()
    for ak, bk in [(True, True), (True, False), (False, False), (False, True)]:
        A = ad if ak else ad.t().contiguous()
        Bm = bd if bk else bd.t().contiguous()
        c = K.gemm(A, Bm, ak, bk)
        err = (c.double().cpu() - ref).abs().max().item() / scale
        assert err <= 2e-5 * max(1, K_ / 256) ** 0.5 + 1e-6, (ak, bk, err)


def test_gemm_splitk_weight_gradient_shape(K):
    T, O, I = 16384, 384, 256
    dy = _rnd("sk.dy", T, O).cuda(); x = _rnd("sk.x", T, I).cuda()
    dw = K.gemm(dy, x, False, False)                       # dW[O,I] = dy^T x, K = T -> split-K path
    ref = dy.double().cpu().t() @ x.double().cpu()
    assert _rel(dw, ref) <= 1e-5
    dw2 = K.gemm(dy, x, False, False)
    assert torch.equal(dw, dw2)                            # deterministic


def test_gemm_epilogues(K):
    M, N, Kd, S = 256, 192, 128, 32
    a = _rnd("e.a", M, Kd).cuda(); w = (_rnd("e.w", N, Kd) * 0.1).cuda(); bias = _rnd("e.b", N).cuda()
    res = _rnd("e.r", M, N).cuda(); gate = (torch.arange(M // S) % 2).float().cuda() / 0.9
    base = (a.double() @ w.double().t()).cpu()
    # bias + GELU (+ pre-activation saved)
    aux = torch.empty(M, N, device="cuda")
    y = K.gemm(a, w, bias=bias, act=K.EPI_GELU, aux=aux)
    pre = base + bias.double().cpu()
    assert _rel(aux, pre) <= 1e-5 and _rel(y, torch.nn.functional.gelu(pre)) <= 1e-5
    # bias + per-sample scale + residual
    y = K.gemm(a, w, bias=bias, rowscale=gate, rows_per_scale=S, res=res)
    ref = res.double().cpu() + gate.double().cpu().repeat_interleave(S)[:, None] * pre
    assert _rel(y, ref) <= 1e-5
    # multiply by gelu'(aux)
    h = _rnd("e.h", M, N).cuda()
    y = K.gemm(a, w, act=K.EPI_MUL_GELU_GRAD, aux=h)
    hd = h.double().cpu().requires_grad_(True)
    torch.nn.functional.gelu(hd).sum().backward()
    assert _rel(y, base * hd.grad) <= 1e-5
    # relu, relu mask, accumulate, alpha
    assert _rel(K.gemm(a, w, act=K.EPI_RELU), base.clamp_min(0)) <= 1e-5
    assert _rel(K.gemm(a, w, act=K.EPI_MUL_RELU_MASK, aux=h), base * (h.double().cpu() > 0)) <= 1e-5
    c = res.clone()
    K.gemm(a, w, out=c, accumulate=True, alpha=0.5)
    assert _rel(c, res.double().cpu() + 0.5 * base) <= 1e-5


def test_gelu_epilogue_against_the_float64_function(K):
    """The one-polynomial GELU / GELU gradient of the GEMM epilogues (csrc/gemm_common.h, round 6; nn.GELU() of utils/transformer_layers.py:130-139) on the device,
    element by element against the float64 function: a product with the identity weight hands every x to the epilogue exactly.  Bars from the fp32 emulation of
    benchmarks/fit_gelu_poly.py (2.4e-7 / 1.5e-7 max abs error; torch's own fp32 erf form has 1.2e-6 / 2.9e-7) with head-room for v_exp_f32 (1 ulp): 4e-7 max(1, |x|)
    for the value, 4e-7 for the gradient -- and the tails: x = -100 gives -0 (not NaN, not -7e-7), x = +100 gives x."""
    from scipy.special import erf
    N = 128
    g = torch.Generator().manual_seed(5)
    xs = torch.cat([torch.linspace(-12, 12, 384 * N - 2048), torch.randn(2048 - 16, generator=g) * 1.5,
                    torch.tensor([0.0, -0.0, 1e-30, -1e-30, 1e-6, -1e-6, 20.0, -20.0, 100.0, -100.0, 5.6568, -5.6568, 5.66, -5.66, 0.7071, -0.7071])]).float()
    x = xs.view(-1, N).cuda().contiguous(); M = x.shape[0]
    eye = torch.eye(N, device="cuda")
    aux = torch.empty(M, N, device="cuda")
    y = K.gemm(x, eye, act=K.EPI_GELU, aux=aux)
    assert torch.equal(aux, x)                                        # the epilogue saw exactly x
    xd = x.double().cpu().numpy()
    ref = 0.5 * xd * (1.0 + erf(xd / np.sqrt(2.0)))
    err = np.abs(y.double().cpu().numpy() - ref)
    assert (err <= 4e-7 * np.maximum(1.0, np.abs(xd))).all(), (err.max(), xd.flat[err.argmax()])
    ones = torch.ones(M, N, device="cuda")
    gy = K.gemm(ones, eye, act=K.EPI_MUL_GELU_GRAD, aux=x)            # 1 * gelu'(x)
    refg = 0.5 * (1.0 + erf(xd / np.sqrt(2.0))) + xd * np.exp(-0.5 * xd * xd) / np.sqrt(2.0 * np.pi)
    errg = np.abs(gy.double().cpu().numpy() - refg)
    assert errg.max() <= 4e-7, (errg.max(), xd.flat[errg.argmax()])
    yl = y.cpu().view(-1)
    assert yl[-8].item() == 100.0 and yl[-7].item() == 0.0 and yl[-10].item() == 20.0 and abs(yl[-9].item()) < 1e-20
    assert torch.isfinite(y).all() and torch.isfinite(gy).all()


@pytest.mark.parametrize("T,D,eps", [(1792, 384, 1e-5), (300, 768, 1e-6), (7, 64, 1e-5), (33, 128, 1e-5), (5, 2048, 1e-5)])
def test_layernorm_fwd_bwd(K, T, D, eps):
    x = _rnd(f"ln.x{T}", T, D); pos = _rnd(f"ln.p{T}", T, D) * 0.3
    g = 1 + 0.1 * _rnd(f"ln.g{D}", D); b = 0.1 * _rnd(f"ln.b{D}", D)
    dy = _rnd(f"ln.dy{T}", T, D); dres = _rnd(f"ln.dr{T}", T, D)
    xd = (x + pos).double().requires_grad_(True); gd = g.double().requires_grad_(True); bd = b.double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xd, (D,), gd, bd, eps)
    (yr * dy.double()).sum().backward()
    y, xin, mean, rstd = K.layernorm_fwd(x.cuda(), pos.cuda(), g.cuda(), b.cuda(), eps)
    assert _rel(y, yr) <= 2e-5 and _rel(xin, x + pos) <= 1e-6
    dx, dg, db = K.layernorm_bwd(dy.cuda(), xin, g.cuda(), mean, rstd, dres=dres.cuda())
    assert _rel(dx, xd.grad + dres.double()) <= 5e-5
    assert _rel(dg, gd.grad) <= 5e-5 and _rel(db, bd.grad) <= 5e-5
    # autograd wrapper without pos
    xt = x.cuda().requires_grad_(True); gt = g.cuda().requires_grad_(True); bt = b.cuda().requires_grad_(True)
    (K.layer_norm(xt, gt, bt, eps) * dy.cuda()).sum().backward()
    x2 = x.double().requires_grad_(True)
    (torch.nn.functional.layer_norm(x2, (D,), g.double(), b.double(), eps) * dy.double()).sum().backward()
    assert _rel(xt.grad, x2.grad) <= 5e-5


def _attn_ref(qkv, B, S, H, hd):
    q, k, v = qkv.double().view(B, S, 3, H, hd).permute(2, 0, 3, 1, 4)
    a = torch.softmax((q @ k.transpose(-2, -1)) * hd ** -0.5, dim=-1)
    return (a @ v).transpose(1, 2).reshape(B * S, H * hd)


@pytest.mark.parametrize("B,S,H,hd", [(3, 14, 6, 64), (2, 64, 6, 64), (2, 128, 12, 64), (5, 33, 2, 64), (2, 100, 3, 64), (9, 1, 2, 64),
                                      (2, 16, 2, 32), (3, 128, 2, 32), (128, 14, 6, 64), (1, 512, 3, 64), (2, 576, 2, 64), (2, 129, 2, 64), (1, 300, 2, 32)])
def test_attention_forward(K, B, S, H, hd):
    qkv = _rnd(f"at{B}{S}{H}", B * S, 3 * H * hd)
    out, lse = K.attention_fwd(qkv.cuda(), B, S, H, hd)
    assert _rel(out, _attn_ref(qkv, B, S, H, hd)) <= 2e-5
    q, k, _ = qkv.double().view(B, S, 3, H, hd).permute(2, 0, 3, 1, 4)
    assert _rel(lse, torch.logsumexp((q @ k.transpose(-2, -1)) * hd ** -0.5, dim=-1)) <= 2e-5


def test_attention_forward_large_logits(K):
    """softmax must be max-subtracted: scores of a few hundred must not overflow."""
    B, S, H, hd = 2, 64, 2, 64
    qkv = _rnd("atbig", B * S, 3 * H * hd) * 6.0
    out, _ = K.attention_fwd(qkv.cuda(), B, S, H, hd)
    assert torch.isfinite(out).all() and _rel(out, _attn_ref(qkv, B, S, H, hd)) <= 5e-5


@pytest.mark.parametrize("B,S,H,hd", [(3, 14, 6, 64), (2, 64, 6, 64), (4, 33, 2, 64), (2, 16, 2, 32), (2, 1, 1, 64), (2, 128, 3, 64),
                                      (2, 100, 2, 64), (1, 103, 2, 64), (2, 128, 2, 32), (1, 512, 2, 64), (2, 300, 1, 64), (1, 129, 2, 32)])
def test_attention_backward(K, B, S, H, hd):
    qkv = _rnd(f"ab{B}{S}{H}", B * S, 3 * H * hd); do = _rnd(f"abd{B}{S}{H}", B * S, H * hd)
    qd = qkv.double().requires_grad_(True)
    (_attn_ref(qd, B, S, H, hd) * do.double()).sum().backward()
    qg = qkv.cuda()
    out, lse = K.attention_fwd(qg, B, S, H, hd)
    dqkv = K.attention_bwd(qg, out, do.cuda(), lse, B, S, H, hd)
    assert _rel(dqkv, qd.grad) <= 5e-5


def _load_block(blk_oracle):
    sd = blk_oracle.state_dict()
    g = lambda k: sd[k].cuda() if k in sd else None
    return dict(n1w=g("norm1.weight"), n1b=g("norm1.bias"), wqkv=g("attn.qkv.weight"), bqkv=g("attn.qkv.bias"),
                wproj=g("attn.proj.weight"), bproj=g("attn.proj.bias"), n2w=g("norm2.weight"), n2b=g("norm2.bias"),
                w1=g("mlp.fc1.weight"), b1=g("mlp.fc1.bias"), w2=g("mlp.fc2.weight"), b2=g("mlp.fc2.bias"))


def _run_block(K, p, x, pos, g1, g2, heads, eps):
    names = ["n1w", "n1b", "wqkv", "bqkv", "wproj", "bproj", "n2w", "n2b", "w1", "b1", "w2", "b2"]
    return K.BlockFn.apply(x, pos, g1, g2, *[p[n] for n in names], heads, eps, True)


def test_block_against_golden_and_oracle(K):
    from oracle import layers as L
    g = golden("g3_block")
    blk = fill_module(L.Block(384, 6), "g3.blk.")
    p = _load_block(blk)
    for v in p.values():
        if v is not None:
            v.requires_grad_(True)
    x = fill_tensor("g3.x", (2, 14, 384), "code").cuda().requires_grad_(True)
    y = _run_block(K, p, x, None, None, None, 6, 1e-5)
    assert _rel(y, torch.from_numpy(g["y"])) <= TOL
    (y * fill_tensor("g3.w", (2, 14, 384), "code").cuda()).sum().backward()
    assert _rel(x.grad, torch.from_numpy(g["dx"])) <= TOL
    name_map = {"norm1.weight": "n1w", "norm1.bias": "n1b", "attn.qkv.weight": "wqkv", "attn.proj.weight": "wproj",
                "attn.proj.bias": "bproj", "norm2.weight": "n2w", "norm2.bias": "n2b", "mlp.fc1.weight": "w1",
                "mlp.fc1.bias": "b1", "mlp.fc2.weight": "w2", "mlp.fc2.bias": "b2"}
    for n, v in zip(g["grad_names"], g["grad_norms"]):
        got = p[name_map[str(n)]].grad.norm().item()
        assert abs(got - v) <= TOL * max(1.0, v), (n, got, v)
    # teacher-style block: qkv bias, eps 1e-6, S=128
    blk_t = fill_module(L.Block(128, 2, qkv_bias=True, eps=1e-6), "g3.blkt.")
    yt = _run_block(K, _load_block(blk_t), fill_tensor("g3.xt", (2, 128, 128), "code").cuda(), None, None, None, 2, 1e-6)
    assert _rel(yt, torch.from_numpy(golden("g3_block_teacher")["y"])) <= TOL


def test_block_with_pos_and_droppath_vs_oracle(K):
    from oracle import layers as L
    torch.manual_seed(0)
    B, S, D, H = 6, 64, 128, 2
    blk = fill_module(L.Block(D, H, drop_path=0.25, tag="t"), "blkdp.").train()
    x = _rnd("bd.x", B, S, D); pos = _rnd("bd.p", B, S, D) * 0.2; w = _rnd("bd.w", B, S, D)
    draws = L.Draws(record=True)
    xo = x.clone().requires_grad_(True); po = pos.clone().requires_grad_(True)
    yo = blk(xo + po, draws)
    (yo * w).sum().backward()
    keep = 0.75
    g1 = (torch.floor(keep + draws.table["t.attn"]) / keep).cuda(); g2 = (torch.floor(keep + draws.table["t.mlp"]) / keep).cuda()
    assert 0 < (g1 == 0).sum() + (g2 == 0).sum() < 2 * B            # both outcomes exercised
    p = _load_block(blk)
    for v in p.values():
        if v is not None:
            v.requires_grad_(True)
    xg = x.cuda().requires_grad_(True); pg = pos.cuda().requires_grad_(True)
    y = _run_block(K, p, xg, pg, g1, g2, H, 1e-5)
    assert _rel(y, yo) <= TOL
    (y * w.cuda()).sum().backward()
    assert _rel(xg.grad, xo.grad) <= TOL and _rel(pg.grad, po.grad) <= TOL
    od = dict(blk.named_parameters())
    for on, pn in [("attn.qkv.weight", "wqkv"), ("mlp.fc1.weight", "w1"), ("mlp.fc2.bias", "b2"), ("norm1.weight", "n1w"),
                   ("attn.proj.bias", "bproj"), ("norm2.bias", "n2b")]:
        assert _rel(p[pn].grad, od[on].grad) <= TOL, on


def test_mlp_linear_autograd(K):
    x = _rnd("ml.x", 50, 3); w1 = _rnd("ml.w1", 128, 3); b1 = _rnd("ml.b1", 128) * 0.1
    w2 = _rnd("ml.w2", 384, 128) * 0.1; b2 = _rnd("ml.b2", 384) * 0.1; dy = _rnd("ml.dy", 50, 384)
    ts = [t.double().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    ref = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(ts[0], ts[1], ts[2])), ts[3], ts[4])
    (ref * dy.double()).sum().backward()
    tg = [t.cuda().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    y = K.mlp(*tg)
    assert _rel(y, ref) <= 2e-5
    (y * dy.cuda()).sum().backward()
    for a, b in zip(tg, ts):
        assert _rel(a.grad, b.grad) <= 5e-5
    # plain linear on a 3-D input
    x3 = _rnd("ml.x3", 4, 7, 128).cuda().requires_grad_(True); wl = w2.cuda().requires_grad_(True)
    yl = K.linear(x3, wl, None)
    assert _rel(yl, x3.detach().double().cpu() @ w2.double().t()) <= 2e-5
    yl.sum().backward()
    assert _rel(wl.grad, x3.detach().double().cpu().reshape(-1, 128).sum(0)[None, :].expand(384, -1)) <= 5e-5


def test_cosine_loss(K):
    s = fill_tensor("g6.s", (4, 51, 384), "code"); t = fill_tensor("g6.t", (4, 51, 384), "code")
    sg = s.cuda().requires_grad_(True)
    loss = K.cosine_distill_loss(sg, t.cuda())
    assert abs(loss.item() - float(golden("g6_cosine")["loss"])) <= 1e-5
    (loss * 3.0).backward()
    sd = s.double().requires_grad_(True)
    (3.0 * (1 - torch.nn.functional.cosine_similarity(sd, t.double(), dim=-1, eps=1e-8)).mean()).backward()
    assert _rel(sg.grad, sd.grad) <= 1e-5


# ---------------------------------------------------------------------------------- mini-PointNet / DGCNN glue kernels
@pytest.mark.parametrize("R,C", [(4096, 128), (32 * 77, 512), (640, 64)])
def test_batchnorm_relu_fwd_bwd(K, R, C):
    x = _rnd(f"bn.x{R}", R, C) * 1.5 + 0.7; dy = _rnd(f"bn.dy{R}", R, C)
    bn_ref = torch.nn.BatchNorm1d(C).double(); bn = torch.nn.BatchNorm1d(C).cuda()
    with torch.no_grad():
        w = 1 + 0.2 * _rnd(f"bn.w{C}", C); b = 0.1 * _rnd(f"bn.b{C}", C)
        bn_ref.weight.copy_(w); bn_ref.bias.copy_(b); bn.weight.copy_(w); bn.bias.copy_(b)
    xd = x.double().requires_grad_(True)
    yr = torch.relu(bn_ref(xd)); (yr * dy.double()).sum().backward()
    xg = x.cuda().requires_grad_(True)
    y = K.batch_norm_act(xg, bn, True, relu=True); (y * dy.cuda()).sum().backward()
    assert _rel(y, yr) <= 2e-5 and _rel(xg.grad, xd.grad) <= 5e-5
    assert _rel(bn.weight.grad, bn_ref.weight.grad) <= 5e-5 and _rel(bn.bias.grad, bn_ref.bias.grad) <= 5e-5
    assert _rel(bn.running_mean, bn_ref.running_mean) <= 1e-5 and _rel(bn.running_var, bn_ref.running_var) <= 1e-5
    assert int(bn.num_batches_tracked) == 1
    bn.eval(); bn_ref.eval()
    assert _rel(K.batch_norm_act(x.cuda(), bn, False, relu=False), bn_ref(x.double())) <= 2e-5


def test_group_max_and_group_add(K):
    G, n, C = 50, 32, 256
    x = _rnd("gm.x", G * n, C); dy = _rnd("gm.dy", G, C)
    x[0:n, 3] = 1.25                                            # exact tie inside a group -> first row wins (torch.max)
    xd = x.double().requires_grad_(True)
    ref = xd.view(G, n, C).max(dim=1)[0]; (ref * dy.double()).sum().backward()
    xg = x.cuda().requires_grad_(True)
    out = K.group_max(xg, n); (out * dy.cuda()).sum().backward()
    assert _rel(out, ref) == 0 and _rel(xg.grad, xd.grad) <= 1e-6
    # y = x w^T + g[row // n]
    w = _rnd("ga.w", 96, C) * 0.1; g = _rnd("ga.g", G, 96); d2 = _rnd("ga.d", G * n, 96)
    ts = [t.double().requires_grad_(True) for t in (x, w, g)]
    r2 = (ts[0] @ ts[1].t()).view(G, n, 96) + ts[2].unsqueeze(1); (r2.reshape(G * n, 96) * d2.double()).sum().backward()
    tg = [t.cuda().requires_grad_(True) for t in (x, w, g)]
    y2 = K.linear_group_add(tg[0], tg[1], tg[2], n); (y2 * d2.cuda()).sum().backward()
    assert _rel(y2, r2.reshape(G * n, 96)) <= 2e-5
    for a, b in zip(tg, ts):
        assert _rel(a.grad, b.grad) <= 5e-5


@pytest.mark.parametrize("B,G,k,C", [(3, 64, 4, 256), (2, 64, 4, 1024), (5, 64, 4, 512), (2, 16, 3, 64), (1, 512, 4, 256)])
def test_dgcnn_edge_tail_and_head(K, B, G, k, C):
    """(…, 64, 4, 1024): the largest slice that takes the single-kernel LDS path (132 KB); (1, 512, …): too large, three-kernel path."""
    import torch.nn.functional as F
    yz = _rnd("eg.yz", B * G, 2 * C); gn = torch.nn.GroupNorm(4, C).cuda()
    with torch.no_grad():
        gn.weight.copy_(_rnd("eg.w", C)); gn.bias.copy_(0.1 * _rnd("eg.b", C))     # negative gammas exercise the min branch
    idx = torch.stack([torch.stack([torch.randperm(G, generator=torch.Generator().manual_seed(b * 10 + j)) for j in range(k)]) for b in range(B)])
    y = yz[:, :C].reshape(B, G, C).double(); z = yz[:, C:].reshape(B, 1, G, C).double()
    pre = (y[torch.arange(B).view(B, 1, 1), idx] + z).permute(0, 3, 2, 1)
    ref = F.leaky_relu(F.group_norm(pre, 4, gn.weight.double().cpu(), gn.bias.double().cpu(), gn.eps), 0.2).max(dim=-1)[0]
    ref = ref.transpose(1, 2).reshape(B * G, C)
    buf = torch.zeros(B * G, C + 40, device="cuda")
    out = K.edge_gn_lrelu_max(yz.cuda(), C, idx.cuda(), B, G, k, C, gn, out=buf, ooff=40)
    assert _rel(out[:, 40:], ref) <= 2e-5 and (out[:, :40] == 0).all()
    saved = os.environ.get("ACT_EDGE_GN_FUSE")              # (read once per process: this only documents which path ran)
    assert saved is None
    # head: GroupNorm + LeakyReLU only
    h = _rnd("eg.h", B * G, C) * 2 + 0.3
    href = F.leaky_relu(F.group_norm(h.double().view(B, G, C).transpose(1, 2), 4, gn.weight.double().cpu(), gn.bias.double().cpu(), gn.eps), 0.2)
    assert _rel(K.edge_gn_lrelu_max(h.cuda(), -1, None, B, G, 1, C, gn), href.transpose(1, 2).reshape(B * G, C)) <= 2e-5


@pytest.mark.parametrize("B,G,k,C", [(3, 64, 4, 256), (2, 16, 4, 32), (2, 200, 3, 64), (1, 512, 4, 64)])
def test_dgcnn_edge_tail_backward(K, B, G, k, C):
    """HIP backward of the edge-conv tail (gather + GroupNorm + LeakyReLU + max over neighbours) and of the k = 1 head
    against float64 autograd through the reference formulation (models/dvae.py:59-117)."""
    import torch.nn.functional as F
    yz = _rnd(f"eb.yz{G}{C}", B * G, 2 * C); do = _rnd(f"eb.do{G}{C}", B * G, C)
    gw = _rnd(f"eb.w{C}", C); gb = 0.1 * _rnd(f"eb.b{C}", C)                  # negative gammas exercise the min branch
    gen = torch.Generator().manual_seed(G * 7 + C)
    idx = torch.stack([torch.stack([torch.randint(0, G, (G,), generator=gen) for j in range(k)]) for b in range(B)])
    idx[:, 0] = torch.arange(G)                                                # k-NN graphs contain the point itself
    yd = yz.double().requires_grad_(True); wd = gw.double().requires_grad_(True); bd = gb.double().requires_grad_(True)
    y = yd[:, :C].reshape(B, G, C); z = yd[:, C:].reshape(B, 1, G, C)
    pre = (y[torch.arange(B).view(B, 1, 1), idx] + z).permute(0, 3, 2, 1)
    ref = F.leaky_relu(F.group_norm(pre, 4, wd, bd, 1e-5), 0.2).max(dim=-1)[0].transpose(1, 2).reshape(B * G, C)
    (ref * do.double()).sum().backward()
    gn = torch.nn.GroupNorm(4, C).cuda()
    with torch.no_grad():
        gn.weight.copy_(gw); gn.bias.copy_(gb)
    yg = yz.cuda().requires_grad_(True)
    out = K.edge_gn_lrelu_max_train(yg, C, idx.cuda(), B, G, k, C, gn)
    (out * do.cuda()).sum().backward()
    assert _rel(out, ref) <= 2e-5
    assert _rel(yg.grad, yd.grad) <= 5e-5
    assert _rel(gn.weight.grad, wd.grad) <= 5e-5 and _rel(gn.bias.grad, bd.grad) <= 5e-5
    # head: GroupNorm + LeakyReLU only
    h = _rnd(f"eb.h{G}{C}", B * G, C) * 2 + 0.3
    hd = h.double().requires_grad_(True); wd2 = gw.double().requires_grad_(True); bd2 = gb.double().requires_grad_(True)
    href = F.leaky_relu(F.group_norm(hd.view(B, G, C).transpose(1, 2), 4, wd2, bd2, 1e-5), 0.2).transpose(1, 2).reshape(B * G, C)
    (href * do.double()).sum().backward()
    gn.zero_grad()
    hg = h.cuda().requires_grad_(True)
    o2 = K.edge_gn_lrelu_max_train(hg, -1, None, B, G, 1, C, gn)
    (o2 * do.cuda()).sum().backward()
    assert _rel(o2, href) <= 2e-5 and _rel(hg.grad, hd.grad) <= 5e-5
    assert _rel(gn.weight.grad, wd2.grad) <= 5e-5 and _rel(gn.bias.grad, bd2.grad) <= 5e-5


@pytest.mark.parametrize("B,G,k,C,with_z", [(3, 64, 4, 256, True), (2, 16, 4, 32, True), (2, 128, 3, 96, True), (4, 64, 4, 1024, True), (2, 64, 4, 128, False)])
def test_dgcnn_edge_tail_backward_lds_form_matches_the_gather_scatter_kernels(K, B, G, k, C, with_z):
    """round 6: the LDS-resident passes of the edge-conv tail backward (slabs of Y / Z staged once, gathers from LDS, dY as a gather over an inverse adjacency
    that lists the incoming edges in the order the scatter images were filled) against the global-gather / scatter-image kernels they replace at G <= 128:
    same terms in the same order (reported when bit-identical), two runs of the new form are bit-identical (deterministic, no atomics)."""
    yz = _rnd(f"el.yz{G}{C}", B * G, 2 * C if with_z else C).cuda(); do = _rnd(f"el.do{G}{C}", B * G, C).cuda()
    gw = _rnd(f"el.w{C}", C); gb = 0.1 * _rnd(f"el.b{C}", C)
    gen = torch.Generator().manual_seed(G * 11 + C)
    idx = torch.stack([torch.stack([torch.randint(0, G, (G,), generator=gen) for j in range(k)]) for b in range(B)])
    idx[:, 0] = torch.arange(G)
    idx[0, 1:, :] = 3                                              # a hub: one row collects (k - 1) G edges of sample 0
    idx = idx.cuda()
    gn = torch.nn.GroupNorm(4, C).cuda()
    with torch.no_grad():
        gn.weight.copy_(gw); gn.bias.copy_(gb)
    res = []
    prev = K.lib.act_edge_bwd_lds(-1)
    try:
        for on in (0, 1, 1):
            K.lib.act_edge_bwd_lds(on)
            gn.zero_grad()
            yg = yz.clone().requires_grad_(True)
            out = K.edge_gn_lrelu_max_train(yg, C if with_z else -1, idx, B, G, k, C, gn)
            (out * do).sum().backward()
            torch.cuda.synchronize()
            res.append((yg.grad.clone(), gn.weight.grad.clone(), gn.bias.grad.clone()))
    finally:
        K.lib.act_edge_bwd_lds(prev)
    for a, b in zip(res[1], res[2]):
        assert torch.equal(a, b)
    same = all(torch.equal(a, b) for a, b in zip(res[0], res[1]))
    print(f"[edge bwd lds] B={B} G={G} k={k} C={C} z={with_z}: bit-identical to the gather / scatter kernels: {same}")
    for a, b in zip(res[0], res[1]):
        assert _rel(b, a) <= 2e-6
    assert res[1][0].abs().max() > 0


@pytest.mark.parametrize("B,G,C,tau", [(2, 16, 64, 0.7), (3, 8, 8192, 1.0), (2, 5, 1000, 0.0625)])
def test_soft_gumbel_softmax_and_kl_to_uniform(K, B, G, C, tau):
    """Stage-I tokenizer: F.gumbel_softmax(hard=False) with injected noise and the KL(mean softmax || uniform) term
    (models/dvae.py:600, 470-476), forward and backward against float64 autograd."""
    import torch.nn.functional as F
    logits = _rnd(f"gs.l{C}", B, G, C) * 2.0
    torch.manual_seed(C)
    noise = -torch.empty(B, G, C).exponential_().log()
    dy = _rnd(f"gs.d{C}", B, G, C)
    ld = logits.double().requires_grad_(True)
    yr = F.softmax((ld + noise.double()) / tau, dim=-1)
    (yr * dy.double()).sum().backward()
    lg = logits.cuda().requires_grad_(True)
    y = K.gumbel_softmax(lg, tau, noise=noise.cuda())
    (y * dy.cuda()).sum().backward()
    assert _rel(y, yr) <= 2e-5 and _rel(lg.grad, ld.grad) <= 5e-5
    # in-kernel Philox noise: rows are proper distributions, deterministic per seed
    y1 = K.gumbel_softmax(logits.cuda(), tau, seed=11); y2 = K.gumbel_softmax(logits.cuda(), tau, seed=11); y3 = K.gumbel_softmax(logits.cuda(), tau, seed=12)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3) and (y1.sum(-1) - 1).abs().max() < 1e-4 and (y1 >= 0).all()
    # KL term
    ld2 = logits.double().requires_grad_(True)
    lq = torch.log(F.softmax(ld2, dim=-1).mean(dim=1))
    lu = torch.log(torch.tensor([1.0 / C], dtype=torch.float64)).expand(B, C)
    ref = F.kl_div(lq, lu, None, None, 'batchmean', log_target=True)
    (ref * 1.7).backward()
    lg2 = logits.cuda().requires_grad_(True)
    kl = K.kl_to_uniform(lg2)
    (kl * 1.7).backward()
    assert abs(kl.item() - ref.item()) <= 2e-5 * max(1.0, abs(ref.item()))
    assert _rel(lg2.grad, ld2.grad) <= 5e-5


def test_in_kernel_gumbel_noise_is_finite_at_full_batch(K):
    """Regression: 24 random bits + 1/2 rounds its largest value up to u == 1 in fp32 -> -log(-log(1)) = +inf once per 2^24 draws, i.e. a NaN
    row in the soft gumbel-softmax of (almost) every Stage-I step at the benchmark batch (128 x 64 x 8192 = 2^26 draws per call).  The noise
    must be finite for every one of the 2^32 bit patterns; here: five full-batch calls (3.4e8 draws) and the extreme bit patterns' images."""
    logits = torch.zeros(128, 64, 8192, device="cuda")
    for seed in (1, 2, 3, 12345, 2 ** 61 + 7):
        y = K.gumbel_softmax(logits, 1.0, seed=seed)
        assert torch.isfinite(y).all(), seed
        assert (y.sum(-1) - 1).abs().max() < 1e-4
        # softmax of pure gumbel noise: y = e^g / sum e^g with e^g = 1/E, E ~ Exp(1); an infinite g would show up as y == 1 exactly
        assert y.max() < 1.0
    del y, logits


def test_prompt_layernorm_fused_dropout(K):
    """LN(dropout(tok) + ppos): exact against the oracle formula for p = 0; for p = 0.1 the rows are valid LayerNorm outputs of
    a mask with the right drop rate, deterministic per seed and different per cloud."""
    import torch.nn.functional as F
    B, P, D = 6, 64, 768
    tok = _rnd("pl.t", P, D); ppos = _rnd("pl.p", P, D) * 0.3; g = 1 + 0.2 * _rnd("pl.g", D); b = 0.1 * _rnd("pl.b", D)
    ref = F.layer_norm((tok + ppos).double(), (D,), g.double(), b.double(), 1e-6)
    y0 = K.prompt_layernorm(tok.cuda(), ppos.cuda(), B, 0.0, 5, g.cuda(), b.cuda(), 1e-6).view(B, P, D)
    for bb in range(B):
        assert _rel(y0[bb], ref) <= 2e-5
    y1 = K.prompt_layernorm(tok.cuda(), ppos.cuda(), B, 0.1, 5, g.cuda(), b.cuda(), 1e-6).view(B, P, D)
    y2 = K.prompt_layernorm(tok.cuda(), ppos.cuda(), B, 0.1, 5, g.cuda(), b.cuda(), 1e-6).view(B, P, D)
    y3 = K.prompt_layernorm(tok.cuda(), ppos.cuda(), B, 0.1, 6, g.cuda(), b.cuda(), 1e-6).view(B, P, D)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3) and not torch.equal(y1[0], y1[1])
    # invert the affine LayerNorm map per row to recover which elements were dropped: v = tok*keep/0.9 + ppos
    xhat = (y1.double().cpu() - b.double()) / g.double()                        # = (v - mean) * rstd
    # a dropped element has v == ppos; solve for (mean, rstd) per row by least squares on the two candidate values
    cand_keep = (tok / 0.9 + ppos).double(); cand_drop = ppos.double()
    frac = []
    for bb in range(2):
        for pp in range(4):
            xr = xhat[bb, pp]
            best = None
            # rows are long (768): fit v = xr / rstd + mean using the assumption that most elements are kept
            A = torch.stack((xr, torch.ones_like(xr)), 1)
            sol = torch.linalg.lstsq(A, cand_keep[pp].unsqueeze(1)).solution.squeeze(1)
            for _ in range(5):
                v = xr * sol[0] + sol[1]
                kept = (v - cand_keep[pp]).abs() < (v - cand_drop[pp]).abs()
                target = torch.where(kept, cand_keep[pp], cand_drop[pp])
                sol = torch.linalg.lstsq(A, target.unsqueeze(1)).solution.squeeze(1)
            v = xr * sol[0] + sol[1]
            err = torch.minimum((v - cand_keep[pp]).abs(), (v - cand_drop[pp]).abs()).max().item()
            assert err < 1e-3, err                                               # every element is one of the two admissible values
            frac.append(1.0 - kept.float().mean().item())
    assert 0.06 < sum(frac) / len(frac) < 0.14                                   # drop rate 0.1


def test_gumbel_argmax_codebook_fused(K):
    import torch.nn.functional as F
    B, G, C, D = 4, 16, 512, 48
    h = _rnd("gu.h", B * G, C); cb = _rnd("gu.cb", C, D); gn = torch.nn.GroupNorm(4, C).cuda()
    with torch.no_grad():
        gn.weight.copy_(1 + 0.3 * _rnd("gu.w", C)); gn.bias.copy_(0.1 * _rnd("gu.b", C))
    torch.manual_seed(5)
    noise = -torch.empty(B, G, C).exponential_().log()
    logits = F.leaky_relu(F.group_norm(h.view(B, G, C).transpose(1, 2), 4, gn.weight.cpu(), gn.bias.cpu(), gn.eps), 0.2).transpose(1, 2)
    ref_idx = (logits + noise).argmax(-1)
    out, idx, lg = K.gn_gumbel_argmax_gather(h.cuda(), B, G, gn, cb.cuda(), noise=noise.cuda(), want_logits=True)
    assert _rel(lg, logits) <= 2e-5
    assert torch.equal(idx.cpu(), ref_idx) and torch.equal(out.cpu(), cb[ref_idx])
    # device RNG path: deterministic per seed, different across seeds, roughly gumbel-distributed choices
    o1, i1, _ = K.gn_gumbel_argmax_gather(h.cuda(), B, G, gn, cb.cuda(), seed=123)
    o2, i2, _ = K.gn_gumbel_argmax_gather(h.cuda(), B, G, gn, cb.cuda(), seed=123)
    o3, i3, _ = K.gn_gumbel_argmax_gather(h.cuda(), B, G, gn, cb.cuda(), seed=124)
    assert torch.equal(i1, i2) and not torch.equal(i1, i3)
    flat = torch.zeros(1, C, device="cuda"); gn1 = torch.nn.GroupNorm(4, C).cuda()
    counts = torch.zeros(C)
    big = torch.zeros(4096, C, device="cuda")                      # equal logits -> argmax of pure gumbel noise is uniform
    _, ib, _ = K.gn_gumbel_argmax_gather(big, 64, 64, gn1, cb.cuda(), seed=7)
    counts = torch.bincount(ib.flatten().cpu(), minlength=C).float()
    assert counts.max() <= 30 and (counts > 0).sum() >= 0.95 * C           # mean 8 per bin


def test_attention_prefix_and_teacher_block_prefix(K):
    """prompt tokens as keys/values only == the patch-token rows of the full block on cat(prompt, x)."""
    from oracle import layers as L
    B, P, G, D, H = 3, 64, 64, 128, 2
    blk = fill_module(L.Block(D, H, qkv_bias=True, eps=1e-6), "pfx.")
    x = _rnd("pf.x", B, G, D); pos = _rnd("pf.p", B, G, D) * 0.2; prm = _rnd("pf.m", B, P, D) * 0.3; ppos = _rnd("pf.q", 1, P, D) * 0.2
    full = blk(torch.cat((prm, x), 1) + torch.cat((ppos.expand(B, -1, -1), pos), 1), L.Draws())[:, P:]
    p = _load_block(blk)
    names = ["n1w", "n1b", "wqkv", "bqkv", "wproj", "bproj", "n2w", "n2b", "w1", "b1", "w2", "b2"]
    with torch.no_grad():
        y = K.block_forward_prefix(x.cuda().reshape(B * G, D), pos.cuda().reshape(B * G, D), (prm + ppos).cuda().reshape(B * P, D),
                                   B, P, G, *[p[n] for n in names], H, 1e-6)
    assert _rel(y.reshape(B, G, D), full) <= TOL
    # raw kernel, odd sizes: S0=20 prompts, Sq=33 queries, hd=64
    B, S0, Sq, H, hd = 2, 20, 33, 3, 64
    kv0 = _rnd("pf.kv", B * S0, 2 * H * hd); qkv = _rnd("pf.qkv", B * Sq, 3 * H * hd)
    out = K.attention_fwd_prefix(kv0.cuda(), S0, qkv.cuda(), Sq, B, H, hd)
    q, k1, v1 = qkv.double().view(B, Sq, 3, H, hd).permute(2, 0, 3, 1, 4)
    k0, v0 = kv0.double().view(B, S0, 2, H, hd).permute(2, 0, 3, 1, 4)
    kk = torch.cat((k0, k1), 2); vv = torch.cat((v0, v1), 2)
    ref = (torch.softmax(q @ kk.transpose(-2, -1) * hd ** -0.5, -1) @ vv).transpose(1, 2).reshape(B * Sq, H * hd)
    assert _rel(out, ref) <= 2e-5


@pytest.mark.parametrize("B,S0,Sq,H,hd", [(2, 20, 33, 3, 64), (2, 64, 64, 12, 64), (3, 8, 16, 2, 32), (1, 100, 70, 2, 64),
                                           (2, 0, 40, 2, 64), (1, 130, 129, 1, 32),
                                           # the register-resident kernels (S0 % 32 == 0, Sq >= 32): shifted tail tiles on either side, both head dims
                                           (2, 64, 40, 2, 64), (2, 32, 33, 2, 64), (1, 64, 104, 3, 32), (1, 64, 512, 2, 64), (3, 96, 32, 1, 64), (5, 0, 63, 2, 32),
                                           # single-pass backward (round 6): two pairs per workgroup with an odd pair count, one key tile, several key blocks with a ragged last one
                                           (3, 0, 64, 1, 64), (1, 32, 32, 3, 64), (3, 0, 32, 1, 32), (1, 128, 200, 1, 64), (2, 64, 64, 6, 32)])
def test_attention_prefix_backward(K, B, S0, Sq, H, hd):
    """dQ / dK / dV of the own rows and dK / dV of the prefix rows against a float64 reference."""
    kv0 = _rnd(f"pb.kv{S0}{Sq}", B * max(S0, 1), 2 * H * hd)[:B * S0]; qkv = _rnd(f"pb.qkv{S0}{Sq}", B * Sq, 3 * H * hd)
    do = _rnd(f"pb.do{S0}{Sq}", B * Sq, H * hd)
    kd = kv0.double().requires_grad_(True); qd = qkv.double().requires_grad_(True)
    q, k1, v1 = qd.view(B, Sq, 3, H, hd).permute(2, 0, 3, 1, 4)
    k0, v0 = kd.view(B, S0, 2, H, hd).permute(2, 0, 3, 1, 4)
    kk = torch.cat((k0, k1), 2); vv = torch.cat((v0, v1), 2)
    ref = (torch.softmax(q @ kk.transpose(-2, -1) * hd ** -0.5, -1) @ vv).transpose(1, 2).reshape(B * Sq, H * hd)
    (ref * do.double()).sum().backward()
    kg, qg = kv0.cuda(), qkv.cuda()
    if S0 == 0:
        kg = torch.zeros(1, 2 * H * hd, device="cuda")[:0]
    out, lse = K.attention_fwd_prefix(kg if S0 else torch.zeros(1, device="cuda"), S0, qg, Sq, B, H, hd, want_lse=True)
    assert _rel(out, ref) <= 2e-5
    dkv0, dqkv = K.attention_bwd_prefix(kg, S0, qg, Sq, out, do.cuda(), lse, B, H, hd)
    assert _rel(dqkv, qd.grad) <= 5e-5
    if S0:
        assert _rel(dkv0, kd.grad) <= 5e-5


def test_prefix_block_training_matches_full_block_autograd(K):
    """PrefixBlockFn (prompts as keys/values only) == autograd through the oracle block on cat(prompt, x), patch rows only:
    output and the gradients w.r.t. the patch tokens, their positions and the prompts."""
    from oracle import layers as L
    B, P, G, D, H = 2, 24, 40, 128, 2
    blk = fill_module(L.Block(D, H, qkv_bias=True, eps=1e-6), "pfx.")
    x = _rnd("pt.x", B, G, D).requires_grad_(True); pos = (_rnd("pt.p", B, G, D) * 0.2).requires_grad_(True)
    prm = (_rnd("pt.m", B, P, D) * 0.3).requires_grad_(True)
    w = _rnd("pt.w", B, G, D)
    full = blk(torch.cat((prm, x + pos), 1), L.Draws())[:, P:]
    (full * w).sum().backward()
    p = _load_block(blk)
    names = ["n1w", "n1b", "wqkv", "bqkv", "wproj", "bproj", "n2w", "n2b", "w1", "b1", "w2", "b2"]
    xg = x.detach().cuda().reshape(B * G, D).requires_grad_(True); pg = pos.detach().cuda().reshape(B * G, D).requires_grad_(True)
    mg = prm.detach().cuda().reshape(B * P, D).requires_grad_(True)
    y = K.PrefixBlockFn.apply(xg, pg, mg, B, P, G, *[p[n] for n in names], H, 1e-6)
    (y * w.cuda().reshape(B * G, D)).sum().backward()
    assert _rel(y.reshape(B, G, D), full) <= TOL
    assert _rel(xg.grad.reshape(B, G, D), x.grad) <= TOL and _rel(pg.grad.reshape(B, G, D), pos.grad) <= TOL
    assert _rel(mg.grad.reshape(B, P, D), prm.grad) <= TOL


@pytest.mark.parametrize("tile", [7, 8, 9, 10, 11, 12])
def test_gemm_m_tail_on_the_16x16x4_kernels(K, tile):
    """token counts that are not a multiple of the tile height (32 clouds x 65 tokens = 2080 rows) stay on the fast kernels:
    A rows clamped on load, C rows guarded on store, with and without split-K and a fused epilogue."""
    for M in (2080, 200, 65):
        N, Kd = 384, 768
        a = _rnd(f"mt.a{M}", M, Kd); b = _rnd("mt.b", N, Kd); bias = _rnd("mt.bias", N); res = _rnd(f"mt.r{M}", M, N)
        ref = a.double() @ b.double().t()
        for sp in (1, 3):
            c = K.gemm(a.cuda(), b.cuda(), True, True, cfg=(tile, sp))
            assert c.shape == (M, N) and _rel(c, ref) <= 2e-5, (tile, M, sp)
            if tile <= 9:                                   # NN layout exists on the 16x16x4 [k][row] kernels only
                bt = b.t().contiguous().cuda()
                assert _rel(K.gemm(a.cuda(), bt, True, False, cfg=(tile, sp)), ref) <= 2e-5, (tile, M, sp, "nn")
        c2 = K.gemm(a.cuda(), b.cuda(), True, True, bias=bias.cuda(), res=res.cuda(), cfg=(tile, 1))
        assert _rel(c2, ref + bias.double() + res.double()) <= 2e-5
        guard = torch.full((M + 4, N), 7.0, device="cuda")  # rows beyond M must not be written
        K.gemm(a.cuda(), b.cuda(), True, True, out=guard[:M], cfg=(tile, 1))
        assert (guard[M:] == 7.0).all()


@pytest.mark.parametrize("kind", ["l2", "smoothl1"])
def test_regression_distillation_losses(K, kind):
    """loss: l2 / smoothl1 (models/act.py:1186-1191,1255) forward + backward against torch."""
    import torch.nn.functional as F
    s = _rnd("rl.s", 6, 51, 384) * 2.0; t = _rnd("rl.t", 6, 51, 384)
    sd = s.double().requires_grad_(True)
    ref = F.mse_loss(sd, t.double()) if kind == "l2" else F.smooth_l1_loss(sd, t.double())
    (ref * 3.0).backward()
    sg = s.cuda().requires_grad_(True)
    loss = K.regression_distill_loss(sg, t.cuda(), kind)
    (loss * 3.0).backward()
    assert abs(loss.item() - ref.item()) <= 2e-6 * max(1.0, abs(ref.item()))
    assert _rel(sg.grad, sd.grad) <= 2e-6


def test_gemm_row_strided_operands(K):
    """column slices of a weight are passed with their leading dimension when that keeps the float4 path (stride % 4 == 0,
    16-byte aligned) and copied otherwise -- FoldingNet's [512, 384+3+2] conv weight is the odd-stride case."""
    x = _rnd("rs.x", 8192, 384).cuda()
    for width, lo, hi in [(389, 0, 384), (512, 128, 512), (388, 4, 388), (389, 5, 389)]:
        w = _rnd(f"rs.w{width}", 512, width).cuda()
        ws = w[:, lo:hi]
        xin = x[:, :hi - lo] if hi - lo <= 384 else _rnd("rs.x2", 8192, hi - lo).cuda()
        ref = xin.double().cpu() @ ws.double().cpu().t()
        assert _rel(K.gemm(xin, ws, True, True), ref) <= 2e-5, (width, lo, hi)
        dy = _rnd(f"rs.dy{width}", 8192, 512).cuda()
        assert _rel(K.gemm(dy, ws, True, False), dy.double().cpu() @ ws.double().cpu()) <= 2e-5       # NN with a strided B


@pytest.mark.parametrize("M,N,Kd", [(128, 3, 262144), (3, 512, 65536), (512, 5, 40000), (1, 70, 4099), (100, 8, 2048), (128, 3, 8192), (64, 8, 3001), (5, 1024, 20000)])
def test_gemm_skinny_weight_gradient(K, M, N, Kd):
    """TN products with one dimension <= 8 (first conv / FoldingNet weight gradients) run the streaming-reduction kernel."""
    a = _rnd(f"sk.a{M}{N}", Kd, M); b = _rnd(f"sk.b{M}{N}", Kd, N)
    ref = a.double().t() @ b.double()
    c = K.gemm(a.cuda(), b.cuda(), False, False)
    assert _rel(c, ref) <= 2e-5
    base = _rnd(f"sk.c{M}{N}", M, N).cuda()
    c2 = K.gemm(a.cuda(), b.cuda(), False, False, out=base.clone(), accumulate=True, alpha=0.5)
    assert _rel(c2, 0.5 * ref + base.double().cpu()) <= 2e-5


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12])
def test_gemm_explicit_configs_and_pipelined_loop(K, tile):
    """every tile shape, with and without the software-pipelined main loop, with and without split-K, all layouts."""
    M, N, Kd = 512, 384, 1024
    a = _rnd("cf.a", M, Kd); b = _rnd("cf.b", N, Kd)
    ref = a.double() @ b.double().t()
    for ak, bk in [(True, True), (True, False), (False, False)]:
        if tile >= 10 and not (ak and bk):
            continue                                       # tiles 10..12 are NT-only
        A = a.cuda() if ak else a.t().contiguous().cuda()
        Bm = b.cuda() if bk else b.t().contiguous().cuda()
        for sp in (1, 2, 4):
            c = K.gemm(A, Bm, ak, bk, cfg=(tile, sp))
            assert _rel(c, ref) <= 2e-5, (tile, ak, bk, sp)
    # K with a single / odd number of tiles exercises the pipeline prologue and drain
    for Kd2 in (32, 64, 96, 160):
        a2 = _rnd(f"cf.a{Kd2}", 256, Kd2); b2 = _rnd(f"cf.b{Kd2}", 128, Kd2)
        assert _rel(K.gemm(a2.cuda(), b2.cuda(), cfg=(tile, 1)), a2.double() @ b2.double().t()) <= 2e-5


@pytest.mark.parametrize("tile", [13, 14, 15, 16])
def test_gemm_quad_fragment_kernels_nn_tn(K, tile):
    """tiles 13 / 14 (2 x 2 waves, 128 columns) and 15 / 16 (4 x 1 waves, 64 columns: NN only): the ds_read_b128 kernels of the layouts with a row-contiguous operand (input gradients NN, weight gradients
    TN): column-/row-interleaved MFMA blocks, float4 epilogue, split-K, M tail (NN), fused epilogues, strided B, accumulate."""
    M, N, Kd = 512, 384, 1024
    a = _rnd("q.a", M, Kd); b = _rnd("q.b", N, Kd)
    ref = a.double() @ b.double().t()
    for sp in (1, 2, 4):
        assert _rel(K.gemm(a.cuda(), b.t().contiguous().cuda(), True, False, cfg=(tile, sp)), ref) <= 2e-5, (tile, "nn", sp)
        if tile == 13:
            assert _rel(K.gemm(a.t().contiguous().cuda(), b.t().contiguous().cuda(), False, False, cfg=(tile, sp)), ref) <= 2e-5, (tile, "tn", sp)
    with pytest.raises(Exception):
        K.gemm(a.cuda(), b.cuda(), True, True, cfg=(tile, 1))                         # NT has its own kernel
    if tile != 13:
        with pytest.raises(Exception):
            K.gemm(a.t().contiguous().cuda(), b.t().contiguous().cuda(), False, False, cfg=(tile, 1))
    # asymmetric operands, single / odd tile counts along K
    for Kd2 in (32, 96, 160):
        a2 = _rnd(f"q.a{Kd2}", 256, Kd2); b2 = _rnd(f"q.b{Kd2}", 128, Kd2)
        assert _rel(K.gemm(a2.cuda(), b2.t().contiguous().cuda(), True, False, cfg=(tile, 1)), a2.double() @ b2.double().t()) <= 2e-5
    # M tail on the NN layout (rows = tokens), guarded stores
    for Mt in (2080, 200, 65):
        at = _rnd(f"q.at{Mt}", Mt, Kd); reft = at.double() @ b.double().t()
        bt = b.t().contiguous().cuda()
        for sp in (1, 3):
            assert _rel(K.gemm(at.cuda(), bt, True, False, cfg=(tile, sp)), reft) <= 2e-5, (tile, Mt, sp)
        guard = torch.full((Mt + 4, N), 7.0, device="cuda")
        K.gemm(at.cuda(), bt, True, False, out=guard[:Mt], cfg=(tile, 1))
        assert (guard[Mt:] == 7.0).all() and _rel(guard[:Mt], reft) <= 2e-5
    # fused epilogue of the input-gradient GEMM: x gelu'(aux), row gate, residual; accumulate; strided B (column slice of a weight)
    aux = _rnd("q.aux", M, N).cuda(); gate = (torch.arange(M // 32) % 3).float().cuda() / 0.9; res = _rnd("q.res", M, N).cuda()
    c = K.gemm(a.cuda(), b.t().contiguous().cuda(), True, False, act=K.EPI_MUL_GELU_GRAD, aux=aux, rowscale=gate, rows_per_scale=32, res=res,
               cfg=(tile, 1))
    x = aux.double().cpu()
    gp = 0.5 * (1 + torch.erf(x / 2 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2 * torch.pi) ** 0.5
    want = ref * gp * gate.double().cpu().repeat_interleave(32)[:, None] + res.double().cpu()
    assert _rel(c, want) <= 2e-5
    base = _rnd("q.base", M, N).cuda()
    c2 = K.gemm(a.cuda(), b.t().contiguous().cuda(), True, False, out=base.clone(), accumulate=True, alpha=0.5, cfg=(tile, 1))
    assert _rel(c2, 0.5 * ref + base.double().cpu()) <= 2e-5
    wide = _rnd("q.wide", Kd, 512).cuda()                                             # B = wide[:, 128:512] : [K][N] with ldb = 512
    assert _rel(K.gemm(a.cuda(), wide[:, 128:], True, False, cfg=(tile, 1)), a.double() @ wide[:, 128:].double().cpu()) <= 2e-5


@pytest.mark.parametrize("group,R,C,N", [(32, 1024, 128, 256), (64, 2048, 384, 512)])
def test_gemm_with_maxpool_backward_generated_on_load(K, group, R, C, N):
    """act_sgemm_fx_f32 with the scattered gradient of torch.max(feature, dim=2) (models/dvae.py:211,214) as a VIRTUAL operand: input gradient
    (1,0) with A generated on load and / or a second scatter added in the epilogue, weight gradient (0,0) with A generated on load (with and
    without the activated-on-load B).  Against the same products on the materialised scatter: the generated values are the stored values, so
    the fp32 result may differ only by the kernel's summation order (here: the same kernel family -> 1e-6), and against float64."""
    import ctypes
    import act_amd.composite as CP
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(group + C)
    G = R // group
    dout = torch.randn(G, C, generator=g).to(dev)
    arg = torch.randint(0, group, (G, C), generator=g, dtype=torch.int32).to(dev)
    dense = torch.zeros(G, group, C, device=dev)
    dense.scatter_(1, arg.long().unsqueeze(1), dout.unsqueeze(1))
    dense = dense.reshape(R, C)                                         # dh[r][c] = arg[r/group][c] == r % group ? dout[r/group][c] : 0
    st = torch.cuda.current_stream().cuda_stream
    epi = K.GemmEpilogue(); epi.alpha = 1.0
    ws = torch.empty(64 << 20, device=dev)

    def call(ak, bk, M, Nn, Kd, A, lda, Bm, ldb, out, fx):
        rc = CP.lib.act_sgemm_fx_f32(ak, bk, M, Nn, Kd, A.data_ptr() if A is not None else None, lda, Bm.data_ptr(), ldb, out.data_ptr(), Nn,
                                     ctypes.byref(epi), ctypes.byref(fx), ws.data_ptr(), ws.numel() * 4, st)
        assert rc == 0, rc
    # ---- input gradient dX = dh . W, W [C][N]
    W = (torch.randn(C, N, generator=g) * 0.1).to(dev)
    ep_src = torch.randn(G, N, generator=g).to(dev)
    ep_arg = torch.randint(0, group, (G, N), generator=g, dtype=torch.int32).to(dev)
    ep_dense = torch.zeros(G, group, N, device=dev).scatter_(1, ep_arg.long().unsqueeze(1), ep_src.unsqueeze(1)).reshape(R, N)
    ref = dense.double() @ W.double()
    fx = CP.GemmFx(); fx.sa_src = dout.data_ptr(); fx.sa_arg = arg.data_ptr(); fx.group = group
    out = torch.empty(R, N, device=dev)
    call(1, 0, R, N, C, None, C, W, N, out, fx)
    assert _rel(out, ref) <= 1e-5
    assert _rel(out, K.gemm(dense, W, True, False, cfg=(13, 1))) <= 1e-6          # the 128x128 quad kernel on the stored operand
    fx2 = CP.GemmFx(); fx2.sa_src = dout.data_ptr(); fx2.sa_arg = arg.data_ptr(); fx2.group = group
    fx2.ep_src = ep_src.data_ptr(); fx2.ep_arg = ep_arg.data_ptr()
    out2 = torch.empty(R, N, device=dev)
    call(1, 0, R, N, C, None, C, W, N, out2, fx2)
    assert _rel(out2, ref + ep_dense.double()) <= 1e-5
    fx3 = CP.GemmFx(); fx3.ep_src = ep_src.data_ptr(); fx3.ep_arg = ep_arg.data_ptr(); fx3.group = group
    out3 = torch.empty(R, N, device=dev)
    call(1, 0, R, N, C, dense, C, W, N, out3, fx3)
    assert torch.equal(out3, out2)                                       # stored vs generated A: bit-identical
    # ---- weight gradient dW = dh^T . act(X), X [R][N]
    X = torch.randn(R, N, generator=g).to(dev)
    sc = (torch.rand(N, generator=g) + 0.5).to(dev); sh = (torch.randn(N, generator=g) * 0.2).to(dev)
    act = torch.relu(X * sc + sh)
    fw = CP.GemmFx(); fw.sa_src = dout.data_ptr(); fw.sa_arg = arg.data_ptr(); fw.group = group
    dw = torch.empty(C, N, device=dev)
    call(0, 0, C, N, R, None, C, X, N, dw, fw)
    assert _rel(dw, dense.double().t() @ X.double()) <= 1e-5
    fw2 = CP.GemmFx(); fw2.sa_src = dout.data_ptr(); fw2.sa_arg = arg.data_ptr(); fw2.group = group; fw2.b_scale = sc.data_ptr(); fw2.b_shift = sh.data_ptr()
    dw2 = torch.empty(C, N, device=dev)
    call(0, 0, C, N, R, None, C, X, N, dw2, fw2)
    assert _rel(dw2, dense.double().t() @ act.double()) <= 1e-5
    fw3 = CP.GemmFx(); fw3.b_scale = sc.data_ptr(); fw3.b_shift = sh.data_ptr()
    dw3 = torch.empty(C, N, device=dev)
    call(0, 0, C, N, R, dense, C, X, N, dw3, fw3)
    assert torch.equal(dw3, dw2)
    # ---- argument checks
    bad = CP.GemmFx(); bad.sa_src = dout.data_ptr()
    assert CP.lib.act_sgemm_fx_f32(1, 0, R, N, C, None, C, W.data_ptr(), N, out.data_ptr(), N, ctypes.byref(epi), ctypes.byref(bad), None, 0, st) != 0
    bad2 = CP.GemmFx(); bad2.sa_src = dout.data_ptr(); bad2.sa_arg = arg.data_ptr(); bad2.group = 48
    assert CP.lib.act_sgemm_fx_f32(1, 0, R, N, C, None, C, W.data_ptr(), N, out.data_ptr(), N, ctypes.byref(epi), ctypes.byref(bad2), None, 0, st) != 0


@pytest.mark.parametrize("group,G,C,N", [(32, 96, 384, 512), (64, 40, 768, 512), (32, 33, 128, 256), (16, 64, 256, 1024)])
def test_maxpool_backward_products_walk_the_live_entries(K, group, G, C, N):
    """csrc/pool_bwd.hip: the two products of Encoder.backward that consume the gradient of torch.max(feature, dim=2) (models/dvae.py:211,214)
    computed from the C live entries per group instead of a dense [G*n, C] operand.  Against float64 on the materialised scatter, against the
    dense on-load kernels (summation order differs: 1e-6), repeated calls bit-identical, a smaller workspace (fewer splits) within rounding,
    strided operands, entries with arg outside [0, n) dropped, gradients that are zero for most groups (the masked patches of Stage II: those groups are
    skipped through a device-side list of live groups) or for all, argument checks."""
    import ctypes
    import act_amd.composite as CP
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(group + C + G)
    R = G * group
    dout = torch.randn(G, C, generator=g)
    if C != 384:
        dout[torch.arange(G) % 5 != 0] = 0                               # Stage II: only the visible 20 % of the patches carry a gradient
        dout[5, 1:] = 0                                                  # (a live group with a single non-zero entry)
    dout = dout.to(dev)
    arg = torch.randint(0, group, (G, C), generator=g, dtype=torch.int32)
    arg[0] = 0                                                           # one group with every channel on row 0 (all other rows empty)
    arg[1, : C // 2] = group - 1
    arg = arg.to(dev)
    dense = torch.zeros(G, group, C, device=dev).scatter_(1, arg.long().unsqueeze(1), dout.unsqueeze(1)).reshape(R, C)
    st = torch.cuda.current_stream().cuda_stream
    lib = K.lib
    # ---- dX = dh . W
    wide = (torch.randn(C, N + 64, generator=g) * 0.1).to(dev)
    for W, ldw in ((wide[:, :N].contiguous(), N), (wide, N + 64)):
        out = torch.full((R, N + 4), 7.0, device=dev)
        assert lib.act_group_max_bwd_matmul_f32(dout.data_ptr(), arg.data_ptr(), G, group, C, W.data_ptr(), ldw, N, out.data_ptr(), N + 4, st) == 0
        ref = dense.double() @ W[:, :N].double()
        assert _rel(out[:, :N], ref) <= 2e-6
        assert (out[:, N:] == 7.0).all()
        out_b = torch.empty(R, N + 4, device=dev)
        assert lib.act_group_max_bwd_matmul_f32(dout.data_ptr(), arg.data_ptr(), G, group, C, W.data_ptr(), ldw, N, out_b.data_ptr(), N + 4, st) == 0
        assert torch.equal(out_b[:, :N], out[:, :N])
    if R % 128 == 0 and C % 128 == 0 and group in (32, 64):               # the dense kernel with the scatter generated on load
        epi = K.GemmEpilogue(); epi.alpha = 1.0
        fx = CP.GemmFx(); fx.sa_src = dout.data_ptr(); fx.sa_arg = arg.data_ptr(); fx.group = group
        d_on = torch.empty(R, N, device=dev)
        Wc = wide[:, :N].contiguous()
        assert CP.lib.act_sgemm_fx_f32(1, 0, R, N, C, None, C, Wc.data_ptr(), N, d_on.data_ptr(), N, ctypes.byref(epi), ctypes.byref(fx), None, 0, st) == 0
        assert _rel(out[:, :N], d_on) <= 1e-6
    # ---- dW = dh^T . act(X)
    X = torch.randn(R, N, generator=g).to(dev)
    sc = (torch.rand(N, generator=g) + 0.5).to(dev); sh = (torch.randn(N, generator=g) * 0.2).to(dev)
    need = lib.act_group_max_bwd_wgrad_workspace(G, group, C, N)
    ws = torch.empty(max(need, 16) // 4, device=dev)
    for scale, shift, act in ((sc, sh, torch.relu(X * sc + sh)), (None, None, X)):
        ref = dense.double().t() @ act.double()
        p = lambda t: t.data_ptr() if t is not None else None
        dw = torch.full((C, N + 8), 7.0, device=dev)
        assert lib.act_group_max_bwd_wgrad_f32(dout.data_ptr(), arg.data_ptr(), G, group, C, X.data_ptr(), N, N, p(scale), p(shift), dw.data_ptr(), N + 8,
                                               ws.data_ptr(), ws.numel() * 4, st) == 0
        assert _rel(dw[:, :N], ref) <= 2e-6
        assert (dw[:, N:] == 7.0).all()
        dw_b = torch.empty(C, N + 8, device=dev)
        assert lib.act_group_max_bwd_wgrad_f32(dout.data_ptr(), arg.data_ptr(), G, group, C, X.data_ptr(), N, N, p(scale), p(shift), dw_b.data_ptr(), N + 8,
                                               ws.data_ptr(), ws.numel() * 4, st) == 0
        assert torch.equal(dw_b[:, :N], dw[:, :N])
        for cap in (0, 2 * C * N * 4 + 16 * G):                          # no workspace: one split straight into dw, every group walked; room for two splits
            dw_c = torch.empty(C, N, device=dev)
            assert lib.act_group_max_bwd_wgrad_f32(dout.data_ptr(), arg.data_ptr(), G, group, C, X.data_ptr(), N, N, p(scale), p(shift), dw_c.data_ptr(), N,
                                                   ws.data_ptr() if cap else None, cap, st) == 0
            assert _rel(dw_c, ref) <= 2e-6
    # ---- an all-zero gradient: zero results (no live group at all)
    zero = torch.zeros_like(dout)
    out = torch.full((R, N), 7.0, device=dev); dw = torch.full((C, N), 7.0, device=dev)
    assert lib.act_group_max_bwd_matmul_f32(zero.data_ptr(), arg.data_ptr(), G, group, C, wide.data_ptr(), N + 64, N, out.data_ptr(), N, st) == 0
    assert lib.act_group_max_bwd_wgrad_f32(zero.data_ptr(), arg.data_ptr(), G, group, C, X.data_ptr(), N, N, None, None, dw.data_ptr(), N,
                                           ws.data_ptr(), ws.numel() * 4, st) == 0
    assert (out == 0).all() and (dw == 0).all()
    # ---- entries whose arg is outside [0, n) contribute nothing (as in the dense kernels: no row matches)
    arg_bad = arg.clone(); arg_bad[2, ::3] = group; arg_bad[3, 1::5] = -1
    keep = ((arg_bad >= 0) & (arg_bad < group))
    dense_b = torch.zeros(G, group, C, device=dev).scatter_(1, arg_bad.clamp(0, group - 1).long().unsqueeze(1), (dout * keep).unsqueeze(1)).reshape(R, C)
    Wc = wide[:, :N].contiguous()
    out = torch.empty(R, N, device=dev)
    assert lib.act_group_max_bwd_matmul_f32(dout.data_ptr(), arg_bad.data_ptr(), G, group, C, Wc.data_ptr(), N, N, out.data_ptr(), N, st) == 0
    assert _rel(out, dense_b.double() @ Wc.double()) <= 2e-6
    dw = torch.empty(C, N, device=dev)
    assert lib.act_group_max_bwd_wgrad_f32(dout.data_ptr(), arg_bad.data_ptr(), G, group, C, X.data_ptr(), N, N, None, None, dw.data_ptr(), N,
                                           ws.data_ptr(), ws.numel() * 4, st) == 0
    assert _rel(dw, dense_b.double().t() @ X.double()) <= 2e-6
    # ---- argument checks
    assert lib.act_group_max_bwd_matmul_f32(dout.data_ptr(), arg.data_ptr(), G, 48, C, Wc.data_ptr(), N, N, out.data_ptr(), N, st) != 0      # 256 % n
    assert lib.act_group_max_bwd_matmul_f32(dout.data_ptr(), arg.data_ptr(), G, group, C, Wc.data_ptr(), N, 384, out.data_ptr(), N, st) != 0  # N
    assert lib.act_group_max_bwd_matmul_f32(None, arg.data_ptr(), G, group, C, Wc.data_ptr(), N, N, out.data_ptr(), N, st) != 0
    assert lib.act_group_max_bwd_wgrad_f32(dout.data_ptr(), arg.data_ptr(), G, group, C, X.data_ptr(), N, N, sc.data_ptr(), None, dw.data_ptr(), N,
                                           ws.data_ptr(), ws.numel() * 4, st) != 0                                                         # scale without shift
    assert lib.act_group_max_bwd_wgrad_f32(dout.data_ptr(), arg.data_ptr(), G, group, C - 64, X.data_ptr(), N, N, None, None, dw.data_ptr(), N,
                                           ws.data_ptr(), ws.numel() * 4, st) != 0                                                         # C % 128
    torch.cuda.synchronize()


def test_batchnorm_backward_skips_dead_groups_without_changing_a_bit(K):
    """act_bn_bwd_groups_f32 / act_group_max_bwd_matmul_live_f32 / act_group_live_i32: with a gradient that is zero on whole groups of rows (the masked
    patches of Stage II) the BatchNorm backward neither reads dy there nor changes a bit of dx / dgamma / dbeta (the skipped terms are exact zeros),
    and the row walk does not write the dead groups' rows at all."""
    dev = torch.device("cuda:0")
    lib = K.lib
    st = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(5)
    G, n, C, N = 96, 32, 384, 512
    R = G * n
    dout = torch.randn(G, C, generator=g)
    dout[torch.arange(G) % 4 != 1] = 0
    dout = dout.to(dev)
    arg = torch.randint(0, n, (G, C), generator=g, dtype=torch.int32).to(dev)
    W = (torch.randn(C, N, generator=g) * 0.1).to(dev)
    live = torch.full((G,), 7, dtype=torch.int32, device=dev)
    assert lib.act_group_live_i32(dout.data_ptr(), G, C, live.data_ptr(), st) == 0
    assert torch.equal(live.cpu(), (torch.arange(G) % 4 == 1).to(torch.int32))
    full = torch.empty(R, N, device=dev); part = torch.full((R, N), 7.0, device=dev)
    assert lib.act_group_max_bwd_matmul_f32(dout.data_ptr(), arg.data_ptr(), G, n, C, W.data_ptr(), N, N, full.data_ptr(), N, st) == 0
    assert lib.act_group_max_bwd_matmul_live_f32(dout.data_ptr(), arg.data_ptr(), G, n, C, W.data_ptr(), N, N, part.data_ptr(), N, live.data_ptr(), st) == 0
    rows_live = live.bool().repeat_interleave(n)
    assert torch.equal(part[rows_live], full[rows_live]) and (part[~rows_live] == 7.0).all() and (full[~rows_live] == 0).all()
    # BatchNorm(+ReLU) backward on that gradient: dead rows of `part` hold garbage (7.0) and must not be read
    x = torch.randn(R, N, generator=g).to(dev)
    gamma = (torch.rand(N, generator=g) + 0.5).to(dev); beta = (torch.randn(N, generator=g) * 0.1).to(dev)
    mean = x.mean(0); rstd = torch.rsqrt(x.var(0, unbiased=False) + 1e-5)
    scale = gamma * rstd; shift = beta - mean * scale
    ws = torch.empty(8 << 20, device=dev)
    outs = []
    for dy, lv in ((full, None), (part, live)):
        dx = torch.empty(R, N, device=dev); dg = torch.empty(N, device=dev); db = torch.empty(N, device=dev)
        assert lib.act_bn_bwd_groups_f32(x.data_ptr(), dy.data_ptr(), scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), rstd.data_ptr(), 1, R, N,
                                         lv.data_ptr() if lv is not None else None, n, dx.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(),
                                         ws.numel() * 4, st) == 0
        outs.append((dx, dg, db))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # against autograd (float64)
    xd = x.double().requires_grad_(True)
    y = torch.relu(torch.nn.functional.batch_norm(xd, None, None, gamma.double(), beta.double(), True, 0.0, 1e-5))
    y.backward(full.double())
    assert _rel(outs[1][0], xd.grad) <= 2e-5
    torch.cuda.synchronize()


@pytest.mark.parametrize("tile,base", [(17, 10), (18, 11)])
def test_gemm_nt_pipelined_loop_is_bit_identical(K, tile, base):
    """tiles 17 / 18: the NT b128 kernels with the software-pipelined main loop (fragments of K-tile t+1 read during the MFMAs of tile t, LDS-only
    barrier) compute the same products in the same order as tiles 10 / 11: bit-identical, with split-K and fused epilogues, for 2 .. 192 K-tiles."""
    for (M, N, Kd) in [(256, 256, 32), (512, 384, 1024), (1024, 768, 3072), (128, 128, 96 * 32)]:
        a = _rnd(f"p.a{Kd}", M, Kd).cuda(); b = _rnd(f"p.b{Kd}", N, Kd).cuda()
        for sp in (1, 2, 3):
            if Kd // sp < 32:
                continue
            assert torch.equal(K.gemm(a, b, True, True, cfg=(tile, sp)), K.gemm(a, b, True, True, cfg=(base, sp))), (tile, M, N, Kd, sp)
        bias = _rnd("p.bias", N).cuda(); res = _rnd("p.res", M, N).cuda()
        assert torch.equal(K.gemm(a, b, True, True, bias=bias, res=res, act=K.EPI_GELU, cfg=(tile, 1)),
                           K.gemm(a, b, True, True, bias=bias, res=res, act=K.EPI_GELU, cfg=(base, 1)))
    a = _rnd("p.a", 512, 1024).cuda(); b = _rnd("p.b", 384, 1024).cuda()
    assert _rel(K.gemm(a, b, True, True, cfg=(tile, 1)), a.double().cpu() @ b.double().cpu().t()) <= 2e-5
    with pytest.raises(Exception):                         # full tiles only
        K.gemm(_rnd("p.t", 200, 64).cuda(), _rnd("p.u", 128, 64).cuda(), True, True, cfg=(tile, 1))
    with pytest.raises(Exception):                         # NT only
        K.gemm(a, b.t().contiguous(), True, False, cfg=(tile, 1))


def test_prompt_rows_fwd_bwd(K):
    """K.prompt_rows: y[b, p, :] = dropout(tok[p, :]) + ppos[p, :] and its backward (sums over the clouds) with an injected keep mask against
    float64 autograd; with in-kernel Philox: every row is tok * {0, 1/(1-p)} + ppos, the drop rate is right, the backward regenerates the same
    mask, and p = 0 is the plain broadcast."""
    B, P, D, p = 6, 64, 768, 0.1
    tok = _rnd("pr.tok", P, D); ppos = _rnd("pr.pos", P, D); dy = _rnd("pr.dy", B * P, D)
    g = torch.Generator().manual_seed(5)
    mask = (torch.rand(B, P, D, generator=g) >= p).float()
    td, pd_ = tok.double().requires_grad_(True), ppos.double().requires_grad_(True)
    ref = (td.unsqueeze(0) * mask.double() / (1 - p) + pd_.unsqueeze(0)).reshape(B * P, D)
    (ref * dy.double()).sum().backward()
    tg, pg = tok.cuda().requires_grad_(True), ppos.cuda().requires_grad_(True)
    y = K.prompt_rows(tg, pg, B, p, 0, mask.cuda())
    (y * dy.cuda()).sum().backward()
    assert _rel(y, ref) <= 1e-6 and _rel(tg.grad, td.grad) <= 2e-6 and _rel(pg.grad, pd_.grad) <= 2e-6
    # in-kernel noise
    t2, p2 = tok.cuda().requires_grad_(True), ppos.cuda().requires_grad_(True)
    y1 = K.prompt_rows(t2, p2, B, p, 1234); y2 = K.prompt_rows(tok.cuda(), ppos.cuda(), B, p, 1234); y3 = K.prompt_rows(tok.cuda(), ppos.cuda(), B, p, 99)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    yk = (tok.cuda() / (1 - p) + ppos.cuda()).unsqueeze(0); yd = ppos.cuda().unsqueeze(0); y3d = y1.detach().reshape(B, P, D)
    kept = (y3d - yk).abs() <= 1e-6 * (1 + yk.abs()); dropped = ((y3d - yd).abs() <= 1e-7) & ~kept      # kept value, or ppos alone
    assert bool((kept | dropped).all()) and abs(dropped.float().mean().item() - p) < 0.01
    assert not torch.equal(dropped[0], dropped[1])                                           # a different mask per cloud
    (y1 * dy.cuda()).sum().backward()
    want = (dy.cuda().reshape(B, P, D) * kept.float() / (1 - p)).sum(0)
    assert _rel(t2.grad, want) <= 2e-6 and _rel(p2.grad, dy.cuda().reshape(B, P, D).sum(0)) <= 2e-6
    assert torch.equal(K.prompt_rows(tok.cuda(), ppos.cuda(), 3, 0.0, 0), (tok.cuda() + ppos.cuda()).repeat(3, 1))


@pytest.mark.parametrize("tile", [3, 6, 9, 12, 13, 17])
def test_gemm_split_k_with_uneven_splits(K, tile):
    """K = 3296 = 103 x 32 rows (32 clouds x 103 visible tokens, the stress geometry): a split-K factor that does not divide K is rounded to
    32-row chunks (1664 + 1632, 1120 + 1120 + 1056, ...); every kernel family reduces exactly K rows."""
    M, N, Kd = 256, 256, 3296
    a = _rnd("us.a", M, Kd); b = _rnd("us.b", N, Kd)
    ref = a.double() @ b.double().t()
    for sp in (2, 3, 4, 6):
        if tile in (13,):
            got = K.gemm(a.t().contiguous().cuda(), b.t().contiguous().cuda(), False, False, cfg=(tile, sp))       # TN
        else:
            got = K.gemm(a.cuda(), b.cuda(), True, True, cfg=(tile, sp))
        assert _rel(got, ref) <= 2e-5, (tile, sp)


def test_gemm_autotuner_times_and_registers_a_config(K):
    """the first-use autotuner (every shape of this suite is in the shipped table, so it never runs by itself here: tests/conftest.py):
    gemm_tune times every candidate of a shape, returns a valid (tile, split-K) pair whose product is correct, and _gemm_config publishes it
    to the C-side table that the composite entry points read (act_gemm_tune_get)."""
    import ctypes
    M, N, Kd = 1024, 512, 1536
    a = _rnd("at.a", M, Kd).cuda(); b = _rnd("at.b", N, Kd).cuda()
    ws = K.workspace(a.device)
    (tile, sp), ms = K.gemm_tune(a, b, True, True, M, N, Kd, ws)
    assert 1 <= tile <= 36 and sp >= 1 and 0 < ms < 10
    assert _rel(K.gemm(a, b, True, True, cfg=(tile, sp)), a.double().cpu() @ b.double().cpu().t()) <= 2e-5
    saved = K.AUTOTUNE
    K.AUTOTUNE = True
    try:
        K._GEMM_CACHE.pop((1, 1, M, N, Kd, a.device.index), None)
        cfg = K._gemm_config(a, b, True, True, M, N, Kd, ws)
    finally:
        K.AUTOTUNE = saved
    t, s_ = ctypes.c_int(-1), ctypes.c_int(-1)
    assert K.lib.act_gemm_tune_get(1, 1, M, N, Kd, ctypes.byref(t), ctypes.byref(s_)) == 0 and (t.value, s_.value) == tuple(cfg)
    assert K._NEW_TUNED.pop((1, 1, M, N, Kd), None) == tuple(cfg)       # (deliberate tuning: not a gap of the shipped table, see conftest.py)


@pytest.mark.parametrize("ak,bk,M,N,Kd", [(1, 1, 1792, 384, 1536), (1, 1, 2080, 384, 768), (1, 0, 1792, 384, 1536), (1, 0, 2080, 1536, 384),
                                          (0, 0, 384, 1536, 1792), (0, 0, 256, 128, 4000), (1, 1, 8192, 768, 3072), (1, 0, 8192, 2304, 768)])
def test_every_autotuner_candidate_is_a_correct_kernel(K, ak, bk, M, N, Kd):
    """every (tile id, split-K) pair gemm_tune can emit for a shape -- 32x32x2 and 16x16x4 loops, pipelined variants, NT b128 fragments, NN / TN
    quad fragments, M tails (2,080 rows), uneven last K range (1,792 = 2 x 608 + 576) -- computes the product (float64 reference)."""
    a = _rnd(f"cand.a{ak}{bk}{M}{Kd}", M if ak else Kd, Kd if ak else M).cuda(); b = _rnd(f"cand.b{ak}{bk}{N}{Kd}", N if bk else Kd, Kd if bk else N).cuda()
    A2 = a.double().cpu() if ak else a.double().cpu().t(); B2 = b.double().cpu().t() if bk else b.double().cpu()
    ref = A2 @ B2
    tr = []
    K.gemm_tune(a, b, bool(ak), bool(bk), M, N, Kd, K.workspace(a.device), reps=1, trace=tr)
    assert len(tr) >= 4 and len({t for t, _, _ in tr}) >= 3, tr
    for tile, sp, _ in tr:
        assert _rel(K.gemm(a, b, bool(ak), bool(bk), cfg=(tile, sp)), ref) <= 2e-5, (tile, sp)


@pytest.mark.parametrize("T,dims,splits", [(1792, [(384, 1536), (1536, 384)], 0), (1792, [(384, 384), (1152, 384)], 0), (1792, [(384, 1536), (1536, 384), (384, 384), (1152, 384)], 4),
                                           (8192, [(384, 384), (1152, 384)], 0), (256, [(128, 128)], 1), (3296, [(768, 3072), (3072, 768)], 0), (64, [(128, 256), (256, 128)], 0)])
def test_grouped_weight_gradients_match_the_single_gemm_path(K, T, dims, splits):
    """act_sgemm_tn_grouped_f32: the dW = dy^T . x of several Linears (+ their bias gradients) in one launch.  With the same K-range count the
    product is BIT-IDENTICAL to the per-GEMM quad-fragment kernel (tile 13) -- same products, same order, same fixed-order fold of the K
    ranges -- and within fp32 noise of a float64 reference; the bias gradient (another summation order than act_colsum_f32) is checked
    against float64.  Operands are row-strided views (dqkv-style leading dimensions) for half of the problems."""
    pairs, ref = [], []
    for i, (M, N) in enumerate(dims):
        dy_full = _rnd(f"gg.dy{T}{M}{N}{i}", T, M + 128 * (i % 2)).cuda(); x = _rnd(f"gg.x{T}{M}{N}{i}", T, N).cuda()
        dy = dy_full[:, :M] if i % 2 else dy_full                       # a strided view: lda = M + 128
        pairs.append((dy, x)); ref.append((dy.double().t() @ x.double(), dy.double().sum(0)))
    dws, dbs = K.gemm_tn_grouped(pairs, splits=splits)
    import ctypes
    probs = (K.GemmTnProblem * len(dims))()
    for i, ((dy, x), dw, db) in enumerate(zip(pairs, dws, dbs)):
        probs[i] = K.GemmTnProblem(dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dw.data_ptr(), dims[i][1], dims[i][0], dims[i][1], db.data_ptr())
    used = splits if splits > 0 else K.lib.act_sgemm_tn_grouped_splits(probs, len(dims), T)
    kps = -(-(-(-T // used)) // 32) * 32
    used = -(-T // kps)                                                 # the K-range count after rounding the ranges to multiples of 32
    for (dy, x), dw, db, (rw, rb) in zip(pairs, dws, dbs, ref):
        assert _rel(dw, rw) <= 2e-5 and _rel(db, rb) <= 2e-5
        single = K.gemm(dy, x, False, False, cfg=(13, used))
        assert torch.equal(dw, single), (dy.shape, x.shape, used)
    again, again_b = K.gemm_tn_grouped(pairs, splits=splits)            # deterministic
    assert all(torch.equal(a, b) for a, b in zip(dws, again)) and all(torch.equal(a, b) for a, b in zip(dbs, again_b))
    no_bias, none = K.gemm_tn_grouped(pairs, want_bias=False, splits=splits)
    assert all(b is None for b in none) and all(torch.equal(a, b) for a, b in zip(dws, no_bias))


def test_grouped_weight_gradients_reject_bad_shapes(K):
    from act_amd._C import ActHipError
    with pytest.raises(ActHipError):
        K.gemm_tn_grouped([(torch.zeros(64, 100, device="cuda"), torch.zeros(64, 128, device="cuda"))])       # M % 128 != 0
    with pytest.raises(ActHipError):
        K.gemm_tn_grouped([(torch.zeros(64, 128, device="cuda"), torch.zeros(32, 128, device="cuda"))])       # different row counts


@pytest.mark.parametrize("M,N,Kd,sp", [(1024, 512, 768, 1), (8192, 768, 3072, 2), (256, 128, 96, 1), (512, 192, 4000, 3)])
def test_nt_kernel_with_32_deep_k_tiles_is_bit_identical(K, M, N, Kd, sp):
    """tiles 20 / 21 (NT, 32-deep K tiles: full 128-byte rows per staging load) execute the products of tiles 10 / 11 in the same order"""
    a = _rnd(f"nt32.a{M}{Kd}", M, Kd).cuda(); b = _rnd(f"nt32.b{N}{Kd}", N, Kd).cuda()
    bias = _rnd(f"nt32.bias{N}", N).cuda()
    for t32, t16 in ((20, 10), (21, 11)):
        if N % (128 if t32 == 20 else 64):
            continue
        assert torch.equal(K.gemm(a, b, True, True, bias=bias, cfg=(t32, sp)), K.gemm(a, b, True, True, bias=bias, cfg=(t16, sp))), (t32, sp)


@pytest.mark.parametrize("tile,base", [(30, 10), (31, 11), (32, 12)])
def test_nt_hand_scheduled_loop_is_bit_identical(K, tile, base):
    """tiles 30 / 31 / 32 (gemm_nt_asm_kernel.h: the K loop as one hand-scheduled asm statement, 32-deep K tiles) execute the products of tiles
    10 / 11 / 12 in the same order: bit-identical for 1, 2, 3 (tail paths), 4, 5 ... 96 K-tiles, with split-K, with an M tail (rows clamped on
    load, guarded on store), with every fused epilogue and on row-strided operands."""
    for (M, N, Kd) in [(256, 256, 32), (256, 128, 64), (128, 256, 96), (384, 128, 128), (256, 256, 160), (512, 384, 1024), (1024, 768, 3072),
                       (2080, 384, 384), (65, 128, 192), (1792, 1152, 384)]:
        a = _rnd(f"nta.a{M}.{Kd}", M, Kd).cuda(); b = _rnd(f"nta.b{N}.{Kd}", N, Kd).cuda()
        for sp in (1, 2, 3):
            if Kd // sp < 32:
                continue
            assert torch.equal(K.gemm(a, b, True, True, cfg=(tile, sp)), K.gemm(a, b, True, True, cfg=(base, sp))), (tile, M, N, Kd, sp)
        bias = _rnd(f"nta.bias{N}", N).cuda(); res = _rnd(f"nta.res{M}.{N}", M, N).cuda()
        for act in (K.EPI_NONE, K.EPI_GELU, K.EPI_RELU):
            assert torch.equal(K.gemm(a, b, True, True, bias=bias, res=res, act=act, cfg=(tile, 1)),
                               K.gemm(a, b, True, True, bias=bias, res=res, act=act, cfg=(base, 1))), (tile, M, N, Kd, act)
    a = _rnd("nta.a", 512, 1024).cuda(); b = _rnd("nta.b", 384, 1024).cuda()
    assert _rel(K.gemm(a, b, True, True, cfg=(tile, 1)), a.double().cpu() @ b.double().cpu().t()) <= 2e-5
    wide_a = _rnd("nta.wa", 512, 1024 + 64).cuda(); wide_b = _rnd("nta.wb", 384, 1024 + 32).cuda()       # row-strided views (lda != K)
    assert torch.equal(K.gemm(wide_a[:, 32:32 + 1024], wide_b[:, :1024], True, True, cfg=(tile, 1)),
                       K.gemm(wide_a[:, 32:32 + 1024], wide_b[:, :1024], True, True, cfg=(base, 1)))
    guard = torch.full((2080 + 64, 384), 7.0, device="cuda")                                             # the M tail never writes past row M
    K.gemm(_rnd("nta.a2080.384", 2080, 384).cuda(), _rnd("nta.b384.384", 384, 384).cuda(), True, True, out=guard[:2080], cfg=(tile, 1))
    assert torch.all(guard[2080:] == 7.0)
    with pytest.raises(Exception):                         # K ranges must be multiples of 32
        K.gemm(_rnd("nta.t", 256, 48).cuda(), _rnd("nta.u", 128, 48).cuda(), True, True, cfg=(tile, 1))
    with pytest.raises(Exception):                         # NT only
        K.gemm(a, b.t().contiguous(), True, False, cfg=(tile, 1))


@pytest.mark.parametrize("tile,base", [(33, 13), (34, 14), (35, 15), (36, 16)])
def test_nn_tn_hand_scheduled_loop_is_bit_identical(K, tile, base):
    """tiles 33..36 (gemm_q_asm_kernel.h: NN / TN on the hand-scheduled main loop, row-contiguous operands read as quad fragments) execute the
    products of the quad-fragment tiles 13..16 in the same order: bit-identical for 1 .. 96 K-tiles, with split-K, an M tail (NN), the fused
    epilogues of the backward pass (gelu', relu mask, bias, residual, accumulate) and row-strided operands."""
    for (M, N, Kd) in [(256, 256, 32), (256, 128, 64), (128, 256, 96), (384, 128, 128), (256, 256, 160), (512, 384, 1024), (1024, 768, 3072),
                       (2080, 384, 384), (65, 128, 192), (1792, 384, 1536)]:
        a = _rnd(f"qa.a{M}.{Kd}", M, Kd).cuda(); bt = _rnd(f"qa.b{N}.{Kd}", Kd, N).cuda()
        for sp in (1, 2, 3):
            if Kd // sp < 32:
                continue
            assert torch.equal(K.gemm(a, bt, True, False, cfg=(tile, sp)), K.gemm(a, bt, True, False, cfg=(base, sp))), (tile, "nn", M, N, Kd, sp)
            if tile == 33 and M % 128 == 0:
                at = a.t().contiguous()
                assert torch.equal(K.gemm(at, bt, False, False, cfg=(tile, sp)), K.gemm(at, bt, False, False, cfg=(base, sp))), (tile, "tn", M, N, Kd, sp)
        bias = _rnd(f"qa.bias{N}", N).cuda(); res = _rnd(f"qa.res{M}.{N}", M, N).cuda(); aux = _rnd(f"qa.aux{M}.{N}", M, N).cuda()
        for act in (K.EPI_NONE, K.EPI_MUL_GELU_GRAD, K.EPI_MUL_RELU_MASK):
            kw = dict(bias=bias, res=res, act=act, aux=aux if act != K.EPI_NONE else None)
            assert torch.equal(K.gemm(a, bt, True, False, cfg=(tile, 1), **kw), K.gemm(a, bt, True, False, cfg=(base, 1), **kw)), (tile, M, N, Kd, act)
        o1 = res.clone(); o2 = res.clone()
        K.gemm(a, bt, True, False, out=o1, accumulate=True, alpha=0.5, cfg=(tile, 1)); K.gemm(a, bt, True, False, out=o2, accumulate=True, alpha=0.5, cfg=(base, 1))
        assert torch.equal(o1, o2)
    a = _rnd("qa.a", 512, 1024).cuda(); bt = _rnd("qa.bt", 1024, 384).cuda()
    assert _rel(K.gemm(a, bt, True, False, cfg=(tile, 1)), a.double().cpu() @ bt.double().cpu()) <= 2e-5
    wide_a = _rnd("qa.wa", 512, 1024 + 64).cuda(); wide_b = _rnd("qa.wb", 1024, 384 + 128).cuda()       # row-strided views
    assert torch.equal(K.gemm(wide_a[:, 32:32 + 1024], wide_b[:, 128:], True, False, cfg=(tile, 1)),
                       K.gemm(wide_a[:, 32:32 + 1024], wide_b[:, 128:], True, False, cfg=(base, 1)))
    guard = torch.full((2080 + 64, 384), 7.0, device="cuda")
    K.gemm(_rnd("qa.a2080.384", 2080, 384).cuda(), _rnd("qa.b384.384", 384, 384).cuda(), True, False, out=guard[:2080], cfg=(tile, 1))
    assert torch.all(guard[2080:] == 7.0)
    with pytest.raises(Exception):                         # NT has its own kernels
        K.gemm(a, _rnd("qa.nt", 384, 1024).cuda(), True, True, cfg=(tile, 1))
    if tile != 33:
        with pytest.raises(Exception):                     # TN: 128 x 128 only
            K.gemm(a.t().contiguous(), bt, False, False, cfg=(tile, 1))


@pytest.mark.parametrize("R,N,Kd,group", [(1024, 256, 128, 32), (512, 384, 512, 32), (768, 192, 256, 64), (256, 512, 1024, 32), (384, 128, 32, 64), (256, 64, 96, 32)])
def test_fused_nt_launches_on_the_hand_scheduled_loop_are_bit_identical(K, R, N, Kd, group):
    """act_sgemm_fx_f32 (1,1): column statistics, A-side affine + ReLU on load, group max (+ arg-max), with and without the C store -- the launches
    on the hand-scheduled main loop (default) against the compiler-scheduled kernels (act_gemm_fx_asm(0)) bit for bit, and against float64."""
    import ctypes
    import act_amd.composite as CP
    a = _rnd(f"fxa.a{R}.{Kd}", R, Kd).cuda(); w = _rnd(f"fxa.w{N}.{Kd}", N, Kd).cuda()
    bias = _rnd(f"fxa.b{N}", N).cuda()
    sc = (_rnd(f"fxa.sc{Kd}", Kd).cuda() * 0.5 + 1.0); sh = _rnd(f"fxa.sh{Kd}", Kd).cuda() * 0.3
    st = K.stream()
    epi = K.GemmEpilogue(alpha=1.0, bias=bias.data_ptr())

    def run(kind, asm):
        prev = CP.lib.act_gemm_fx_asm(asm)
        try:
            fx = CP.GemmFx(); out = torch.full((R, N), 3.0, device="cuda")
            res = [out]
            if kind == "stats":
                ts = torch.zeros(CP.lib.act_sgemm_fx_tile_stats_floats(R, N), device="cuda"); fx.tile_stats = ts.data_ptr(); res.append(ts)
            else:
                fx.a_scale = sc.data_ptr(); fx.a_shift = sh.data_ptr(); fx.group = group
                gm = torch.zeros(R // group, N, device="cuda"); ga = torch.zeros(R // group, N, dtype=torch.int32, device="cuda")
                fx.gmax = gm.data_ptr(); fx.garg = ga.data_ptr(); fx.store_c = 1 if kind == "affine_max" else 0
                res += [gm, ga]
            rc = CP.lib.act_sgemm_fx_f32(1, 1, R, N, Kd, a.data_ptr(), Kd, w.data_ptr(), Kd, out.data_ptr(), N, ctypes.byref(epi), ctypes.byref(fx),
                                         None, 0, st)
            assert rc == 0, (kind, asm, rc)
            torch.cuda.synchronize()
            return res
        finally:
            CP.lib.act_gemm_fx_asm(prev)

    for kind in ("stats", "affine_max", "affine_max_nostore"):
        r1, r0 = run(kind, 1), run(kind, 0)
        for x, y in zip(r1, r0):
            assert torch.equal(x, y), (kind, R, N, Kd)
        ad = a.double().cpu()
        if kind != "stats":
            ad = torch.clamp_min(ad * sc.double().cpu() + sh.double().cpu(), 0.0)
        ref = ad @ w.double().cpu().t() + bias.double().cpu()
        if kind == "affine_max_nostore":
            assert torch.all(r1[0] == 3.0)                                 # C untouched
        else:
            assert _rel(r1[0], ref) <= 2e-5
        if kind == "stats":
            ts = r1[1].view(R // 128, 2, N).double().cpu(); blocks = ref.view(R // 128, 128, N)
            assert _rel(ts[:, 0], blocks.mean(1)) <= 2e-5 and _rel(ts[:, 1], ((blocks - blocks.mean(1, keepdim=True)) ** 2).sum(1)) <= 1e-4
        else:
            assert _rel(r1[1], ref.view(R // group, group, N).max(1)[0]) <= 2e-5


@pytest.mark.parametrize("tile,ak,bk", [(10, 1, 1), (11, 1, 1), (12, 1, 1), (21, 1, 1), (30, 1, 1), (31, 1, 1), (32, 1, 1), (33, 1, 0), (36, 1, 0), (33, 0, 0), (13, 1, 0), (14, 1, 0), (13, 0, 0), (7, 1, 0), (3, 1, 1)])
def test_gemm_epilogue_scalar_fallback_matches_the_vector_path(K, tile, ak, bk):
    """the vector epilogue (float4 / float2 accesses of C, bias, residual, aux: gemm_common.h::epilogue_rows) needs 16-byte aligned pointers and
    leading dimensions % 4 == 0; anything else takes the scalar accesses of the same code.  C-ABI level: C / bias / residual / aux shifted by one
    float inside larger allocations (odd leading dimensions), every activation -> bit-identical to the aligned launch."""
    import ctypes
    lib, ptr, stream = K.lib, K.ptr, K.stream
    M, N, Kd = 256, 256, 192
    a = _rnd(f"epi.a{tile}", *((M, Kd) if ak else (Kd, M))).cuda(); b = _rnd(f"epi.b{tile}", *((N, Kd) if bk else (Kd, N))).cuda()
    bias = _rnd("epi.bias", N + 1).cuda(); res = _rnd("epi.res", M, N + 5).cuda(); auxin = _rnd("epi.aux", M, N + 5).cuda()
    rs = (torch.rand(M // 32, device="cuda") + 0.5)
    ws = K.workspace(a.device)

    def run(shift, act, use_res, row_div):
        ldo = N + 5 if shift else N + 4
        cbuf = torch.zeros(M * ldo + 8, device="cuda"); xbuf = torch.zeros(M * ldo + 8, device="cuda")
        xbuf[shift:shift + M * ldo].view(M, ldo)[:, :N] = auxin[:, :N]
        rbuf = torch.zeros(M * ldo + 8, device="cuda"); rbuf[shift:shift + M * ldo].view(M, ldo)[:, :N] = res[:, :N]
        bb = torch.zeros(N + 8, device="cuda"); bb[shift:shift + N] = bias[:N]
        e = K.GemmEpilogue(alpha=0.5, act=act, accumulate=0, rows_per_scale=32, ldr=ldo if use_res else 0, ldaux=ldo, res_row_div=row_div,
                           bias=bb.data_ptr() + 4 * shift, rowscale=rs.data_ptr(), res=(rbuf.data_ptr() + 4 * shift) if use_res else None,
                           aux=xbuf.data_ptr() + 4 * shift)
        rc = lib.act_sgemm_ex_f32(ak, bk, M, N, Kd, a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), cbuf.data_ptr() + 4 * shift, ldo,
                                  ctypes.byref(e), ptr(ws), ws.numel() * 4, tile, 1, stream())
        assert rc == 0, rc
        torch.cuda.synchronize()
        return cbuf[shift:shift + M * ldo].view(M, ldo)[:, :N].clone(), xbuf[shift:shift + M * ldo].view(M, ldo)[:, :N].clone()

    for act in (K.EPI_NONE, K.EPI_GELU, K.EPI_RELU, K.EPI_MUL_GELU_GRAD, K.EPI_MUL_RELU_MASK):
        for use_res, row_div in ((True, 0), (True, 32), (False, 0)):
            c0, x0 = run(0, act, use_res, row_div)
            c1, x1 = run(1, act, use_res, row_div)
            assert torch.equal(c0, c1), (tile, act, use_res, row_div)
            assert torch.equal(x0, x1)
            assert c0.abs().max() > 0


def test_first_use_tuning_cannot_change_a_result_bit(K):
    """Product default (ACT_GEMM_AUTOTUNE=1): a shape in no table is timed over kernels.stable_candidates only -- one tile family at one
    shape-determined split-K -- so whichever candidate the stopwatch prefers, the result is the same.  Every candidate of an unlisted NT, NN and
    TN shape (incl. an M tail and a split-K case) gives torch.equal outputs; tuning the shape twice with the candidate order reversed (a different
    winner is possible) gives torch.equal products; a shape no fast family serves is not timed at all."""
    ws = K.workspace(torch.device("cuda:0"))
    assert K.AUTOTUNE and not K.AUTOTUNE_FULL
    cases = [(True, True, 1344, 384, 1536), (True, True, 6144, 768, 768), (True, True, 2080, 1152, 384),
             (True, False, 1344, 384, 1536), (True, False, 6144, 768, 768), (False, False, 384, 1536, 6144)]
    for ak, bk, M, N, Kd in cases:
        a = _rnd(f"st.a{M}.{Kd}", *((M, Kd) if ak else (Kd, M))).cuda(); b = _rnd(f"st.b{N}.{Kd}", *((N, Kd) if bk else (Kd, N))).cuda()
        cands = K.stable_candidates(a, b, ak, bk, M, N, Kd, ws)
        assert cands and len({sp for _, sp in cands}) == 1, (ak, bk, M, N, Kd, cands)
        if ak:
            assert len(cands) >= 3
        outs = [K.gemm(a, b, ak, bk, cfg=c) for c in cands]
        for c, o in zip(cands[1:], outs[1:]):
            assert torch.equal(o, outs[0]), (ak, bk, M, N, Kd, cands[0], c)
        ref = (a.double() if ak else a.double().t()) @ (b.double().t() if bk else b.double())
        assert _rel(outs[0], ref) <= 2e-5
        w1, _ = K.gemm_tune(a, b, ak, bk, M, N, Kd, ws, cands=cands)
        w2, _ = K.gemm_tune(a, b, ak, bk, M, N, Kd, ws, cands=cands[::-1])
        assert w1 in cands and w2 in cands and torch.equal(K.gemm(a, b, ak, bk, cfg=w1), K.gemm(a, b, ak, bk, cfg=w2))
    assert stable_split_is_a_function_of_the_shape(K)
    # the split-K case really splits, the wide case does not
    assert K.stable_split(1344, 384, 1536, 1 << 30) > 1 and K.stable_split(6144, 768, 768, 1 << 30) == 1
    # no fast family (K % 32 != 0): cost model, nothing timed
    a = _rnd("st.odd.a", 1024, 1000).cuda(); b = _rnd("st.odd.b", 512, 1000).cuda()
    assert K.stable_candidates(a, b, True, True, 1024, 512, 1000, ws) == [] and K.first_use_config(a, b, True, True, 1024, 512, 1000, ws) == (0, 0)
    # end to end through the product entry: an unlisted shape is tuned on first use, cached, registered C-side, and reproducible
    M, N, Kd = 1344, 768, 1536
    a = _rnd("st.e2e.a", M, Kd).cuda(); b = _rnd("st.e2e.b", N, Kd).cuda()
    assert (1, 1, M, N, Kd) not in K._GEMM_TABLE
    K._GEMM_CACHE.pop((1, 1, M, N, Kd, 0), None)
    c1 = K.gemm(a, b)
    cfg1 = K._GEMM_CACHE[(1, 1, M, N, Kd, 0)]
    K._GEMM_CACHE.pop((1, 1, M, N, Kd, 0), None)
    c2 = K.gemm(a, b)
    assert cfg1 in K.stable_candidates(a, b, True, True, M, N, Kd, ws) and torch.equal(c1, c2)
    K._NEW_TUNED.pop((1, 1, M, N, Kd), None)


def stable_split_is_a_function_of_the_shape(K):
    return all(K.stable_split(M, N, Kd, 1 << 28) == K.stable_split(M, N, Kd, 1 << 28) for (M, N, Kd) in [(1344, 384, 1536), (64, 64, 8192)])
