"""GPU parity of the model mirror (act_amd.models) against the goldens produced by the reference's own modules and
against the CPU oracle with every random draw replayed (mask, DropPath, gumbel, prompt dropout)."""
import numpy as np
import pytest
import torch

from tests.conftest import golden
from tests.golden.fill import fill_module, fill_tensor, clouds, TINY_STAGE2, TINY_B, TINY_N

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _rel(a, ref):
    a = torch.as_tensor(a).detach().double().cpu(); ref = torch.as_tensor(ref).detach().double().cpu()
    assert a.shape == ref.shape, (a.shape, ref.shape)
    return ((a - ref).abs().max() / max(1.0, ref.abs().max())).item()


def _grad_close(a, ref, name, tol=TOL, frac=1e-3):
    """Element-wise 1e-4 (relative to the largest element) -- or, for the full-geometry graphs whose discrete selections (max over 32 / 64
    points, max over k neighbours, arg-min of Chamfer, LeakyReLU / ReLU kinks at 262,144 x 512 activations) sit within rounding distance
    of a switch: the FLIPPED elements are counted.  A flipped selection reroutes one gradient row, so at most a 1e-3 fraction of a
    gradient's elements may exceed the tolerance and the gradient as a whole must still agree to 5e-3 in the L2 sense.  (The reference's
    own fp32 math moves by as much under a different summation order: test_oracle_golden.py::test_stage1_tiny_gradient_conditioning.)"""
    a = torch.as_tensor(a).detach().double().cpu(); ref = torch.as_tensor(ref).detach().double().cpu()
    assert a.shape == ref.shape, (name, a.shape, ref.shape)
    scale = max(1.0, ref.abs().max().item())
    err = (a - ref).abs()
    if err.max().item() <= tol * scale:
        return 0
    flipped = int((err > tol * scale).sum())
    l2 = (err.norm() / ref.norm().clamp_min(1e-30)).item()
    # (a gradient with a few hundred elements: allow two of them -- each is a sum over 262,144 rows, one rerouted row moves it)
    assert flipped <= max(frac * err.numel(), 2) and l2 <= 5e-3, (name, "flipped elements", flipped, "of", err.numel(), "max", err.max().item() / scale, "l2", l2)
    print(f"[flip-tolerant] {name}: {flipped} of {err.numel()} elements beyond {tol} (max {err.max().item() / scale:.2e}, L2 {l2:.2e})")
    return flipped


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _gumbel_noise(shape):
    torch.manual_seed(777)
    return -torch.empty(shape).exponential_().log()


def _tiny(dev, prefix="g4."):
    from act_amd.models import build_model_from_cfg
    from act_amd.utils.config import EasyDict
    torch.manual_seed(0)
    model = build_model_from_cfg(EasyDict(TINY_STAGE2))
    fill_module(model, prefix)
    return model.to(dev).train()


def test_registry_contract():
    from act_amd.models import build_model_from_cfg, MODELS
    from act_amd.utils.config import EasyDict
    assert "ACT_PointDistillation" in MODELS and "ACTPromptedDiscreteVAEwithVIT" in MODELS
    with pytest.raises(KeyError):
        build_model_from_cfg(EasyDict(NAME="NoSuchModel"))
    with pytest.raises(KeyError):
        build_model_from_cfg(EasyDict(foo=1))


def test_encoder_and_transformer_encoder_golden(dev):
    from act_amd.models.dvae import Encoder
    from act_amd.models.act import TransformerEncoder
    g = golden("g2_encoder")
    nb = torch.from_numpy(golden("g1_group")["neighborhood"]).to(dev)
    enc = fill_module(Encoder(128), "g2.enc.").to(dev)
    tenc = fill_module(TransformerEncoder(embed_dim=128, depth=2, num_heads=2, drop_path_rate=0.0), "g2.tenc.").to(dev)
    pos = fill_tensor("g2.pos", (4, 64, 128), "b").to(dev)
    enc.train(); tok = enc(nb)
    assert _rel(tok, g["tok_train"]) <= TOL
    assert _rel(enc.first_conv[1].running_mean, g["bn1_running_mean"]) <= TOL
    assert _rel(enc.first_conv[1].running_var, g["bn1_running_var"]) <= TOL
    enc.eval(); assert _rel(enc(nb), g["tok_eval"]) <= TOL
    assert _rel(tenc(tok, pos), g["out"]) <= TOL


def test_stage2_tiny_golden_loss_grads_adamw(dev):
    from act_amd.utils.draws import Draws
    from act_amd.tools import builder
    g = golden("g4_stage2")
    model = _tiny(dev)
    model.dvae_tokenizer.prompt_dropout.p = 0.0
    pts = torch.from_numpy(clouds(4, TINY_B, TINY_N)).to(dev)
    draws = Draws({"mask": torch.from_numpy(g["mask"]), "gumbel": _gumbel_noise((TINY_B, 16, 64))}, device=dev)
    with torch.no_grad():
        nb, c = model.group_divider(pts)
        assert _rel(model.dvae_tokenizer.forward_tokenizer_features(nb, c, draws=draws), g["teacher_feat"]) <= TOL
    loss = model(pts, draws=draws)
    assert abs(loss.item() - g["loss"][0]) <= TOL
    loss.backward()
    pd = dict(model.named_parameters())
    for n, v in zip(g["grad_names"], g["grad_norms"]):
        assert abs(pd[str(n)].grad.norm().item() - v) <= TOL * max(1.0, v), n
    assert _rel(pd["ACT_encoder.blocks.blocks.0.attn.qkv.weight"].grad, g["grad_qkv0"]) <= TOL
    groups = builder.add_weight_decay(model, 0.05)
    assert [len(groups[0]["params"]), len(groups[1]["params"])] == g["n_param_groups"].tolist()
    opt = torch.optim.AdamW(groups, lr=1e-3, weight_decay=0.05)
    opt.step(); model.zero_grad()
    loss2 = model(pts, draws=draws); loss2.backward(); opt.step()
    assert abs(loss2.item() - g["loss"][1]) <= TOL
    for n, v in zip(g["grad_names"][:3], g["norms_after_2_steps"]):
        assert abs(pd[str(n)].detach().norm().item() - v) <= TOL * max(1.0, v)


def test_stage2_tiny_vs_oracle_all_draws_active(dev):
    """drop_path 0.3 + prompt dropout 0.1 + mask + gumbel: record in the oracle, replay in the HIP path."""
    import copy
    from oracle import models as OM, layers as OL
    from act_amd.models import build_model_from_cfg
    from act_amd.utils.config import EasyDict
    from act_amd.utils.draws import Draws
    cfg = copy.deepcopy(TINY_STAGE2)
    cfg["transformer_config"]["drop_path_rate"] = 0.3
    torch.manual_seed(1)
    oracle = fill_module(OM.ACT_PointDistillation(OM.edict(cfg)), "dp.").train()
    model = build_model_from_cfg(EasyDict(cfg))
    model.load_state_dict(oracle.state_dict(), strict=True)
    model.to(dev).train()
    pts = torch.from_numpy(clouds(5, 4, TINY_N))
    rec = OL.Draws(record=True)
    lo = oracle(pts, rec); lo.backward()
    assert any(k.startswith("enc.1") for k in rec.table) and any(k.startswith("prompt.") for k in rec.table)
    lg = model(pts.to(dev), draws=Draws(rec.table, device=dev)); lg.backward()
    assert abs(lg.item() - lo.item()) <= TOL
    od = dict(oracle.named_parameters())
    for n, p in model.named_parameters():
        if p.requires_grad and od[n].grad is not None and p.grad is not None:
            assert _rel(p.grad, od[n].grad) <= TOL, (n, _rel(p.grad, od[n].grad))


@pytest.mark.parametrize("kind,cls_loss", [("ntxent", False), ("barlow", False), ("ntxent", True)])
def test_stage2_contrastive_losses_vs_oracle(dev, kind, cls_loss):
    """loss: ntxent | barlow (models/act.py:1192-1195,1250-1254) at the tiny geometry, every draw replayed: loss and every parameter gradient
    against the oracle, with and without the cls_loss global term.  (Restated from lightly 1.2.28's published algorithms on both sides: the
    package is absent, parity against it is unpinned -- oracle/layers.py.)"""
    import copy
    from oracle import models as OM, layers as OL
    from act_amd.models import build_model_from_cfg
    from act_amd.utils.config import EasyDict
    from act_amd.utils.draws import Draws
    cfg = copy.deepcopy(TINY_STAGE2)
    cfg["loss"] = kind
    if cls_loss:
        cfg["transformer_config"]["cls_loss"] = True
        cfg["transformer_config"]["register_shallow_hook"] = 1
    torch.manual_seed(1)
    oracle = fill_module(OM.ACT_PointDistillation(OM.edict(cfg)), "ct.").train()
    model = build_model_from_cfg(EasyDict(cfg))
    model.load_state_dict(oracle.state_dict(), strict=True)
    model.to(dev).train()
    pts = torch.from_numpy(clouds(7, 4, TINY_N))
    rec = OL.Draws(record=True)
    lo = oracle(pts, rec); lo.backward()
    lg = model(pts.to(dev), draws=Draws(rec.table, device=dev)); lg.backward()
    assert abs(lg.item() - lo.item()) <= TOL * max(1.0, abs(lo.item())), (lg.item(), lo.item())
    od = dict(oracle.named_parameters())
    checked = 0
    for n, p in model.named_parameters():
        if p.requires_grad and od[n].grad is not None and p.grad is not None:
            assert _rel(p.grad, od[n].grad) <= TOL, (n, _rel(p.grad, od[n].grad))
            checked += 1
    assert checked > 20


@pytest.mark.parametrize("B", [2, 8, 128])
def test_stage2_full_geometry_vs_oracle(dev, B):
    """configs[1] geometry (N=1024, G=64, M=32, d=384 x 12, ViT-B teacher) at B = 2, B = 8 and the HEADLINE batch B = 128 (BatchNorm over 262,144
    rows, the split-K / tile choices of the benchmarked launches) against the CPU oracle with every draw replayed: loss and frozen-teacher
    features within 1e-4, six gradients across the graph (reference: models/act.py:1203-1258, models/dvae.py:189-215)."""
    from oracle import models as OM, layers as OL
    from act_amd.models import build_model_from_cfg
    from act_amd.utils.config import cfg_from_yaml_file
    from act_amd.utils.draws import Draws
    cfg = cfg_from_yaml_file("cfgs/pretrain/pretrain_act_distill.yaml").model
    cfg.dvae_config.ckpt = "none"
    torch.manual_seed(2)
    oracle = OM.ACT_PointDistillation(OM.edict(cfg)).train()
    model = build_model_from_cfg(cfg)
    model.load_state_dict(oracle.state_dict(), strict=True)
    model.to(dev).train()
    pts = torch.from_numpy(clouds(6, B, 1024))
    rec = OL.Draws(record=True)
    nthreads = torch.get_num_threads()
    try:
        if B >= 64:
            torch.set_num_threads(min(32, nthreads))                 # (the oracle's step on every visible core of the GPU box is oversubscribed: bench.py)
        lo = oracle(pts, rec); lo.backward()
        with torch.no_grad():                                        # frozen-teacher features of the same batch, same prompt-dropout draws
            nb_o, c_o = oracle.group_divider(pts)
            tf_o = oracle.dvae_tokenizer.forward_tokenizer_features(nb_o, c_o, OL.Draws(rec.table))
    finally:
        torch.set_num_threads(nthreads)
    draws = Draws(rec.table, device=dev)
    lg = model(pts.to(dev), draws=draws); lg.backward()
    assert abs(lg.item() - lo.item()) <= TOL, (lg.item(), lo.item())
    with torch.no_grad():
        nb_g, c_g = model.group_divider(pts.to(dev))
        assert torch.equal(nb_g.cpu(), nb_o) and torch.equal(c_g.cpu(), c_o)          # Group is bit-exact
        tf_g = model.dvae_tokenizer.forward_tokenizer_features(nb_g, c_g, draws=Draws(rec.table, device=dev))
    assert _rel(tf_g, tf_o) <= TOL, _rel(tf_g, tf_o)
    od = dict(oracle.named_parameters())
    for n in ["ACT_encoder.blocks.blocks.11.mlp.fc1.weight", "ACT_encoder.encoder.first_conv.0.weight", "mask_token",
              "ACT_decoder.blocks.0.attn.qkv.weight", "proj_head.bias", "ACT_encoder.pos_embed.2.weight"]:
        _grad_close(dict(model.named_parameters())[n].grad, od[n].grad, n)


def test_stage1_tiny_golden(dev):
    from act_amd.models import build_model_from_cfg
    from act_amd.utils.config import EasyDict
    from act_amd.utils.draws import Draws
    g = golden("g7_stage1")
    torch.manual_seed(0)
    cfg = EasyDict(TINY_STAGE2["dvae_config"]); cfg.NAME = "ACTPromptedDiscreteVAEwithVIT"
    vae = fill_module(build_model_from_cfg(cfg), "g7.").to(dev).train()
    vae.prompt_dropout.p = 0.0
    pts = torch.from_numpy(clouds(4, TINY_B, TINY_N)).to(dev)
    ret = vae(pts, temperature=0.7, hard=False, draws=Draws({"gumbel": _gumbel_noise((TINY_B, 16, 64))}, device=dev))
    assert _rel(ret[2], g["coarse"]) <= TOL and _rel(ret[3], g["fine"]) <= TOL and _rel(ret[5], g["logits"]) <= TOL
    assert _rel(ret[1], g["whole_fine"]) <= TOL
    lr, lk = vae.get_loss(ret, pts)
    assert abs(lr.item() - g["loss"][0]) <= TOL and abs(lk.item() - g["loss"][1]) <= TOL
    (lr + 0.1 * lk).backward()
    pd = dict(vae.named_parameters())
    for n, v in zip(g["grad_names"], g["grad_norms"]):
        # gradient NORMS of the reference golden at 5e-4: this graph is ill-conditioned in fp32 -- the reference's own math (the CPU
        # oracle) run with 1 vs 8 threads, i.e. nothing but another fp32 summation order, moves these gradients by 1.5e-4 .. 3.5e-4
        # in ||e||/||r|| (tests/test_oracle_golden.py::test_stage1_tiny_gradient_conditioning; ReLU / LeakyReLU / max / arg-min
        # switch points within rounding distance).  Element-wise the HIP gradients are within 1e-4 of the oracle's (below).
        assert abs(pd[str(n)].grad.norm().item() - v) <= 5e-4 * max(1.0, v), (n, pd[str(n)].grad.norm().item(), v)
    # ---- element-wise gradients against the oracle.  The discrete selections of this graph (ReLU / LeakyReLU kinks, max over points and neighbours, Chamfer
    # arg-min) sit within rounding distance of a switch on most inputs: over twelve (cloud, noise) seeds the HIP gradients are either within ~3e-6 of the
    # oracle's (every selection agrees) or off by 2e-4 .. 6e-3 in the rows one differing selection reroutes -- with the erff GELU of rounds 1-5 on 8 of the 12
    # seeds, with the one-polynomial GELU of round 6 on 6 of them, and not the same ones (profiles/r06_stage1_tiny_grad_seeds.txt,
    # benchmarks/diag/stage1_tiny_grad_diff.py).  A 1e-4 check on ONE seed therefore tests the luck of that seed (rounds 1-5: seed 777 at 9.2e-5).  Instead:
    # on the golden's seed and on three more, every gradient must agree in the flip-tolerant sense (||e|| / ||r|| <= 5e-3: the reference arithmetic's own
    # spread, test_oracle_golden.py::test_stage1_tiny_gradient_conditioning), and on at least ONE of the seeds every selection must agree -- where the bar is
    # 1e-5, ten times tighter than the old check.
    from oracle import models as OM, layers as OL

    def seeded_noise(seed):
        torch.manual_seed(seed)
        return -torch.empty((TINY_B, 16, 64)).exponential_().log()

    def hip_and_oracle_grads(cloud_seed, noise_seed):
        p_ = torch.from_numpy(clouds(cloud_seed, TINY_B, TINY_N)).to(dev)
        vae.zero_grad(set_to_none=True)
        r_ = vae(p_, temperature=0.7, hard=False, draws=Draws({"gumbel": seeded_noise(noise_seed)}, device=dev))
        a_, b_ = vae.get_loss(r_, p_); (a_ + 0.1 * b_).backward()
        torch.manual_seed(0)
        ora = fill_module(OM.ACTPromptedDiscreteVAEwithVIT(OM.edict(cfg)), "g7.").train(); ora.prompt_p = 0.0
        ro = ora(p_.cpu(), OL.Draws({"gumbel": seeded_noise(noise_seed)}), temperature=0.7, hard=False)
        lo = ora.get_loss(ro); (lo[0] + 0.1 * lo[1]).backward()
        assert abs(a_.item() - lo[0].item()) <= TOL and abs(b_.item() - lo[1].item()) <= TOL
        return {n: q.grad.detach().double().cpu() for n, q in vae.named_parameters() if q.grad is not None}, \
               {n: q.grad.detach().double() for n, q in ora.named_parameters() if q.grad is not None}

    flip_free = 0
    for cloud_seed, noise_seed in ((4, 777), (6, 779), (9, 782), (13, 786)):
        hg, og = hip_and_oracle_grads(cloud_seed, noise_seed)
        common = [n for n in hg if n in og]
        assert len(common) >= 60
        worst = 0.0
        for n in common:
            l2 = ((hg[n] - og[n]).norm() / og[n].norm().clamp_min(1e-30)).item()
            assert l2 <= 5e-3 or _rel(hg[n], og[n]) <= TOL, (noise_seed, n, "L2", l2, "max", _rel(hg[n], og[n]))
            worst = max(worst, _rel(hg[n], og[n]))
        print(f"[stage1 tiny] seed {noise_seed}: worst element-wise deviation {worst:.2e}" + ("  (every selection agrees)" if worst <= 1e-5 else "  (a selection differs)"))
        flip_free += worst <= 1e-5
    assert flip_free >= 1, "no seed on which HIP and oracle take the same discrete selections: element-wise agreement unverified"


def test_plain_dvae_golden_and_oracle(dev):
    """g15: `DiscreteVAE` (models/dvae.py:278-358) -- Stage-I forward with soft gumbel, both losses, every gradient, and the frozen
    tokenizer features (hard gumbel) against the reference golden and, element-wise, against the oracle."""
    from act_amd.models import build_model_from_cfg
    from act_amd.utils.config import EasyDict
    from act_amd.utils.draws import Draws
    from tests.golden.fill import TINY_DVAE
    g = golden("g15_dvae")
    torch.manual_seed(15)
    vae = fill_module(build_model_from_cfg(EasyDict(dict(TINY_DVAE))), "g15.").to(dev).train()
    assert type(vae).__name__ == "DiscreteVAE" and sorted(vae.state_dict().keys()) == [str(k) for k in g["state_dict_keys"]]
    pts = torch.from_numpy(clouds(15, TINY_B, TINY_N)).to(dev)
    ret = vae(pts, temperature=0.7, hard=False, draws=Draws({"gumbel": _gumbel_noise((TINY_B, 16, 64))}, device=dev))
    assert _rel(ret[2], g["coarse"]) <= TOL and _rel(ret[3], g["fine"]) <= TOL and _rel(ret[5], g["logits"]) <= TOL
    assert _rel(ret[1], g["whole_fine"]) <= TOL and _rel(ret[0], g["whole_coarse"]) <= TOL
    lr, lk = vae.get_loss(ret, pts)
    assert abs(lr.item() - g["loss"][0]) <= TOL and abs(lk.item() - g["loss"][1]) <= TOL
    (lr + 0.1 * lk).backward()
    pd = dict(vae.named_parameters())
    for n, v in zip(g["grad_names"], g["grad_norms"]):
        assert abs(pd[str(n)].grad.norm().item() - v) <= 5e-4 * max(1.0, v), (n, pd[str(n)].grad.norm().item(), v)   # conditioning: see test_stage1_tiny_golden
    with torch.no_grad():
        nb, c = vae.group_divider(pts)
        feat = vae.forward_tokenizer_features(nb, c, draws=Draws({"gumbel": _gumbel_noise((TINY_B, 16, 64))}, device=dev))
    assert _rel(feat, g["tokenizer_feat"]) <= TOL
    from oracle import models as OM, layers as OL
    torch.manual_seed(15)
    ora = fill_module(OM.DiscreteVAE(OM.edict(TINY_DVAE)), "g15.").train()
    ro = ora(pts.cpu(), OL.Draws({"gumbel": _gumbel_noise((TINY_B, 16, 64))}), temperature=0.7, hard=False)
    lo = ora.get_loss(ro); (lo[0] + 0.1 * lo[1]).backward()
    od = dict(ora.named_parameters())
    checked = 0
    for n, p in pd.items():
        if p.grad is not None and od[n].grad is not None:
            # tiny tensors: ONE flipped selection (a max over k, a LeakyReLU kink) reroutes a 2,304-wide row = 0.2 % of dgcnn_2.layer5's
            # weight gradient; the oracle run with 1 vs 8 threads differs from itself in exactly these elements
            # (benchmarks/diag/dvae_tiny_grad_diff.py: HIP == oracle at 1 thread everywhere within 1e-4)
            _grad_close(p.grad, od[n].grad, n, frac=5e-3)
            checked += 1
    assert checked >= 50


@pytest.mark.parametrize("tag", ["shallow", "noprompt", "novit"])
def test_prompt_variants_golden_and_oracle(dev, tag):
    """g16: shallow prompts / no prompts (frozen Transformer under no_grad) / no image Transformer (models/dvae.py:513-534)."""
    from act_amd.models import build_model_from_cfg
    from act_amd.utils.config import EasyDict
    from act_amd.utils.draws import Draws
    from tests.golden.fill import PROMPT_VARIANTS
    g = golden("g16_prompt_variants")
    cfg = dict(TINY_STAGE2["dvae_config"]); cfg.update(PROMPT_VARIANTS[tag]); cfg["NAME"] = "ACTPromptedDiscreteVAEwithVIT"
    torch.manual_seed(16)
    vae = fill_module(build_model_from_cfg(EasyDict(cfg)), f"g16.{tag}.").to(dev).train()
    if hasattr(vae, "prompt_dropout"):
        vae.prompt_dropout.p = 0.0
    assert sorted(vae.state_dict().keys()) == [str(k) for k in g[f"{tag}.state_dict_keys"]]
    pts = torch.from_numpy(clouds(16, TINY_B, TINY_N)).to(dev)
    ret = vae(pts, temperature=0.7, hard=False, draws=Draws({"gumbel": _gumbel_noise((TINY_B, 16, 64))}, device=dev))
    assert _rel(ret[3], g[f"{tag}.fine"]) <= TOL and _rel(ret[5], g[f"{tag}.logits"]) <= TOL
    lr, lk = vae.get_loss(ret, pts)
    assert abs(lr.item() - g[f"{tag}.loss"][0]) <= TOL and abs(lk.item() - g[f"{tag}.loss"][1]) <= TOL
    (lr + 0.1 * lk).backward()
    pd = dict(vae.named_parameters())
    assert sorted(n for n, p in pd.items() if p.grad is not None) == [str(n) for n in g[f"{tag}.grad_names"]]
    for n, v in zip(g[f"{tag}.grad_names"], g[f"{tag}.grad_norms"]):
        assert abs(pd[str(n)].grad.norm().item() - v) <= 5e-4 * max(1.0, v), (n, pd[str(n)].grad.norm().item(), v)
    with torch.no_grad():
        nb, c = vae.group_divider(pts)
        feat = vae.forward_tokenizer_features(nb, c, draws=Draws({"gumbel": _gumbel_noise((TINY_B, 16, 64))}, device=dev))
    assert _rel(feat, g[f"{tag}.tokenizer_feat"]) <= TOL
    # every gradient against the oracle, measured in units of the reference's own fp32 noise: the oracle run with 1 and with N CPU threads
    # (another summation order, nothing else) moves these gradients by up to 1e-3 in L2 on the `noprompt` graph (large activations behind the
    # un-prompted Transformer, BatchNorm + ReLU switch points in the folding decoder); the HIP gradient must sit within 3x that spread of one of
    # the two runs, or within 5e-3 in L2 (flipped discrete decisions: benchmarks/diag/variant_grad_diff.py lists them)
    from oracle import models as OM, layers as OL
    ocfg = dict(cfg); ocfg.pop("NAME")
    runs = []
    nthreads = torch.get_num_threads()
    try:
        for nt in (1, max(2, nthreads)):
            torch.set_num_threads(nt)
            ora = fill_module(OM.ACTPromptedDiscreteVAEwithVIT(OM.edict(ocfg)), f"g16.{tag}.").train(); ora.prompt_p = 0.0
            ro = ora(pts.cpu(), OL.Draws({"gumbel": _gumbel_noise((TINY_B, 16, 64))}), temperature=0.7, hard=False)
            lo = ora.get_loss(ro); (lo[0] + 0.1 * lo[1]).backward()
            runs.append({n: q.grad.double() for n, q in ora.named_parameters() if q.grad is not None})
    finally:
        torch.set_num_threads(nthreads)
    l2 = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    for n, p in pd.items():
        if p.grad is None or runs[0][n].norm().item() < 1e-6:          # (conv biases in front of a BatchNorm: analytically zero)
            continue
        a = p.grad.double().cpu()
        spread = l2(runs[0][n], runs[1][n])
        # 5e-3: ONE flipped Chamfer arg-min (a tie within an ulp of the forward distances) among the 2 x 16 x (8 + 8) nearest-neighbour terms of this
        # tiny problem perturbs EVERY upstream gradient densely by ~1/256 of its norm; the forward values above agree to 1e-5
        assert min(l2(a, runs[0][n]), l2(a, runs[1][n])) <= max(5e-3, 3 * spread), (n, l2(a, runs[0][n]), l2(a, runs[1][n]), spread)


def test_no_host_sync_in_training_step(dev):
    """the Stage-II step must not synchronise with the host (reference: 5 boolean-index syncs + loss loop)."""
    model = _tiny(dev)
    pts = torch.from_numpy(clouds(4, TINY_B, TINY_N)).to(dev)
    model(pts).backward()                      # warm-up (allocations, workspace)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        model(pts).backward()
    finally:
        torch.cuda.set_sync_debug_mode("default")


def test_no_host_sync_in_stage1_and_finetune_steps(dev):
    """the Stage-I autoencoder step and the finetune step (FPS pool, random subset, rotation, CE loss, accuracy, clipping,
    AdamW) run without a device->host synchronisation as well (the reference: numpy-drawn index upload, two .item() per step)."""
    from act_amd.models import build_model_from_cfg
    from act_amd.tools import builder
    from act_amd.tools.runner_pretrain import _Single
    from act_amd.tools import runner_autoencoder as RA, runner_finetune as RF
    from act_amd.utils.config import EasyDict
    from tests.golden.fill import TINY_FINETUNE
    opt_cfg = dict(optimizer=dict(type="AdamW", kwargs=dict(lr=1e-3, weight_decay=0.05)),
                   scheduler=dict(type="CosLR", kwargs=dict(epochs=300, initial_epochs=10)), step_per_update=1)
    # Stage I
    mc = EasyDict(TINY_STAGE2["dvae_config"]); mc.NAME = "ACTPromptedDiscreteVAEwithVIT"
    vae = _Single(fill_module(build_model_from_cfg(mc), "g7.").to(dev).train())
    cfg1 = EasyDict(dict(opt_cfg, temp=dict(start=1, target=0.0625, ntime=100000), kldweight=dict(start=0, target=0.1, ntime=100000)))
    opt1, _ = builder.build_opti_sche(vae, cfg1)
    pts = torch.from_numpy(clouds(4, TINY_B, TINY_N)).to(dev)
    # finetune
    ft = _Single(fill_module(build_model_from_cfg(EasyDict(dict(TINY_FINETUNE, num_group=32, group_size=16))), "g10.full.").to(dev).train())
    cfg3 = EasyDict(dict(opt_cfg, npoints=1024, grad_norm_clip=10))
    opt3, _ = builder.build_opti_sche(ft, cfg3)
    raw = torch.from_numpy(clouds(5, 4, 2048)).to(dev); label = torch.tensor([1, 0, 3, 2], device=dev)
    for _ in range(2):                                   # warm-up (allocations, workspaces, first-use tuning)
        RA.train_step(vae, opt1, pts, cfg1, 20000)
        RF.train_step(ft, opt3, raw, label, cfg3, next_points=raw)
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        RA.train_step(vae, opt1, pts, cfg1, 20001)
        RF.train_step(ft, opt3, raw, label, cfg3, next_points=raw)
    finally:
        torch.cuda.set_sync_debug_mode("default")


@pytest.mark.parametrize("B", [1, 4])
def test_stress_geometry_vs_oracle(dev, B):
    """BASELINE configs[4] geometry (N=8192, 512 groups x 64 neighbours, 24-layer d=768 student, 104 / 512 / 576-token
    sequences) at B=1 and B=4 against the CPU oracle (reference: models/act.py:1203-1258): exercises the multi-wave FPS, 128-points-per-lane kNN,
    chunked (online-softmax) attention forward and the key/query-chunked attention backward; B=4 puts several clouds into the BatchNorm statistics
    (131,072 rows), the batched attention grids and the group-of-clouds GEMM tilings of the benchmarked launches.  Group bit-exact, loss and
    frozen-teacher features within 1e-4, five gradients across the graph."""
    from oracle import models as OM, layers as OL
    from act_amd.models import build_model_from_cfg
    from act_amd.utils.config import cfg_from_yaml_file
    from act_amd.utils.draws import Draws
    cfg = cfg_from_yaml_file("cfgs/pretrain/pretrain_act_distill.yaml").model
    cfg.dvae_config.ckpt = "none"
    cfg.transformer_config.embed_dim = 768; cfg.transformer_config.encoder_dims = 768; cfg.transformer_config.depth = 24
    cfg.transformer_config.num_heads = 12; cfg.transformer_config.decoder_num_heads = 12
    for k in ("encoder_dims", "tokens_dims", "decoder_dims"):
        cfg.dvae_config[k] = 768
    cfg.dvae_config.num_group = 512; cfg.dvae_config.group_size = 64
    torch.manual_seed(3)
    oracle = OM.ACT_PointDistillation(OM.edict(cfg)).train()
    model = build_model_from_cfg(cfg)
    model.load_state_dict(oracle.state_dict(), strict=True)
    model.to(dev).train()
    pts = torch.from_numpy(clouds(8, B, 8192))
    rec = OL.Draws(record=True)
    nthreads = torch.get_num_threads()
    try:
        torch.set_num_threads(min(32, nthreads))
        lo = oracle(pts, rec); lo.backward()
        with torch.no_grad():
            nb_o, c_o = oracle.group_divider(pts)
            tf_o = oracle.dvae_tokenizer.forward_tokenizer_features(nb_o, c_o, OL.Draws(rec.table))
    finally:
        torch.set_num_threads(nthreads)
    lg = model(pts.to(dev), draws=Draws(rec.table, device=dev)); lg.backward()
    assert abs(lg.item() - lo.item()) <= TOL, (lg.item(), lo.item())
    with torch.no_grad():
        nb_g, c_g = model.group_divider(pts.to(dev))
        assert torch.equal(nb_g.cpu(), nb_o) and torch.equal(c_g.cpu(), c_o)          # Group is bit-exact
        tf_g = model.dvae_tokenizer.forward_tokenizer_features(nb_g, c_g, draws=Draws(rec.table, device=dev))
    assert _rel(tf_g, tf_o) <= TOL, _rel(tf_g, tf_o)
    od = dict(oracle.named_parameters())
    for n in ["ACT_encoder.blocks.blocks.23.mlp.fc1.weight", "ACT_encoder.encoder.second_conv.0.weight", "mask_token",
              "ACT_decoder.blocks.1.attn.qkv.weight", "ACT_encoder.pos_embed.0.weight"]:
        _grad_close(dict(model.named_parameters())[n].grad, od[n].grad, n)


def test_forward_eval_cls_feature_vs_oracle(dev):
    """model(pts, noaug=True) -> cls feature [B, cls_dim] (models/act.py:1197-1201), no masking, no grad."""
    from oracle import models as OM, layers as OL
    torch.manual_seed(4)
    oracle = fill_module(OM.ACT_PointDistillation(OM.edict(TINY_STAGE2)), "ev.").eval()
    model = _tiny(dev, "ev.").eval()
    pts = torch.from_numpy(clouds(9, 3, TINY_N))
    fo = oracle(pts, OL.Draws(), noaug=True)
    fg = model(pts.to(dev), noaug=True)
    assert tuple(fg.shape) == (3, 32) and not fg.requires_grad
    assert _rel(fg, fo) <= TOL


def test_state_dict_keys_match_reference_listing(dev):
    """key groups of SURVEY 8(b): a reference checkpoint ({'base_model': sd}) must load strictly."""
    model = _tiny(dev)
    keys = set(model.state_dict())
    for k in ["mask_token", "ACT_encoder.cls_token", "ACT_encoder.cls_pos", "ACT_encoder.encoder.first_conv.0.weight",
              "ACT_encoder.encoder.first_conv.1.running_mean", "ACT_encoder.encoder.second_conv.3.bias", "ACT_encoder.pos_embed.2.weight",
              "ACT_encoder.blocks.blocks.1.norm1.weight", "ACT_encoder.blocks.blocks.1.attn.qkv.weight", "ACT_encoder.blocks.blocks.1.attn.proj.bias",
              "ACT_encoder.blocks.blocks.1.mlp.fc2.weight", "ACT_encoder.norm.bias", "ACT_encoder.lm_head.weight", "ACT_encoder.cls_head.2.bias",
              "dvae_tokenizer.codebook", "dvae_tokenizer.visual_prompt_token", "dvae_tokenizer.deep_prompt_pos", "dvae_tokenizer.encoder.first_conv.0.weight",
              "dvae_tokenizer.dgcnn_1.layer5.0.weight", "dvae_tokenizer.dgcnn_2.input_trans.bias", "dvae_tokenizer.decoder.final_conv.6.weight",
              "dvae_tokenizer.visual_embed.0.0.attn.qkv.bias", "dvae_tokenizer.visual_embed.1.weight", "dvae_tokenizer.proj_pre.weight",
              "dvae_tokenizer.visual_pos_embed.0.weight", "dvae_tokenizer.proj_post.bias", "proj_head.weight", "decoder_pos_embed.2.bias",
              "ACT_decoder.blocks.0.attn.qkv.weight", "ACT_decoder.norm.weight"]:
        assert k in keys, k
    assert not any("qkv.bias" in k for k in keys if k.startswith("ACT_encoder.") or k.startswith("ACT_decoder."))   # qkv_bias=False


def test_cross_step_teacher_prefetch_is_exact(dev):
    """Software pipelining across steps (next batch's grouping + frozen-teacher forward on the auxiliary stream during this
    batch's backward) must not change a single bit of the training trajectory: 4 AdamW steps, pipelined vs sequential."""
    import copy
    import argparse
    from act_amd.tools import builder
    from act_amd.tools.runner_pretrain import train_step, _Single, freeze_unused_heads
    from act_amd.utils.config import EasyDict
    cfg = EasyDict(optimizer=dict(type="AdamW", kwargs=dict(lr=1e-3, weight_decay=0.05)),
                   scheduler=dict(type="CosLR", kwargs=dict(epochs=300, initial_epochs=10)), step_per_update=1)
    batches = [torch.from_numpy(clouds(20 + i, TINY_B, TINY_N)).to(dev) for i in range(4)]
    results = []
    for pipelined in (False, True):
        model = _tiny(dev)
        model.dvae_tokenizer.prompt_dropout.p = 0.0          # prompt dropout shares the device RNG with the mask draws
        freeze_unused_heads(model)
        wrapped = _Single(model)
        opt, _ = builder.build_opti_sche(wrapped, cfg)
        torch.manual_seed(123)
        pts = [b.clone() for b in batches]
        losses = []
        for i in range(4):
            nxt = pts[i + 1] if (pipelined and i + 1 < 4) else None
            losses.append(train_step(wrapped, opt, pts[i], cfg, next_points=nxt))
            if pipelined and nxt is not None:
                assert model._prefetched is not None and model._prefetched[0] is nxt
        torch.cuda.synchronize()
        results.append((torch.stack(losses).cpu(), {n: p.detach().clone().cpu() for n, p in model.named_parameters()}))
    assert torch.equal(results[0][0], results[1][0]), (results[0][0], results[1][0])
    for n, p in results[0][1].items():
        assert torch.equal(p, results[1][1][n]), n


def test_visible_only_patch_embedding_is_exact(dev):
    """The student's Encoder computes its last conv + max-pool for the visible patches only (models/act.py:269-275 reads x[~bool_masked_pos]; every
    layer in front, BatchNorm statistics included, still runs on all patches): the 3-step training trajectory is bit-identical to running it on all."""
    from act_amd.tools import builder
    from act_amd.tools.runner_pretrain import train_step, _Single, freeze_unused_heads
    from act_amd.utils.config import EasyDict
    import act_amd.models.act as ACT
    cfg = EasyDict(optimizer=dict(type="AdamW", kwargs=dict(lr=1e-3, weight_decay=0.05)),
                   scheduler=dict(type="CosLR", kwargs=dict(epochs=300, initial_epochs=10)), step_per_update=1)
    import copy
    from act_amd.models import build_model_from_cfg
    mcfg = copy.deepcopy(TINY_STAGE2)                       # 32-point patches, 128-wide tokens: the geometry the fused patch embedding takes
    mcfg["dvae_config"].update(group_size=32, num_group=16)
    mcfg["transformer_config"].update(encoder_dims=128)
    batches = [torch.from_numpy(clouds(40 + i, 4, 512)).to(dev) for i in range(3)]
    results = []
    saved = ACT.NEED_VISIBLE_ONLY
    try:
        for visible_only in (False, True, False, True):     # (first pair: warm-up -- first-use GEMM tuning of either mode's shapes draws random numbers)
            ACT.NEED_VISIBLE_ONLY = visible_only
            torch.manual_seed(0)
            model = fill_module(build_model_from_cfg(EasyDict(mcfg)), "vis.").to(dev).train()
            model.dvae_tokenizer.prompt_dropout.p = 0.0
            freeze_unused_heads(model)
            wrapped = _Single(model)
            opt, _ = builder.build_opti_sche(wrapped, cfg)
            torch.manual_seed(77)
            losses = [train_step(wrapped, opt, b.clone(), cfg) for b in batches]
            torch.cuda.synchronize()
            results.append((torch.stack(losses).cpu(), {n: p.detach().clone().cpu() for n, p in model.named_parameters()}))
    finally:
        ACT.NEED_VISIBLE_ONLY = saved
    results = results[2:]
    assert torch.equal(results[0][0], results[1][0]), (results[0][0], results[1][0])
    for n, p in results[0][1].items():
        assert torch.equal(p, results[1][1][n]), n


def test_block_mask_type_matches_reference_and_trains(dev):
    """mask_type: block (models/act.py:215-242): device-side implementation == the reference's per-cloud loop (golden g12, injected
    seed indices); a Stage-II step with it runs and agrees with the oracle."""
    from act_amd.models import build_model_from_cfg
    from act_amd.utils.config import EasyDict
    from act_amd.utils.draws import Draws
    from oracle import models as OM
    from oracle import layers as OL
    import copy
    g = golden("g12_block_mask")
    cfg = copy.deepcopy(TINY_STAGE2); cfg["transformer_config"]["mask_type"] = "block"; cfg["transformer_config"]["mask_ratio"] = 0.8
    model = fill_module(build_model_from_cfg(EasyDict(cfg)), "g4.").to(dev).train()
    center = torch.from_numpy(golden("g1_group")["center"]).to(dev)
    m = model.ACT_encoder._mask_center_block(center, draws=Draws({"mask_seed": torch.from_numpy(g["seed_index"])}, device=dev))
    assert np.array_equal(m.cpu().numpy(), g["mask"])
    free = model.ACT_encoder._mask_center_block(center)
    assert free.dtype == torch.bool and (free.sum(1) == 51).all()
    # end to end against the oracle
    oracle = fill_module(OM.ACT_PointDistillation(OM.edict(cfg)), "g4.").train()
    oracle.dvae_tokenizer.prompt_p = 0.0; model.dvae_tokenizer.prompt_dropout.p = 0.0
    pts = torch.from_numpy(clouds(4, TINY_B, TINY_N))
    rec = OL.Draws({"mask_seed": torch.tensor([3, 11])}, record=True)
    lo = oracle(pts, rec)
    lg = model(pts.to(dev), draws=Draws(rec.table, device=dev))
    assert abs(lg.item() - lo.item()) <= TOL, (lg.item(), lo.item())


@pytest.mark.parametrize("B", [2, 128])
def test_stage1_full_geometry_vs_oracle(dev, B):
    """BASELINE configs[2] geometry from the shipped YAML (N=1024, G=64, M=32, 8192-token codebook, 12-layer ViT-B with 64 deep
    prompts through PrefixBlockFn, 64x32 FoldingNet, Chamfer-L1 + KL) at B=2 and at the BENCHMARKED batch B=128 (BatchNorm over 262,144 rows, the
    split-K / tile choices of the timed launches) against the CPU oracle with every draw replayed: both losses, the fine reconstruction ret[3] and the
    logits ret[5] within 1e-4, ten gradients through the noise-aware check (reference: models/dvae.py:594-615 forward, :450-478 losses)."""
    from oracle import models as OM, layers as OL
    from act_amd.models import build_model_from_cfg
    from act_amd.utils.config import cfg_from_yaml_file
    from act_amd.utils.draws import Draws
    cfg = cfg_from_yaml_file("cfgs/autoencoder/act_dvae_with_pretrained_transformer.yaml").model
    torch.manual_seed(5)
    oracle = OM.ACTPromptedDiscreteVAEwithVIT(OM.edict(cfg)).train()
    vae = build_model_from_cfg(cfg)
    vae.load_state_dict(oracle.state_dict(), strict=True)
    vae.to(dev).train()
    pts = torch.from_numpy(clouds(11, B, 1024))
    rec = OL.Draws(record=True)
    nthreads = torch.get_num_threads()
    # two oracle runs = two fp32 summation orders of the reference's own math: B=2 on all threads and on one; B=128 thread-capped (every visible core of
    # the GPU box oversubscribes the intra-op pool: bench.py) at 32 and at 12 threads
    th_a, th_b = (min(32, nthreads), max(1, min(12, nthreads // 2))) if B >= 64 else (nthreads, 1)
    try:
        torch.set_num_threads(th_a)
        ro = oracle(pts, rec, temperature=0.6, hard=False)
        lro, lko = oracle.get_loss(ro)
        (lro + 0.05 * lko).backward()
    finally:
        torch.set_num_threads(nthreads)
    assert "gumbel" in rec.table and any(k.startswith("prompt.") for k in rec.table)
    rg = vae(pts.to(dev), temperature=0.6, hard=False, draws=Draws(rec.table, device=dev))
    lrg, lkg = vae.get_loss(rg, pts.to(dev))
    (lrg + 0.05 * lkg).backward()
    assert tuple(rg[0].shape) == (B, 512, 3) and tuple(rg[1].shape) == (B, 2048, 3) and tuple(rg[5].shape) == (B, 64, 8192)
    assert abs(lrg.item() - lro.item()) <= TOL and abs(lkg.item() - lko.item()) <= TOL, (lrg.item(), lro.item(), lkg.item(), lko.item())
    assert _rel(rg[3], ro[3]) <= TOL and _rel(rg[5], ro[5]) <= TOL
    od, pd = dict(oracle.named_parameters()), dict(vae.named_parameters())
    names = ["encoder.first_conv.0.weight", "dgcnn_1.layer5.0.weight", "codebook", "deep_prompt_tokens", "visual_prompt_pos",
             "proj_pre.weight", "dgcnn_2.layer3.0.weight", "decoder.mlp.2.weight", "decoder.final_conv.3.weight", "proj_post.bias"]
    # The reference's own fp32 noise on this graph: the same oracle step with another thread count (another summation order, nothing else).  Where
    # a HIP gradient is not element-wise within 1e-4 of the oracle (flipped max / ReLU / arg-min decisions reroute whole gradient rows), it must
    # be as close to one of the two oracle runs, in L2, as 3x their distance from each other -- or pass the flipped-element count.
    first = {n: od[n].grad.double().clone() for n in names}
    del ro
    try:
        torch.set_num_threads(th_b)
        oracle.zero_grad(set_to_none=True)
        r1 = oracle(pts, OL.Draws(rec.table), temperature=0.6, hard=False)
        l1 = oracle.get_loss(r1); (l1[0] + 0.05 * l1[1]).backward()
    finally:
        torch.set_num_threads(nthreads)
    l2 = lambda a, b: ((a - b).norm() / b.norm().clamp_min(1e-30)).item()
    for n in names:
        assert od[n].grad is not None and pd[n].grad is not None, n
        a, second = pd[n].grad.double().cpu(), od[n].grad.double()
        spread = l2(first[n], second)
        if min(l2(a, first[n]), l2(a, second)) <= 3 * spread:
            if (a - first[n]).abs().max() > TOL * max(1.0, first[n].abs().max().item()):
                print(f"[reference-noise] {n}: L2 to the oracle {min(l2(a, first[n]), l2(a, second)):.2e}, oracle {th_b} vs {th_a} threads {spread:.2e}")
            continue
        _grad_close(pd[n].grad, first[n], n)
    assert all(p.grad is None for n, p in pd.items() if n.startswith("visual_embed."))     # frozen Transformer: no dW


def test_rand_mask_has_exact_count_per_row(dev):
    """_mask_center_rand (models/act.py:244-267): exactly int(mask_ratio * G) ones per cloud, different per cloud, on the device."""
    from act_amd.models.act import random_mask
    m = random_mask(128, 64, int(0.8 * 64), dev)
    assert m.dtype == torch.bool and tuple(m.shape) == (128, 64)
    assert (m.sum(1) == 51).all()
    assert len({tuple(r.tolist()) for r in m.cpu()}) > 100
    model = _tiny(dev)
    center = torch.from_numpy(golden("g1_group")["center"]).to(dev)[:, :16].contiguous()
    mk = model.ACT_encoder._mask_center_rand(center)
    assert (mk.sum(1) == int(0.75 * 16)).all()
    assert not model.ACT_encoder._mask_center_rand(center, noaug=True).any()


def test_cls_loss_branch_golden_and_oracle(dev):
    """transformer_config.cls_loss: True (models/act.py:1208-1249): golden g13 from the reference's own code (loss + gradient norms), and
    every gradient element-wise against the oracle with DropPath active (second decoder pass draws its own gates)."""
    import copy
    from oracle import models as OM, layers as OL
    from act_amd.models import build_model_from_cfg
    from act_amd.utils.config import EasyDict
    from act_amd.utils.draws import Draws
    g = golden("g13_cls_loss")
    cfg = copy.deepcopy(TINY_STAGE2)
    cfg["transformer_config"].update(depth=3, cls_loss=True, register_shallow_hook=1)
    torch.manual_seed(0)
    model = fill_module(build_model_from_cfg(EasyDict(cfg)), "g13.").to(dev).train()
    model.dvae_tokenizer.prompt_dropout.p = 0.0
    pts = torch.from_numpy(clouds(13, TINY_B, TINY_N))
    loss = model(pts.to(dev), draws=Draws({"mask": torch.from_numpy(g["mask"]), "gumbel": _gumbel_noise((TINY_B, 16, 64))}, device=dev))
    loss.backward()
    assert abs(loss.item() - g["loss"][0]) <= TOL
    pd = dict(model.named_parameters())
    for n, v in zip(g["grad_names"], g["grad_norms"]):
        assert abs(pd[str(n)].grad.norm().item() - v) <= TOL * max(1.0, v), n
    assert _rel(pd["cls_pos"].grad, g["grad_cls_pos"]) <= TOL
    assert sorted(k for k in model.state_dict() if not k.startswith("dvae_tokenizer.")) == [str(k) for k in g["state_dict_keys"]]
    # DropPath on: oracle records, HIP path replays
    cfg["transformer_config"]["drop_path_rate"] = 0.3
    torch.manual_seed(1)
    oracle = fill_module(OM.ACT_PointDistillation(OM.edict(cfg)), "g13dp.").train()
    m2 = build_model_from_cfg(EasyDict(cfg)); m2.load_state_dict(oracle.state_dict(), strict=True); m2.to(dev).train()
    rec = OL.Draws(record=True)
    lo = oracle(pts, rec); lo.backward()
    assert any(k.startswith("dec_shallow.") for k in rec.table)
    lg = m2(pts.to(dev), draws=Draws(rec.table, device=dev)); lg.backward()
    assert abs(lg.item() - lo.item()) <= TOL
    od = dict(oracle.named_parameters())
    for n, p in m2.named_parameters():
        if p.requires_grad and od[n].grad is not None and p.grad is not None:
            assert _rel(p.grad, od[n].grad) <= TOL, (n, _rel(p.grad, od[n].grad))
    with pytest.raises(Exception):
        bad = copy.deepcopy(cfg); bad["transformer_config"]["register_shallow_hook"] = -1
        build_model_from_cfg(EasyDict(bad))


def test_reference_format_checkpoints_load_strictly(dev, tmp_path):
    """SURVEY 8(f)4 / tools/builder.py:97-173, models/act.py:1153-1156: files in the reference's exact container
    ({'base_model', 'optimizer', 'epoch', 'metrics', 'best_metrics'}, keys carrying DDP's 'module.' prefix) written from the oracle's
    state_dict load STRICTLY through (a) dvae_config.ckpt of ACT_PointDistillation (Stage-I -> Stage-II hand-over), (b) builder.load_model,
    (c) builder.resume_model, and the loaded product models reproduce the oracle's outputs."""
    import argparse
    import copy
    from oracle import models as OM, layers as OL
    from act_amd.models import build_model_from_cfg
    from act_amd.tools import builder
    from act_amd.utils.config import EasyDict
    from act_amd.utils.draws import Draws
    torch.manual_seed(11)
    dcfg = dict(TINY_STAGE2["dvae_config"]); dcfg["NAME"] = "ACTPromptedDiscreteVAEwithVIT"
    o_vae = fill_module(OM.ACTPromptedDiscreteVAEwithVIT(OM.edict(dcfg)), "ck.vae.")
    dvae_path = str(tmp_path / "ckpt_act_dvae.pth")
    torch.save({"base_model": {"module." + k: v for k, v in o_vae.state_dict().items()}, "optimizer": {}, "epoch": 299,
                "metrics": {"CDL1": 1.0}, "best_metrics": {"CDL1": 0.9}}, dvae_path)
    # (a) Stage II built with dvae_config.ckpt pointing at the Stage-I file
    cfg = copy.deepcopy(TINY_STAGE2); cfg["dvae_config"]["ckpt"] = dvae_path
    student = build_model_from_cfg(EasyDict(cfg))
    for k, v in o_vae.state_dict().items():
        assert torch.equal(student.dvae_tokenizer.state_dict()[k].cpu(), v), k
    assert not any(p.requires_grad for p in student.dvae_tokenizer.parameters())
    # (b) + (c) a Stage-II ckpt-last.pth in the reference's format
    o_s2 = fill_module(OM.ACT_PointDistillation(OM.edict(TINY_STAGE2)), "ck.s2.")
    exp = tmp_path / "exp"; exp.mkdir()
    torch.save({"base_model": {"module." + k: v for k, v in o_s2.state_dict().items()}, "optimizer": {"state": {}, "param_groups": []},
                "epoch": 41, "metrics": {"acc": 0.0}, "best_metrics": {"acc": 12.5}}, str(exp / "ckpt-last.pth"))
    m_load = build_model_from_cfg(EasyDict(TINY_STAGE2))
    builder.load_model(m_load, str(exp / "ckpt-last.pth"))
    m_res = build_model_from_cfg(EasyDict(TINY_STAGE2))
    start, best = builder.resume_model(m_res, argparse.Namespace(experiment_path=str(exp), local_rank=0))
    assert start == 42 and best == {"acc": 12.5}
    bad = {"base_model": {k: v for k, v in list(o_s2.state_dict().items())[:-1]}}
    torch.save(bad, str(tmp_path / "bad.pth"))
    with pytest.raises(RuntimeError):                       # strict: a missing key is an error, as in the reference
        builder.load_model(build_model_from_cfg(EasyDict(TINY_STAGE2)), str(tmp_path / "bad.pth"))
    # the loaded product model computes what the oracle computes
    o_s2.train(); o_s2.dvae_tokenizer.prompt_p = 0.0
    pts = torch.from_numpy(clouds(17, TINY_B, TINY_N))
    rec = OL.Draws(record=True)
    lo = o_s2(pts, rec)
    for m in (m_load, m_res):
        m.to(dev).train(); m.dvae_tokenizer.prompt_dropout.p = 0.0
        assert abs(m(pts.to(dev), draws=Draws(rec.table, device=dev)).item() - lo.item()) <= TOL


def test_act_pointbert_golden_and_oracle(dev):
    """ACT_PointBERT (models/act.py:913-1096): golden g14 from the reference's own forward/backward (three losses, gradient norms,
    queue update, momentum update), every gradient element-wise against the oracle, one runner step (tuple loss summed), noaug feature."""
    import copy
    from oracle import models as OM, layers as OL
    from act_amd.models import build_model_from_cfg, MODELS
    from act_amd.tools import builder
    from act_amd.tools.runner_pretrain import train_step, _Single
    from act_amd.utils.config import EasyDict
    from act_amd.utils.draws import Draws
    from tests.golden.fill import TINY_POINTBERT
    from tests.test_oracle_golden import _pointbert_draws, _pointbert_oracle
    assert "ACT_PointBERT" in MODELS
    g = golden("g14_pointbert")
    oracle = _pointbert_oracle(g)
    torch.manual_seed(3)
    with pytest.warns(UserWarning):
        model = build_model_from_cfg(EasyDict(copy.deepcopy(TINY_POINTBERT)))
    assert sorted(k for k in model.state_dict() if not k.startswith("dvae.")) == [str(k) for k in g["state_dict_keys"]]
    model.load_state_dict(oracle.state_dict(), strict=True)
    model.to(dev).train()
    assert not any(p.requires_grad for p in model.transformer_k.parameters()) and not any(p.requires_grad for p in model.dvae.parameters())
    pts = torch.from_numpy(clouds(14, 4, 128))
    lo = oracle(pts, _pointbert_draws(g, OL.Draws)); sum(lo).backward()
    lg = model(pts.to(dev), draws=_pointbert_draws(g, Draws, device=dev)); (lg[0] + lg[1] + lg[2]).backward()
    for got, ora, want in zip(lg, lo, g["losses"]):
        assert abs(got.item() - want) <= TOL * max(1.0, abs(want)) and abs(got.item() - ora.item()) <= TOL, (got.item(), ora.item(), want)
    pd, od = dict(model.named_parameters()), dict(oracle.named_parameters())
    for n, v in zip(g["grad_names"], g["grad_norms"]):
        assert abs(pd[str(n)].grad.norm().item() - v) <= TOL * max(1.0, v), n
    for n, p in pd.items():
        if p.grad is not None and od[n].grad is not None:
            assert _rel(p.grad, od[n].grad) <= TOL, (n, _rel(p.grad, od[n].grad))
    assert _rel(model.queue, g["queue1"]) <= 1e-5 and int(model.queue_ptr) == int(g["queue_ptr"][0])
    assert abs(pd["transformer_k.blocks.blocks.1.mlp.fc1.weight"].norm().item() - g["key_norm_after"][0]) <= 1e-5
    # eval feature + one optimisation step through the runner (free-running draws, tuple loss)
    f = model(pts.to(dev), noaug=True)
    assert tuple(f.shape) == (4, 32) and _rel(f, oracle(pts, OL.Draws(), noaug=True)) <= TOL
    cfg = EasyDict(optimizer=dict(type="AdamW", kwargs=dict(lr=1e-3, weight_decay=0.05)),
                   scheduler=dict(type="CosLR", kwargs=dict(epochs=300, initial_epochs=10)), step_per_update=1)
    wrapped = _Single(model)
    opt, _ = builder.build_opti_sche(wrapped, cfg)
    before = pd["transformer_q.lm_head.weight"].detach().clone()
    loss = train_step(wrapped, opt, pts.to(dev).clone(), cfg)
    assert torch.isfinite(loss).all() and not torch.equal(before, pd["transformer_q.lm_head.weight"].detach())
    assert int(model.queue_ptr) == 8


def test_mask_ratio_zero_regresses_every_token(dev):
    """transformer_config.mask_ratio: 0 (models/act.py:1175-1178,1238-1240): no mask token / decoder, the student's 64 tokens are
    regressed onto the teacher's directly; state_dict has no decoder keys; loss + gradients vs the oracle."""
    import copy
    from oracle import models as OM, layers as OL
    from act_amd.models import build_model_from_cfg
    from act_amd.utils.config import EasyDict
    from act_amd.utils.draws import Draws
    cfg = copy.deepcopy(TINY_STAGE2); cfg["transformer_config"]["mask_ratio"] = 0; cfg["loss"] = "l2"   # the reference's cosine branch is broken here
    torch.manual_seed(6)
    oracle = fill_module(OM.ACT_PointDistillation(OM.edict(cfg)), "nm.").train()
    model = build_model_from_cfg(EasyDict(cfg))
    assert not any(k.startswith(("mask_token", "ACT_decoder", "decoder_pos_embed")) for k in model.state_dict())
    model.load_state_dict(oracle.state_dict(), strict=True)
    model.to(dev).train()
    pts = torch.from_numpy(clouds(19, TINY_B, TINY_N))
    rec = OL.Draws(record=True)
    lo = oracle(pts, rec); lo.backward()
    lg = model(pts.to(dev), draws=Draws(rec.table, device=dev)); lg.backward()
    assert abs(lg.item() - lo.item()) <= TOL
    od = dict(oracle.named_parameters())
    for n, p in model.named_parameters():
        if p.requires_grad and od[n].grad is not None and p.grad is not None:
            assert _rel(p.grad, od[n].grad) <= TOL, n
    from act_amd.tools.runner_pretrain import train_step, _Single, freeze_unused_heads
    from act_amd.tools import builder
    freeze_unused_heads(model)
    ocfg = EasyDict(optimizer=dict(type="AdamW", kwargs=dict(lr=1e-3, weight_decay=0.05)),
                    scheduler=dict(type="CosLR", kwargs=dict(epochs=300, initial_epochs=10)), step_per_update=1)
    w = _Single(model); opt, _ = builder.build_opti_sche(w, ocfg)
    assert torch.isfinite(train_step(w, opt, pts.to(dev).clone(), ocfg, next_points=pts.to(dev).clone()))
