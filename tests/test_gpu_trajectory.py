"""Twenty-step training-trajectory parity, HIP path against the CPU oracle (missing #5 / next #7).

The two-step golden (G8, tests/test_gpu_model.py::test_stage2_tiny_golden_loss_grads_adamw) cannot see drift that builds up in optimizer state or in
BatchNorm running statistics.  Here the tiny Stage-II graph is trained for 20 optimizer steps on 20 DIFFERENT batches with every random draw of every
step recorded in the oracle and replayed in the product (mask, gumbel, DropPath 0.3, prompt dropout 0.1), AdamW(lr 1e-3, wd 0.05) with the reference's
two parameter groups (tools/builder.py:38-51) -- the oracle through torch.optim.AdamW, the product through its own runner step
(runner_pretrain.train_step -> builder.FusedAdamW, block stack, cross-step teacher prefetch).  Reference loop: tools/runner_pretrain.py:123-149."""
import copy

import pytest
import torch

from tests.golden.fill import fill_module, clouds, TINY_STAGE2, TINY_N

pytestmark = pytest.mark.gpu
STEPS = 20
TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _augmented(seed, B):
    """a batch after PointcloudScaleAndTranslate (datasets/data_transforms.py:20-34) with seeded factors -- applied on the host so that both sides train on
    the same tensor (the device augmentation kernel has its own parity tests: g9, tests/test_gpu_point_ops.py)"""
    pts = torch.from_numpy(clouds(seed, B, TINY_N))
    g = torch.Generator().manual_seed(10_000 + seed)
    scale = torch.empty(B, 1, 3).uniform_(2. / 3., 3. / 2., generator=g)
    shift = torch.empty(B, 1, 3).uniform_(-0.2, 0.2, generator=g)
    return pts * scale + shift


@pytest.mark.parametrize("prefetch", [True, False])
def test_twenty_step_trajectory_vs_oracle(dev, prefetch):
    from oracle import models as OM, layers as OL
    from act_amd.models import build_model_from_cfg
    from act_amd.tools import builder
    from act_amd.tools.runner_pretrain import train_step, _Single, freeze_unused_heads
    from act_amd.utils.config import EasyDict
    from act_amd.utils.draws import Draws
    B = 4
    cfg = copy.deepcopy(TINY_STAGE2)
    cfg["transformer_config"]["drop_path_rate"] = 0.3
    torch.manual_seed(3)
    oracle = fill_module(OM.ACT_PointDistillation(OM.edict(cfg)), "traj.").train()
    model = build_model_from_cfg(EasyDict(copy.deepcopy(cfg)))
    model.load_state_dict(oracle.state_dict(), strict=True)
    model.to(dev).train()
    for m in (oracle, model):                                # the never-used heads: frozen on both sides (no gradient -> AdamW skips them either way)
        freeze_unused_heads(m)
    opt_o = torch.optim.AdamW(OM.param_groups(oracle, 0.05), lr=1e-3, weight_decay=0.05)
    rcfg = EasyDict(optimizer=dict(type="AdamW", kwargs=dict(lr=1e-3, weight_decay=0.05)),
                    scheduler=dict(type="CosLR", kwargs=dict(epochs=300, initial_epochs=10)), step_per_update=1)
    wrapped = _Single(model)
    opt_g, _ = builder.build_opti_sche(wrapped, rcfg)
    assert [len(g["params"]) for g in opt_g.param_groups] == [len(g["params"]) for g in opt_o.param_groups]
    for g in opt_g.param_groups:                             # (the scheduler's epoch-0 warm-up value is the runner's business: same constant lr on both sides)
        g["lr"] = 1e-3

    # A Conv1d bias in front of a BatchNorm has a mathematically ZERO gradient (the normalisation removes it); what AdamW sees there is fp32 rounding noise,
    # which it normalises to steps of up to lr -- in the reference as much as here.  Those biases random-walk differently on the two sides and shift the BN
    # input mean with them, so for the two student BatchNorms the running MEAN is compared after removing exactly that (recorded) contribution.
    BIASED = {"ACT_encoder.encoder.first_conv.1.running_mean": "ACT_encoder.encoder.first_conv.0.bias",
              "ACT_encoder.encoder.second_conv.1.running_mean": "ACT_encoder.encoder.second_conv.0.bias"}
    hist_o = {b: [] for b in BIASED.values()}; hist_g = {b: [] for b in BIASED.values()}
    batches = [_augmented(300 + k, B) for k in range(STEPS)]
    lo, tables = [], []
    for k in range(STEPS):
        for bn in hist_o:
            hist_o[bn].append(dict(oracle.named_parameters())[bn].detach().clone())
        rec = OL.Draws(record=True)
        loss = oracle(batches[k], rec)
        loss.backward()
        opt_o.step(); opt_o.zero_grad()
        lo.append(loss.item()); tables.append(rec.table)
    assert any(key.startswith("enc.1") for key in tables[0]) and any(key.startswith("prompt.") for key in tables[0])

    dbat = [b.to(dev) for b in batches]
    lg = []
    pgm = dict(model.named_parameters())
    for k in range(STEPS):
        for bn in hist_g:
            hist_g[bn].append(pgm[bn].detach().cpu().clone())
        nxt = dbat[k + 1] if (prefetch and k + 1 < STEPS) else None
        # (the look-ahead teacher forward of batch k+1 consumes the NEXT step's teacher draws: gumbel, prompt dropout)
        lg.append(train_step(wrapped, opt_g, dbat[k], rcfg, next_points=nxt, augment=False, draws=Draws(tables[k], device=dev),
                             next_draws=Draws(tables[k + 1], device=dev) if nxt is not None else None))
    lg = torch.stack(lg).tolist()
    worst = max(abs(a - b) / (TOL * (1 + k / 5)) for k, (a, b) in enumerate(zip(lg, lo)))
    print(f"[trajectory] prefetch={prefetch}: loss {lo[0]:.6f} -> {lo[-1]:.6f}; worst |HIP - oracle| / (1e-4 (1 + k/5)) = {worst:.3f}")
    for k, (a, b) in enumerate(zip(lg, lo)):
        assert abs(a - b) <= TOL * (1 + k / 5), (k, a, b)
    assert lo[-1] < lo[0]                                    # it trains

    # after step 20: parameter norms (the three of golden G8 + the decoder and the prediction head), every BatchNorm running statistic and the step counters
    po, pg = dict(oracle.named_parameters()), dict(model.named_parameters())
    for n in ("ACT_encoder.blocks.blocks.0.attn.qkv.weight", "ACT_encoder.encoder.first_conv.0.weight", "mask_token",
              "ACT_decoder.blocks.1.mlp.fc2.weight", "ACT_encoder.pos_embed.2.weight"):
        a, b = pg[n].detach().norm().item(), po[n].detach().norm().item()
        assert abs(a - b) <= 5 * TOL * max(1.0, b), (n, a, b)
    bo, bg = dict(oracle.named_buffers()), dict(model.named_buffers())
    n_stats = 0
    for n, t in bo.items():
        if n.endswith("num_batches_tracked"):
            assert int(bg[n].item()) == int(t.item()), n
        elif n.endswith("running_mean") or n.endswith("running_var"):
            diff = bg[n].detach().cpu().double() - t.double()
            if n in BIASED:                                  # EMA (momentum 0.1) of the bias difference the statistic saw at each step
                exp = torch.zeros_like(diff)
                for k in range(STEPS):
                    exp = 0.9 * exp + 0.1 * (hist_g[BIASED[n]][k].double() - hist_o[BIASED[n]][k].double())
                print(f"[trajectory] {n}: raw difference {diff.abs().max().item():.2e}, of which the zero-gradient conv bias explains {exp.abs().max().item():.2e}")
                diff = diff - exp
            d = diff.abs().max().item()
            assert d <= 5 * TOL * max(1.0, t.abs().max().item()), (n, d)
            n_stats += 1
    assert n_stats >= 4                                      # student mini-PointNet (2 BN) + the train-mode teacher's
    so = opt_o.state_dict()["state"]; sg = opt_g.state_dict()["state"]
    assert all(int(sg[k]["step"].item() if torch.is_tensor(sg[k]["step"]) else sg[k]["step"]) == STEPS for k in sg) and len(sg) == len(so)
