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

    # Gauge direction.  A Conv1d bias in front of a train-mode BatchNorm has a mathematically ZERO gradient (the normalisation removes it); what AdamW sees
    # there is fp32 rounding noise, which it normalises to steps of up to lr -- in the reference as much as here.  first_conv.0.bias random-walks differently on
    # the two sides and moves the input mean of BN1 with it, nothing else: that running mean is compared after removing exactly the recorded contribution
    # (validated on the oracle alone: 1 vs 8 threads differ by 2.4e-3 raw, 3.5e-8 after the correction).  The statistics of BN2 sit behind further such
    # directions (first_conv.3.bias, columns of second_conv.0.weight that see per-group constants); there the bar is the reference arithmetic's OWN spread:
    # the oracle is run a second time with another thread count (same draws) and the HIP path must stay within 3x that spread (or 5e-4, whichever is larger).
    E = "ACT_encoder.encoder."
    GAUGE = (E + "first_conv.0.bias",)
    hist_o = {b: [] for b in GAUGE}; hist_g = {b: [] for b in GAUGE}
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
    # the reference arithmetic's own spread: the same trajectory (same weights, batches, draws) with a different intra-op thread count
    nthr = torch.get_num_threads()
    torch.set_num_threads(1 if nthr > 1 else 2)
    try:
        torch.manual_seed(3)
        oracle2 = fill_module(OM.ACT_PointDistillation(OM.edict(cfg)), "traj.").train()
        freeze_unused_heads(oracle2)
        opt_2 = torch.optim.AdamW(OM.param_groups(oracle2, 0.05), lr=1e-3, weight_decay=0.05)
        hist_2 = {b: [] for b in GAUGE}
        lo2 = []
        for k in range(STEPS):
            for bn in hist_2:
                hist_2[bn].append(dict(oracle2.named_parameters())[bn].detach().clone())
            loss = oracle2(batches[k], OL.Draws(tables[k]))
            loss.backward()
            opt_2.step(); opt_2.zero_grad()
            lo2.append(loss.item())
    finally:
        torch.set_num_threads(nthr)

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
    def stat_diff(bufs, hist, n):
        """bufs[n] - oracle's, with the zero-gradient bias contribution (EMA, momentum 0.1, of the recorded bias difference) removed for BN1's mean"""
        diff = bufs[n].detach().cpu().double() - bo[n].double()
        if n == E + "first_conv.1.running_mean":
            exp = torch.zeros_like(diff)
            for k in range(STEPS):
                exp = 0.9 * exp + 0.1 * (hist[GAUGE[0]][k].double() - hist_o[GAUGE[0]][k].double())
            diff = diff - exp
        return diff.abs().max().item()
    b2 = dict(oracle2.named_buffers())
    report, bad = [], []
    for n, t in bo.items():
        if n.endswith("num_batches_tracked"):
            assert int(bg[n].item()) == int(t.item()), n
        elif n.endswith("running_mean") or n.endswith("running_var"):
            scale = max(1.0, t.abs().max().item())
            d, spread = stat_diff(bg, hist_g, n), stat_diff(b2, hist_2, n)
            report.append(f"{n}: |HIP - oracle| {d:.2e}; oracle vs oracle (other thread count) {spread:.2e}; scale {scale:.2e}")
            # BN2 sits behind several zero-gradient directions at once (first_conv.3.bias, second_conv.0.bias, the columns of second_conv.0.weight that see
            # per-group constants: 1 vs 8 oracle threads move them by 3e-4 .. 5e-4 in 20 steps, and BN2's mean with them by 8e-4 .. 3.5e-3): on top of the
            # measured spread a fixed allowance of 5e-3 of the statistic's scale -- a wrong momentum or variance correction would be off by 1e-1
            allow = 5e-3 * scale if n.startswith(E + "second_conv.1.") else 0.0
            if d > max(5 * TOL * scale, 3 * spread, allow):
                bad.append(report[-1])
            n_stats += 1
    print("[trajectory] oracle vs oracle loss spread %.2e\n[trajectory] " % max(abs(x - y) for x, y in zip(lo, lo2)) + "\n[trajectory] ".join(report))
    assert not bad, bad
    so = opt_o.state_dict()["state"]; sg = opt_g.state_dict()["state"]
    assert all(int(sg[k]["step"].item() if torch.is_tensor(sg[k]["step"]) else sg[k]["step"]) == STEPS for k in sg) and len(sg) == len(so)
