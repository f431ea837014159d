"""GPU: the two runner entry points (run_net, same signature as the reference's tools/runner_*.py) train for a few steps on
the synthetic dataset, write the reference's checkpoint format, resume from it, and the Stage-II loss goes down."""
import argparse
import copy
import os

import pytest
import torch

from tests.golden.fill import TINY_STAGE2

pytestmark = pytest.mark.gpu


def _args(tmp, **kw):
    a = argparse.Namespace(log_name="test", use_gpu=True, local_rank=0, distributed=False, sync_bn=False, resume=False,
                           start_ckpts=None, experiment_path=str(tmp), num_workers=0, world_size=1, val_freq=1)
    a.__dict__.update(kw)
    return a


def _config(model_cfg, bs=8, npoints=128, max_epoch=1):
    from act_amd.utils.config import EasyDict
    ds = lambda subset: dict(_base_=dict(NAME="ShapeNet", N_POINTS=8192, SYNTHETIC=True, NUM_SAMPLES=32, DATA_PATH="none", PC_PATH="none"),
                             others=dict(subset=subset, npoints=npoints, bs=bs))
    return EasyDict(optimizer=dict(type="AdamW", kwargs=dict(lr=1e-3, weight_decay=0.05)),
                    scheduler=dict(type="CosLR", kwargs=dict(epochs=300, initial_epochs=10)),
                    dataset=dict(train=ds("train"), val=ds("test")), model=copy.deepcopy(model_cfg), total_bs=bs, step_per_update=1,
                    max_epoch=max_epoch, consider_metric="CDL1", temp=dict(start=1, target=0.0625, ntime=100000),
                    kldweight=dict(start=0, target=0.1, ntime=100000))


def test_stage2_run_net_trains_checkpoints_and_resumes(tmp_path):
    from act_amd.tools.runner_pretrain import run_net
    torch.manual_seed(0)
    cfg = _config(TINY_STAGE2, max_epoch=2)
    log = run_net(_args(tmp_path), cfg, log_every=1)
    assert len(log) == 12 and all(torch.isfinite(torch.tensor(log)))
    assert sum(log[-4:]) / 4 < sum(log[:4]) / 4                     # the cosine distillation loss goes down
    ck = torch.load(os.path.join(tmp_path, "ckpt-last.pth"), map_location="cpu")
    assert set(ck) == {"base_model", "optimizer", "epoch", "metrics", "best_metrics"} and ck["epoch"] == 2
    assert "ACT_encoder.blocks.blocks.0.attn.qkv.weight" in ck["base_model"] and "dvae_tokenizer.codebook" in ck["base_model"]
    cfg2 = _config(TINY_STAGE2, max_epoch=3)
    log2 = run_net(_args(tmp_path, resume=True), cfg2, log_every=1)
    assert len(log2) == 4                                            # only epoch 3 is left


def test_stage1_run_net_and_schedules(tmp_path):
    from act_amd.tools.runner_autoencoder import run_net, get_temp, kld_weight
    from act_amd.utils.config import EasyDict
    torch.manual_seed(0)
    mc = dict(TINY_STAGE2["dvae_config"]); mc["NAME"] = "ACTPromptedDiscreteVAEwithVIT"
    cfg = _config(mc, max_epoch=1)
    log = run_net(_args(tmp_path), cfg, log_every=1)
    assert len(log) == 8 and all(l1 > 0 for l1, _ in log)
    assert abs(get_temp(cfg, 0) - 1.0) < 1e-9 and abs(get_temp(cfg, 100001) - 0.0625) < 1e-9
    assert abs(get_temp(cfg, 50000) - (0.0625 + (1 - 0.0625) * 0.5)) < 1e-6
    assert kld_weight(cfg, 5000) == 0.0 and abs(kld_weight(cfg, 10000 + 100000) - 0.1) < 1e-9
    assert os.path.exists(os.path.join(tmp_path, "ckpt-last.pth"))


def test_plain_dvae_run_net(tmp_path):
    """the pointbert_dvae recipe (reference cfgs/autoencoder/pointbert_dvae.yaml, model DiscreteVAE) through the same runner: the
    reconstruction loss goes down over two epochs and the checkpoint has the reference's container."""
    from act_amd.tools.runner_autoencoder import run_net
    from tests.golden.fill import TINY_DVAE
    torch.manual_seed(0)
    cfg = _config(dict(TINY_DVAE), max_epoch=3)
    log = run_net(_args(tmp_path), cfg, log_every=1)
    rec = [l1 for l1, _ in log]
    assert len(rec) >= 12 and all(r > 0 for r in rec) and sum(rec[-4:]) < sum(rec[:4])
    ck = torch.load(os.path.join(tmp_path, "ckpt-last.pth"), map_location="cpu")
    assert set(ck) == {"base_model", "optimizer", "epoch", "metrics", "best_metrics"} and "codebook" in ck["base_model"]
    assert not any(k.startswith("visual_embed") or "prompt" in k for k in ck["base_model"])


def test_cosine_lr_schedule_matches_timm_formula():
    from act_amd.tools.builder import CosineLRScheduler
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=1e-3)
    s = CosineLRScheduler(opt, t_initial=300, lr_min=1e-7, warmup_lr_init=1e-6, warmup_t=10, cycle_limit=1, t_in_epochs=True)
    assert abs(opt.param_groups[0]["lr"] - 1e-6) < 1e-12
    s.step(5); assert abs(opt.param_groups[0]["lr"] - (1e-6 + 5 * (1e-3 - 1e-6) / 10)) < 1e-12
    import math
    s.step(150); assert abs(opt.param_groups[0]["lr"] - (1e-7 + 0.5 * (1e-3 - 1e-7) * (1 + math.cos(math.pi * 150 / 300)))) < 1e-12
    s.step(300); assert abs(opt.param_groups[0]["lr"] - 1e-7) < 1e-12
