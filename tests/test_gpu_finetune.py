"""GPU parity of the finetune / inference path (PointTransformer, models/act.py:727-910 +
tools/runner_finetune.py) against the golden produced by the reference's own classes (g10) and against the CPU oracle with
DropPath / head-dropout draws replayed."""
import argparse
import copy
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.conftest import golden
from tests.golden.fill import fill_module, clouds, TINY_FINETUNE, TINY_FT_LABELS

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _rel(a, ref):
    a = torch.as_tensor(a).detach().double().cpu(); ref = torch.as_tensor(ref).detach().double().cpu()
    assert a.shape == ref.shape, (a.shape, ref.shape)
    return ((a - ref).abs().max() / max(1.0, ref.abs().max())).item()


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("R,C", [(4, 10), (32, 40), (257, 1000)])
def test_softmax_xent_fwd_bwd_acc(dev, R, C):
    import act_amd.kernels as K
    g = torch.Generator().manual_seed(R * 7 + C)
    z = (torch.randn(R, C, generator=g) * 3).requires_grad_(True)
    lab = torch.randint(0, C, (R,), generator=g)
    z.data[0, :] = 0.0                                                  # an all-ties row: arg-max = index 0
    ref = F.cross_entropy(z, lab)
    ref.backward()
    zd = z.detach().to(dev).requires_grad_(True)
    loss, acc = K.softmax_xent(zd, lab.to(dev))
    (loss * 2.5).backward()
    assert abs(loss.item() - ref.item()) <= 1e-5 * max(1.0, abs(ref.item()))
    assert _rel(zd.grad / 2.5, z.grad) <= 1e-6
    assert abs(acc.item() - (z.detach().argmax(-1) == lab).float().mean().item()) < 1e-7


def test_rotate_matches_reference_transform(dev):
    from act_amd.datasets.data_transforms import PointcloudRotate
    g = golden("g10_finetune")
    pc = torch.from_numpy(clouds(11, 3, 64)).to(dev)
    out = PointcloudRotate()(pc.clone(), u=torch.from_numpy(g["rotate_u"]))
    assert _rel(out, g["rotate_out"]) <= 1e-6


@pytest.mark.parametrize("ttype", ["full", "linear", "side"])
def test_point_transformer_golden(dev, ttype):
    from act_amd.models import build_model_from_cfg
    from act_amd.utils.config import EasyDict
    g = golden("g10_finetune")
    model = fill_module(build_model_from_cfg(EasyDict(dict(TINY_FINETUNE, transfer_type=ttype))), f"g10.{ttype}.").to(dev)
    if ttype == "side":
        with torch.no_grad():
            model.side_alpha.fill_(0.3)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    pts = torch.from_numpy(clouds(10, 4, 128)).to(dev)
    label = torch.tensor(TINY_FT_LABELS, device=dev)
    model.train()
    logits = model(pts)
    loss, acc = model.get_loss_acc(logits, label)
    assert _rel(logits, g[f"{ttype}_logits_train"]) <= TOL
    assert abs(loss.item() - float(g[f"{ttype}_loss"])) <= TOL
    assert float(acc) == float(g[f"{ttype}_acc"])
    if ttype != "linear":
        names = [str(n) for n in g[f"{ttype}_trainable"]]
        assert sorted(names) == sorted(n for n, p in model.named_parameters() if p.requires_grad)
        loss.backward()
        prm = dict(model.named_parameters())
        norms = np.array([prm[n].grad.norm().item() for n in names])
        np.testing.assert_allclose(norms, g[f"{ttype}_grad_norms"], rtol=2e-3, atol=2e-5)   # conv biases before BN: exact zeros + noise
    if ttype == "full":
        bn = model.cls_head_finetune[1]
        assert _rel(bn.running_mean, g["full_head_bn_running_mean"]) <= TOL
        assert _rel(bn.running_var, g["full_head_bn_running_var"]) <= TOL
    model.eval()
    with torch.no_grad():
        assert _rel(model(pts), g[f"{ttype}_logits_eval"]) <= TOL


def test_point_transformer_vs_oracle_with_dropout_and_droppath(dev):
    """B=8, N=256, drop_path 0.2, head dropout 0.5: all draws recorded on the oracle and replayed on the HIP path."""
    from act_amd.models import build_model_from_cfg
    from act_amd.utils.config import EasyDict
    from act_amd.utils.draws import Draws
    from oracle import models as OM
    from oracle.layers import Draws as ODraws
    cfg = dict(TINY_FINETUNE, drop_path_rate=0.2, depth=3, num_group=32, group_size=16, encoder_dims=64)
    ref = fill_module(OM.PointTransformer(OM.edict(cfg)), "ft.").train()
    pts = torch.from_numpy(clouds(12, 8, 256))
    label = torch.tensor([0, 9, 2, 2, 5, 7, 1, 3])
    torch.manual_seed(5)
    od = ODraws(record=True)
    ref_logits = ref(pts, od)
    ref_loss, ref_acc = ref.get_loss_acc(ref_logits, label)
    ref_loss.backward()
    assert "head.drop1" in od.table and any(k.endswith(".attn") for k in od.table)
    model = fill_module(build_model_from_cfg(EasyDict(cfg)), "ft.").to(dev).train()
    logits = model(pts.to(dev), draws=Draws(od.table, device=dev))
    loss, acc = model.get_loss_acc(logits, label.to(dev))
    loss.backward()
    assert _rel(logits, ref_logits) <= TOL
    assert abs(loss.item() - ref_loss.item()) <= TOL and float(acc) == float(ref_acc)
    rp = dict(ref.named_parameters())
    for n, p in model.named_parameters():
        # Frobenius-relative: the graph has discrete switch points (max-pool arg-max, ReLU kink); an element within rounding
        # distance of one moves a handful of gradient entries by ~1e-3 between summation orders, never the bulk
        # (+2e-5 absolute: conv biases in front of BatchNorm have exactly-zero gradients, both sides hold rounding noise)
        d = (p.grad.double().cpu() - rp[n].grad.double()).norm().item()
        assert d <= 5e-4 * rp[n].grad.double().norm().item() + 2e-5, (n, d)


def test_subsample_pool_is_fps_prefix_and_bit_exact_vs_oracle(dev, oracle_c):
    """8192 -> 1200 FPS pool of the finetune loop (tools/runner_finetune.py:152-155) equals the C oracle's indices."""
    import ctypes
    from act_amd.tools.runner_finetune import subsample
    pts = clouds(13, 2, 8192)
    ref = np.empty((2, 1200), dtype=np.int32)
    assert oracle_c.oracle_fps_f32(pts.ctypes.data_as(ctypes.c_void_p), 2, 8192, 1200, ref.ctypes.data_as(ctypes.c_void_p), 0) == 0
    choice = np.random.RandomState(0).choice(1200, 1024, False)
    out, fps_idx = subsample(torch.from_numpy(pts).to(dev), 1024, 1200, choice=choice)
    assert np.array_equal(fps_idx.cpu().numpy(), ref)
    np.testing.assert_array_equal(out.cpu().numpy(), np.take_along_axis(pts, ref[:, choice][..., None].astype(np.int64), axis=1))


def _args(tmp, **kw):
    a = argparse.Namespace(log_name="test_ft", use_gpu=True, local_rank=0, distributed=False, sync_bn=False, resume=False,
                           start_ckpts=None, ckpts=None, experiment_path=str(tmp), num_workers=0, world_size=1, val_freq=1,
                           vote=False)
    a.__dict__.update(kw)
    return a


def _config(bs=8, max_epoch=2):
    from act_amd.utils.config import EasyDict
    ds = lambda subset: dict(_base_=dict(NAME="ModelNet", N_POINTS=2048, NUM_CATEGORY=4, USE_NORMALS=False, SYNTHETIC=True,
                                         NUM_SAMPLES=48, DATA_PATH="none"), others=dict(subset=subset, bs=bs))
    mc = dict(TINY_FINETUNE, cls_dim=4, num_group=32, group_size=16)
    return EasyDict(optimizer=dict(type="AdamW", kwargs=dict(lr=2e-3, weight_decay=0.05)),
                    scheduler=dict(type="CosLR", kwargs=dict(epochs=300, initial_epochs=0)),
                    dataset=dict(train=ds("train"), val=ds("test"), test=ds("test")), model=mc, npoints=1024, total_bs=bs,
                    step_per_update=1, max_epoch=max_epoch, grad_norm_clip=10)


def test_finetune_run_net_validate_vote_and_test(tmp_path, dev):
    from act_amd.tools import runner_finetune as RF
    from act_amd.tools import builder
    torch.manual_seed(0); np.random.seed(0)
    cfg = _config(max_epoch=5)
    log = RF.run_net(_args(tmp_path), cfg, log_every=2)
    assert len(log) == 6 * 6 and all(np.isfinite(l) for l, _ in log)
    assert np.mean([l for l, _ in log[-6:]]) < np.mean([l for l, _ in log[:6]])          # cross-entropy goes down
    ck = torch.load(os.path.join(tmp_path, "ckpt-last.pth"), map_location="cpu")
    assert set(ck) == {"base_model", "optimizer", "epoch", "metrics", "best_metrics"}
    assert "cls_head_finetune.8.weight" in ck["base_model"] and set(ck["metrics"]) == {"acc", "acc_avg"}
    m, best_vote = RF.test_net(_args(tmp_path, ckpts=os.path.join(tmp_path, "ckpt-last.pth")), cfg, vote_rounds=1)
    assert 0.0 <= m.acc <= 100.0 and 0.0 <= best_vote <= 100.0
    assert np.mean([a for _, a in log[-6:]]) > 40.0                      # train accuracy: 4 learnable classes, chance = 25 %


def test_accuracy_scores_match_sklearn_semantics():
    from act_amd.tools.runner_finetune import accuracy_scores
    label = torch.tensor([0, 0, 0, 1, 1, 3])
    pred = torch.tensor([0, 0, 1, 1, 0, 3])
    acc, bal = accuracy_scores(label, pred)
    assert abs(acc - 100 * 4 / 6) < 1e-4
    assert abs(bal - 100 * (2 / 3 + 1 / 2 + 1) / 3) < 1e-4              # class 2 is absent from the labels: not averaged


def test_pretrained_student_checkpoint_loads_into_point_transformer(tmp_path, dev):
    """ckpt-last.pth of Stage II ('ACT_encoder.*' keys) -> PointTransformer.load_model_from_ckpt (models/act.py:832-870)."""
    from act_amd.models import build_model_from_cfg
    from act_amd.utils.config import EasyDict
    from tests.golden.fill import TINY_STAGE2
    s2 = build_model_from_cfg(EasyDict(TINY_STAGE2))
    path = os.path.join(tmp_path, "s2.pth")
    torch.save({"base_model": {"module." + k: v for k, v in s2.state_dict().items()}}, path)
    ft = build_model_from_cfg(EasyDict(dict(TINY_FINETUNE, encoder_dims=64)))
    ft.load_model_from_ckpt(path)
    sd = s2.state_dict()
    assert torch.equal(ft.blocks.blocks[1].mlp.fc1.weight, sd["ACT_encoder.blocks.blocks.1.mlp.fc1.weight"])
    assert torch.equal(ft.encoder.first_conv[0].weight, sd["ACT_encoder.encoder.first_conv.0.weight"])
    assert torch.equal(ft.cls_pos, sd["ACT_encoder.cls_pos"])


def test_modelnet_cache_is_built_with_the_hip_fps_kernel(tmp_path, dev):
    """first use of a ModelNet split without a cache: every shape is reduced to N_POINTS by the HIP farthest-point sampler
    (start index 0) and the reference's pickle cache is written."""
    import pickle
    from act_amd.datasets import build_dataset_from_cfg
    from act_amd.utils.config import EasyDict
    from oracle import point_ops as OP
    root = tmp_path / "mn"
    (root / "chair").mkdir(parents=True); (root / "night_stand").mkdir()
    (root / "modelnet40_shape_names.txt").write_text("chair\nnight_stand\n")
    (root / "modelnet40_train.txt").write_text("chair_0001\nnight_stand_0007\n")
    (root / "modelnet40_test.txt").write_text("chair_0001\n")
    rs = np.random.RandomState(3)
    raw = {}
    for shape, sid in (("chair", "chair_0001"), ("night_stand", "night_stand_0007")):
        a = rs.standard_normal((500, 6)).astype(np.float32)
        raw[sid] = a
        np.savetxt(root / shape / f"{sid}.txt", a, delimiter=",", fmt="%.6f")
    base = EasyDict(NAME="ModelNet", DATA_PATH=str(root), N_POINTS=128, NUM_CATEGORY=40, USE_NORMALS=False)
    ds = build_dataset_from_cfg(base, EasyDict(subset="train"))
    assert os.path.exists(root / "modelnet40_train_128pts_fps.dat") and len(ds) == 2
    with open(root / "modelnet40_train_128pts_fps.dat", "rb") as f:
        pts, labels = pickle.load(f)
    assert [int(l[0]) for l in labels] == [0, 1] and pts[1].shape == (128, 6)
    loaded = np.loadtxt(root / "night_stand" / "night_stand_0007.txt", delimiter=",").astype(np.float32)
    want = OP.fps_ref(np.ascontiguousarray(loaded[None, :, :3]), 128)[0]
    assert np.array_equal(pts[1], loaded[want])                          # bit-exact selection against the oracle
    assert ds[1][2][1] == 1 and ds[1][2][0].shape == (128, 3)
