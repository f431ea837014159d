"""CPU: host-side logic of the finetune / inference path that needs no GPU (config surface, registry, synthetic labelled
dataset, accuracy metrics, FPS-pool table of tools/runner_finetune.py:141-150)."""
import os
import numpy as np
import pytest
import torch


def test_finetune_yaml_surface_and_registry():
    from act_amd.models import MODELS
    from act_amd.utils.config import cfg_from_yaml_file
    assert "PointTransformer" in MODELS
    for path, ttype in [("cfgs/finetune_classification/full/finetune_modelnet.yaml", "full"),
                        ("cfgs/finetune_classification/linear/finetune_modelnet_linear.yaml", "linear"),
                        ("cfgs/finetune_classification/mlp3/finetune_modelnet_mlp3.yaml", "mlp-3")]:
        cfg = cfg_from_yaml_file(path)
        assert cfg.model.NAME == "PointTransformer" and cfg.model.transfer_type == ttype
        assert (cfg.npoints, cfg.total_bs, cfg.grad_norm_clip, cfg.model.num_group, cfg.model.group_size) == (1024, 32, 10, 64, 32)
        assert cfg.dataset.train._base_.NAME == "ModelNet" and cfg.dataset.train._base_.N_POINTS == 8192


def test_synthetic_modelnet_item_contract():
    from act_amd.datasets import build_dataset_from_cfg
    from act_amd.utils.config import EasyDict
    ds = build_dataset_from_cfg(EasyDict(NAME="ModelNet", N_POINTS=512, NUM_CATEGORY=7, SYNTHETIC=True, NUM_SAMPLES=5, DATA_PATH="none"),
                                EasyDict(subset="train"))
    assert len(ds) == 5
    tax, mid, (pts, label) = ds[3]
    assert (tax, mid) == ("ModelNet", "sample") and pts.shape == (512, 3) and pts.dtype == torch.float32 and 0 <= label < 7
    assert abs(pts.norm(dim=1).max().item() - 1.0) < 1e-5 and pts.mean(0).abs().max().item() < 1e-5       # pc_norm semantics
    assert torch.equal(ds[3][2][0], pts)                                                                   # deterministic per index


def test_accuracy_scores_and_fps_pool_table():
    from act_amd.tools.runner_finetune import accuracy_scores, point_all_for, Acc_Metric
    label = torch.tensor([0, 0, 0, 1, 1, 3]); pred = torch.tensor([0, 0, 1, 1, 0, 3])
    acc, bal = accuracy_scores(label, pred)
    assert abs(acc - 100 * 4 / 6) < 1e-4 and abs(bal - 100 * (2 / 3 + 1 / 2 + 1) / 3) < 1e-4
    assert [point_all_for(n) for n in (1024, 2048, 4096, 8192)] == [1200, 2400, 4800, 8192]
    with pytest.raises(NotImplementedError):
        point_all_for(512)
    with pytest.raises(NotImplementedError):
        point_all_for(2048, train=False)                     # the reference's vote loop has no 2048 branch (:311-318)
    assert Acc_Metric(91.0, 88.0).better_than(Acc_Metric({"acc": 90.5, "acc_avg": 89.0}))
    assert Acc_Metric(Acc_Metric(1.0, 2.0)).state_dict() == {"acc": 1.0, "acc_avg": 2.0}


def test_point_transformer_freezing_rules_match_reference_names():
    from act_amd.models import build_model_from_cfg
    from act_amd.utils.config import EasyDict
    from tests.conftest import golden
    from tests.golden.fill import TINY_FINETUNE
    g = golden("g10_finetune")
    for ttype in ("full", "side"):
        m = build_model_from_cfg(EasyDict(dict(TINY_FINETUNE, transfer_type=ttype)))
        assert sorted(n for n, p in m.named_parameters() if p.requires_grad) == sorted(str(n) for n in g[f"{ttype}_trainable"])
    m = build_model_from_cfg(EasyDict(dict(TINY_FINETUNE, transfer_type="bit-fit")))
    assert all(("bias" in n or "cls" in n) == p.requires_grad for n, p in m.named_parameters())


def _edict(**kw):
    from act_amd.utils.config import EasyDict
    return EasyDict(kw)


def test_shapenet55_file_loader_contract(tmp_path):
    """file-backed ShapeNet55 (datasets/ShapeNet55Dataset.py:9-70): list file, <taxonomy>-<model>.npy, random subset, pc_norm."""
    from act_amd.datasets import build_dataset_from_cfg
    root, pcs = tmp_path / "ShapeNet-55", tmp_path / "shapenet_pc"
    root.mkdir(); pcs.mkdir()
    rs = np.random.RandomState(0)
    names = ["02691156-aaaa1111.npy", "03001627-bbbb2222.npy", "04379243-cccc3333.npy"]
    for n in names:
        np.save(pcs / n, (rs.standard_normal((8192, 3)) * 3 + 1).astype(np.float32))
    (root / "train.txt").write_text("\n".join(names[:2]) + "\n")
    (root / "test.txt").write_text(names[2] + "\n")
    base = _edict(NAME="ShapeNet", N_POINTS=8192, DATA_PATH=str(root), PC_PATH=str(pcs))
    ds = build_dataset_from_cfg(base, _edict(subset="train", npoints=1024))
    assert len(ds) == 2
    tax, mid, pts = ds[1]
    assert (tax, mid) == ("03001627", "bbbb2222") and pts.shape == (1024, 3) and pts.dtype == torch.float32
    assert abs(pts.norm(dim=1).max().item() - 1.0) < 1e-5 and pts.mean(0).abs().max().item() < 1e-5
    full = np.load(pcs / names[1])
    sub = ds[1][2]                                                      # a different random subset of the same cloud
    assert not torch.equal(sub, pts)
    whole = build_dataset_from_cfg(base, _edict(subset="train", npoints=1024, whole=True))
    assert len(whole) == 3 and whole[0][0] == "04379243"                # test list first, like the reference
    with pytest.raises(ValueError):
        from act_amd.datasets.SyntheticDataset import read_points
        read_points(str(tmp_path / "cloud.pcd"))


def test_modelnet_file_loader_from_cache(tmp_path):
    """file-backed ModelNet40 (datasets/ModelNetDataset.py:52-149) from a processed-data cache in the reference's pickle format."""
    import pickle
    from act_amd.datasets import build_dataset_from_cfg
    root = tmp_path / "modelnet40_normal_resampled"
    root.mkdir()
    names = [f"class{i:02d}" for i in range(40)]
    (root / "modelnet40_shape_names.txt").write_text("\n".join(names) + "\n")
    ids = ["class03_0001", "class17_0042", "class03_0002"]
    (root / "modelnet40_train.txt").write_text("\n".join(ids) + "\n")
    (root / "modelnet40_test.txt").write_text(ids[1] + "\n")
    rs = np.random.RandomState(1)
    pts = [np.concatenate((rs.standard_normal((64, 3)) * 2 + 5, rs.standard_normal((64, 3))), 1).astype(np.float32) for _ in ids]
    labels = [np.array([3], np.int32), np.array([17], np.int32), np.array([3], np.int32)]
    with open(root / "modelnet40_train_64pts_fps.dat", "wb") as f:
        pickle.dump([pts, labels], f)
    base = _edict(NAME="ModelNet", DATA_PATH=str(root), N_POINTS=64, NUM_CATEGORY=40, USE_NORMALS=False)
    ds = build_dataset_from_cfg(base, _edict(subset="train"))
    assert len(ds) == 3
    tax, mid, (p, label) = ds[1]
    assert (tax, mid, label) == ("ModelNet", "sample", 17) and p.shape == (64, 3) and p.dtype == torch.float32
    assert abs(p.norm(dim=1).max().item() - 1.0) < 1e-5                 # xyz normalised, normals dropped
    ref = pts[1][:, :3] - pts[1][:, :3].mean(0); ref = ref / np.sqrt((ref ** 2).sum(1)).max()
    assert np.allclose(np.sort(p.numpy(), axis=0), np.sort(ref, axis=0), atol=1e-6)      # train: same points, shuffled
    dsn = build_dataset_from_cfg(_edict(**dict(base, USE_NORMALS=True)), _edict(subset="train"))
    assert dsn[0][2][0].shape == (64, 6)


def test_state_dict_listing_equals_the_reference(tmp_path):
    """checkpoint wire compatibility: names AND shapes of every parameter / buffer equal the reference's own models
    (g11_state_dict.npz, generated by importing the reference classes), so {'base_model': sd} checkpoints interchange."""
    from act_amd.models import build_model_from_cfg
    from act_amd.utils.config import EasyDict
    from tests.conftest import golden
    from tests.golden.fill import TINY_STAGE2, TINY_FINETUNE
    g = golden("g11_state_dict")

    def listing(m):
        return {k: "x".join(str(d) for d in v.shape) for k, v in m.state_dict().items()}

    def ref(prefix):
        return dict(zip((str(n) for n in g[prefix + "_names"]), (str(s) for s in g[prefix + "_shapes"])))

    mc = dict(TINY_STAGE2["dvae_config"]); mc["NAME"] = "ACTPromptedDiscreteVAEwithVIT"
    assert listing(build_model_from_cfg(EasyDict(mc))) == ref("dvae")
    s2 = build_model_from_cfg(EasyDict(TINY_STAGE2))
    assert listing(s2) == ref("stage2")
    for ttype in ("full", "linear", "side"):
        assert listing(build_model_from_cfg(EasyDict(dict(TINY_FINETUNE, transfer_type=ttype)))) == ref("ft_" + ttype), ttype
    # and the checkpoint container written by the builder is the reference's (tools/builder.py:138-144)
    import argparse
    from act_amd.tools import builder
    args = argparse.Namespace(local_rank=0, experiment_path=str(tmp_path))
    opt = torch.optim.AdamW([p for p in s2.parameters() if p.requires_grad], lr=1e-3)

    class M:
        def state_dict(self):
            return {"acc": 0.0}
    builder.save_checkpoint(s2, opt, 3, M(), M(), "ckpt-last", args)
    ck = torch.load(str(tmp_path / "ckpt-last.pth"), map_location="cpu")
    assert set(ck) == {"base_model", "optimizer", "epoch", "metrics", "best_metrics"} and ck["epoch"] == 3
    assert listing_keys(ck["base_model"]) == set(ref("stage2"))


def listing_keys(sd):
    return {k[len("module."):] if k.startswith("module.") else k for k in sd}


def test_fewshot_and_scanobjectnn_loader_contracts(tmp_path):
    """ModelNetFewShot from a pickle in the reference's layout (datasets/ModelNetDatasetFewShot.py:28-71) and ScanObjectNN through an injected
    h5 reader (datasets/ScanObjectNNDataset.py:11-88; h5py is not in this image): item tuples, normalisation, train-only point shuffling."""
    import pickle
    from act_amd.datasets import build_dataset_from_cfg, DATASETS
    from act_amd.datasets import FinetuneDatasets as FD
    from act_amd.utils.config import EasyDict
    rs = np.random.RandomState(0)
    samples = {s: [(rs.standard_normal((64, 6)).astype(np.float32) * 3 + 1, lab, None) for lab in (3, 7, 7, 1)] for s in ("train", "test")}
    d = tmp_path / "5way_10shot"; d.mkdir()
    with open(d / "2.pkl", "wb") as f:
        pickle.dump(samples, f)
    base = EasyDict(NAME="ModelNetFewShot", DATA_PATH=str(tmp_path), N_POINTS=8192, NUM_CATEGORY=40, USE_NORMALS=False)
    with pytest.raises(RuntimeError):
        build_dataset_from_cfg(base, EasyDict(subset="train"))                              # --way / --shot / --fold missing
    ds = build_dataset_from_cfg(base, EasyDict(subset="test", way=5, shot=10, fold=2))
    tax, mid, (p, label) = ds[1]
    assert len(ds) == 4 and (tax, mid, label) == ("ModelNet", "sample", 7) and p.shape == (64, 3) and p.dtype == torch.float32
    ref = samples["test"][1][0][:, :3].astype(np.float64); ref = ref - ref.mean(0); ref = ref / np.sqrt((ref ** 2).sum(1)).max()
    assert np.abs(p.numpy() - ref).max() < 1e-5                                             # test split: stored order, pc_normalize
    assert torch.equal(ds[1][2][0], p)                                                      # the stored array is not modified by a read
    tr = build_dataset_from_cfg(base, EasyDict(subset="train", way=5, shot=10, fold=2))
    np.random.seed(1)
    q = tr[0][2][0]
    assert not torch.equal(q, tr[0][2][0]) and torch.allclose(q.sum(0), tr[0][2][0].sum(0), atol=1e-4)   # a permutation of the same points
    # ScanObjectNN: file names per class / split, raw (un-normalised) points, label passthrough
    seen = []

    def reader(path):
        seen.append(path)
        return rs.standard_normal((5, 2048, 3)).astype(np.float32), np.arange(5)
    for name, files in (("ScanObjectNN", ("training_objectdataset.h5", "test_objectdataset.h5")),
                        ("ScanObjectNN_hardest", ("training_objectdataset_augmentedrot_scale75.h5", "test_objectdataset_augmentedrot_scale75.h5"))):
        assert name in DATASETS
        cls = getattr(FD, name)
        for subset, fn in zip(("train", "test"), files):
            sd = cls(EasyDict(subset=subset, ROOT="/data/scan"), reader=reader)
            assert seen[-1] == "/data/scan/" + fn and len(sd) == 5
            tax, mid, (p, label) = sd[4]
            assert (tax, mid, label) == ("ScanObjectNN", "sample", 4) and p.shape == (2048, 3)
            assert (subset == "train") != torch.equal(p, torch.from_numpy(sd.points[4]))
    with pytest.raises(NotImplementedError):
        FD.ScanObjectNN(EasyDict(subset="val", ROOT="/x"), reader=reader)
    with pytest.raises(RuntimeError, match="h5py"):
        build_dataset_from_cfg(EasyDict(NAME="ScanObjectNN", ROOT=str(tmp_path)), EasyDict(subset="train"))


def test_every_shipped_yaml_resolves():
    """all recipes of cfgs/ (the reference's set: pretrain, 2 autoencoders, 18 finetune recipes + the synthetic variants) load through
    cfg_from_yaml_file with their dataset _base_ files, name a registered model / dataset and carry the keys the runners read."""
    import glob, os
    from act_amd.models import MODELS
    from act_amd.datasets import DATASETS
    from act_amd.utils.config import cfg_from_yaml_file
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "act_amd")
    files = [f for f in sorted(glob.glob(os.path.join(root, "cfgs", "**", "*.yaml"), recursive=True)) if "dataset_configs" not in f]
    assert len(files) >= 25
    for f in files:
        cfg = cfg_from_yaml_file(os.path.relpath(f, root))
        assert cfg.model.NAME in MODELS, f
        for split in cfg.dataset.values():
            assert split._base_.NAME in DATASETS, (f, split._base_.NAME)
        assert cfg.optimizer.type == "AdamW" and cfg.scheduler.type == "CosLR" and cfg.total_bs > 0 and cfg.max_epoch > 0, f
        if cfg.model.NAME == "PointTransformer":
            assert cfg.npoints in (1024, 2048, 8192) and cfg.model.cls_dim in (15, 40) and cfg.model.transfer_type in ("full", "linear", "mlp-3", "linaer"), f
            # ('linaer': the reference's finetune_scan_objonly_linear.yaml spells it so; models/act.py:771-809 then builds the mlp-3 head and
            #  freezes nothing, and so does this build)


def test_fewshot_recipe_builds_its_datasets_from_cli_args(tmp_path):
    """The shipped few_shot recipe through the runner's own plumbing: --way / --shot / --fold reach the dataset sections (main.py:72-78) so that
    builder.dataset_builder constructs ModelNetFewShot; without them the construction fails as in the reference."""
    import argparse
    import pickle
    from act_amd.tools import builder
    from act_amd.utils.config import cfg_from_yaml_file, apply_fewshot_args
    rs = np.random.RandomState(0)
    samples = {s: [(rs.standard_normal((64, 6)).astype(np.float32), lab, None) for lab in (0, 1, 2, 3, 4, 0, 1, 2)] for s in ("train", "test")}
    d = tmp_path / "5way_10shot"; d.mkdir()
    with open(d / "3.pkl", "wb") as f:
        pickle.dump(samples, f)
    config = cfg_from_yaml_file("cfgs/finetune_classification/few_shot/fewshot_modelnet.yaml")
    for sec in (config.dataset.train, config.dataset.val):
        sec._base_.DATA_PATH = str(tmp_path)
        sec.others.bs = 4
    args = argparse.Namespace(distributed=False, num_workers=0, way=5, shot=10, fold=3)
    with pytest.raises(RuntimeError):
        builder.dataset_builder(args, config.dataset.train)                                  # the CLI values have not been applied yet
    apply_fewshot_args(args, config)
    apply_fewshot_args(args, config)                                                         # idempotent
    for sec, n_batches in ((config.dataset.train, 2), (config.dataset.val, 2)):
        assert (sec.others.way, sec.others.shot, sec.others.fold) == (5, 10, 3)
        _, loader = builder.dataset_builder(args, sec)
        assert len(loader) == n_batches
        tax, mid, (pts, label) = next(iter(loader))
        assert pts.shape == (4, 64, 3) and label.shape == (4,)
    cfg2 = cfg_from_yaml_file("cfgs/finetune_classification/full/finetune_modelnet.yaml")
    apply_fewshot_args(argparse.Namespace(shot=-1, way=-1, fold=-1), cfg2)
    assert "shot" not in cfg2.dataset.train.others


def test_announced_batch_mark_is_per_model():
    """runner_pretrain.train_step must not augment twice the batch it already announced to the teacher prefetch -- and that memory belongs to the
    model it was announced to: two models trained in one process do not see each other's batches; an in-place change of the tensor voids the mark."""
    import torch
    from act_amd.tools.runner_pretrain import _Announced
    m1, m2 = torch.nn.Linear(2, 2), torch.nn.Linear(2, 2)
    a, b = torch.zeros(3), torch.zeros(3)
    assert not _Announced.is_marked(m1, a)
    _Announced.mark(m1, a)
    assert _Announced.is_marked(m1, a) and not _Announced.is_marked(m2, a) and not _Announced.is_marked(m1, b)
    _Announced.mark(m2, b)
    assert _Announced.is_marked(m1, a) and _Announced.is_marked(m2, b)
    a.add_(1.0)                                              # a new version of the tensor is a new batch
    assert not _Announced.is_marked(m1, a)
    assert "_act_announced" not in m1.state_dict()


def test_bench_refuses_to_spawn_more_ranks_than_devices():
    """``python bench.py --gpus 2`` with no launcher environment spawns its own ranks -- but only when the devices exist: on a box with fewer
    GPUs it must say so at once instead of starting ranks that die one by one (here: 0 GPUs)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "ACT_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=300)
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two devices visible: the spawn would go ahead")
    assert r.returncode != 0 and "--gpus 2 but" in r.stderr and "GPU(s) visible" in r.stderr


def test_fused_adamw_fast_path_equals_torch_adamw():
    """builder.FusedAdamW: the cached-list step issues the same two native calls as torch.optim.AdamW(fused=True) -- parameters, moments and step
    counters stay torch.equal over steps with a changing lr, a parameter that loses its gradient for one step (stock fallback), step hooks and a
    state_dict round trip (tools/builder.py:38-55 builds these two groups)."""
    import copy
    from act_amd.tools.builder import FusedAdamW
    torch.manual_seed(0)
    net_a = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.LayerNorm(8), torch.nn.Linear(8, 3))
    net_b = copy.deepcopy(net_a)

    def groups(net):
        nd = [p for p in net.parameters() if p.dim() == 1]; d = [p for p in net.parameters() if p.dim() > 1]
        return [{'params': nd, 'weight_decay': 0.}, {'params': d, 'weight_decay': 0.05}]
    oa = FusedAdamW(groups(net_a), lr=1e-3, weight_decay=0.05, fused=True)
    ob = torch.optim.AdamW(groups(net_b), lr=1e-3, weight_decay=0.05, fused=True)
    fired = []
    oa.register_step_pre_hook(lambda *a: fired.append(1))
    for i in range(6):
        x = torch.randn(5, 6)
        for net, o in ((net_a, oa), (net_b, ob)):
            for g in o.param_groups:
                g['lr'] = 1e-3 / (i + 1)
            net(x).pow(2).sum().backward()
            if i == 3:
                net[2].bias.grad = None                  # a parameter without gradient: the stock step handles it
            o.step(); o.zero_grad()
        if i == 1:
            assert set(oa._fast) == {0, 1}               # the cached lists are in use from the second step on
        if i == 4:
            oa.load_state_dict(copy.deepcopy(oa.state_dict()))
    assert len(fired) == 6
    assert all(p.grad is None for p in net_a.parameters())
    for pa, pb in zip(net_a.parameters(), net_b.parameters()):
        assert torch.equal(pa, pb)
    sa, sb = oa.state_dict()['state'], ob.state_dict()['state']
    for k in sb:
        assert all(torch.equal(sa[k][n], sb[k][n]) for n in ('step', 'exp_avg', 'exp_avg_sq'))


def test_fused_adamw_cache_follows_direct_state_replacement():
    """moments replaced WITHOUT load_state_dict (``optimizer.state[p]['exp_avg'] = t`` on the first / last parameter of a group, or
    ``optimizer.state.clear()``) must not leave the fast path updating orphaned tensors: trajectories stay torch.equal to torch.optim.AdamW under
    the same surgery, and state_dict() reports the tensors that are being updated."""
    import copy
    from act_amd.tools.builder import FusedAdamW
    torch.manual_seed(0)
    net_a = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Linear(8, 3))
    net_b = copy.deepcopy(net_a)
    oa = FusedAdamW(net_a.parameters(), lr=1e-2, weight_decay=0.05, fused=True)
    ob = torch.optim.AdamW(net_b.parameters(), lr=1e-2, weight_decay=0.05, fused=True)
    for i in range(8):
        x = torch.randn(5, 6)
        for net, o in ((net_a, oa), (net_b, ob)):
            net(x).pow(2).sum().backward()
            o.step(); o.zero_grad()
            ps = list(net.parameters())
            if i == 2:
                o.state[ps[0]]['exp_avg'] = torch.full_like(ps[0], 0.5)          # first parameter's first moment replaced
            if i == 4:
                o.state[ps[-1]]['exp_avg_sq'] = torch.full_like(ps[-1], 0.25)    # last parameter's second moment replaced
            if i == 6:
                o.state.clear()                                                  # optimizer state dropped: lazily re-created by the stock step
        if i == 1:
            assert 0 in oa._fast
    for pa, pb in zip(net_a.parameters(), net_b.parameters()):
        assert torch.equal(pa, pb)
    sa, sb = oa.state_dict()['state'], ob.state_dict()['state']
    for k in sb:
        assert all(torch.equal(sa[k][n], sb[k][n]) for n in ('step', 'exp_avg', 'exp_avg_sq'))
    c = oa._lists(0, oa.param_groups[0])
    assert c is not None and all(oa.state[p]['exp_avg'] is m for p, m in zip(c[1], c[2]))


def test_announced_batch_marker_does_not_break_pickling():
    """runner_pretrain._Announced keeps its weak reference OUTSIDE the module: a model that went through train_step bookkeeping pickles"""
    import io
    from act_amd.tools.runner_pretrain import _Announced
    m = torch.nn.Linear(2, 2); t = torch.zeros(3)
    _Announced.mark(m, t)
    assert _Announced.is_marked(m, t)
    t.add_(1)
    assert not _Announced.is_marked(m, t)
    torch.save(m, io.BytesIO())


def test_split_bf16_teacher_is_off_unless_asked_for():
    """the opt-in split-bf16 path of the frozen teacher (ACT_TEACHER_BF16X3) must never be on by default: every headline number is f32-input MFMA"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import act_amd.composite as CP; print(int(CP.TEACHER_BF16X3))"
    env = {k: v for k, v in os.environ.items() if k != "ACT_TEACHER_BF16X3"}
    assert subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300).stdout.strip() == "0"
    assert subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(env, ACT_TEACHER_BF16X3="1"), capture_output=True, text=True,
                          timeout=300).stdout.strip() == "1"


def test_rank_affinity_plan_is_disjoint_and_numa_aware():
    """dist_utils.plan_affinity / pin_rank (eight enqueue loops share one host): equal disjoint slices of the allowed CPUs, cut from
    the GPU's NUMA node when known, never empty; pin_rank really narrows the affinity of a process (checked in a child so this process keeps its own)."""
    import subprocess
    import sys
    from act_amd.utils.dist_utils import plan_affinity, _parse_cpulist
    assert _parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    allowed = list(range(256))
    slices = [plan_affinity(allowed, r, 8) for r in range(8)]
    assert all(len(s) == 32 for s in slices) and sorted(c for s in slices for c in s) == allowed
    # two sockets: GPUs 0-3 on node 0 (cpus 0-63 + 128-191), GPUs 4-7 on node 1
    node0 = list(range(0, 64)) + list(range(128, 192)); node1 = [c for c in allowed if c not in set(node0)]
    sl = [plan_affinity(allowed, r, 8, node0 if r < 4 else node1, (r % 4, 4)) for r in range(8)]
    assert all(len(s) == 32 for s in sl) and sorted(c for s in sl for c in s) == allowed
    assert all(set(sl[r]) <= set(node0) for r in range(4)) and all(set(sl[r]) <= set(node1) for r in range(4, 8))
    assert plan_affinity([3, 5], 5, 8) == [5] and plan_affinity(allowed, 0, 8, max_cores=4) == [0, 1, 2, 3]
    assert plan_affinity(allowed, 1, 2, node_cpus=[999], ranks_on_node=(0, 1)) == list(range(128, 256))       # a node outside the cpuset: plain slices
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.path.insert(0, %r); from act_amd.utils.dist_utils import pin_rank; a = sorted(os.sched_getaffinity(0)); "
            "i = pin_rank(1, 2); b = sorted(os.sched_getaffinity(0)); "
            "print(len(a), len(b), i['pinned'], b == a[len(a) // 2: 2 * (len(a) // 2)] if len(a) > 1 else True); "
            "os.environ['ACT_PIN_CORES'] = '0'; print(pin_rank(0, 2)['pinned'], pin_rank(0, 1)['pinned'])" % root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]
    l1, l2 = r.stdout.strip().splitlines()[-2:]
    na, nb, pinned, ok = l1.split()
    assert ok == "True" and (pinned == "True") == (int(na) > 1) and l2 == "False False"
