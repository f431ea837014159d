import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(os.path.join(ROOT, "act_amd"))
from act_amd.utils.config import cfg_from_yaml_file
from act_amd.models import build_model_from_cfg
from act_amd.tools import builder
from act_amd.tools.runner_autoencoder import train_step
import bench
dev = torch.device("cuda:0")
config = cfg_from_yaml_file("cfgs/autoencoder/act_dvae_with_pretrained_transformer.yaml")
torch.manual_seed(0)
model = build_model_from_cfg(config.model).to(dev).train()
from act_amd.tools.runner_pretrain import _Single
w = _Single(model)
opt, _ = builder.build_opti_sche(w, config)
pool = [bench.synthetic_clouds(128, 1024, 1234 + i, dev) for i in range(4)]
for i in range(12):
    l1, l2, _ = train_step(w, opt, pool[i % 4], config, 20000 + i)
    gn = sum(float(p.grad.norm() ** 2) for p in model.parameters() if p.grad is not None) ** 0.5 if False else 0
    print(i, float(l1), float(l2), flush=True)
    bad = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    if bad:
        print("non-finite params:", bad[:10]); break
