"""One tiny Stage-II forward+backward on the GPU, checked against the CPU oracle (used by __graft_entry__.smoke)."""
import torch


def run(dev):
    from tests.golden.fill import fill_module, clouds, TINY_STAGE2, TINY_B, TINY_N
    from oracle import models as OM, layers as OL
    from act_amd.models import build_model_from_cfg
    from act_amd.utils.config import EasyDict
    from act_amd.utils.draws import Draws

    import act_amd.kernels as K
    K.AUTOTUNE = False                      # deterministic launch configurations (cost model / shipped table): same result on every box
    torch.manual_seed(0)
    oracle = fill_module(OM.ACT_PointDistillation(OM.edict(TINY_STAGE2)), "g4.").train()
    model = build_model_from_cfg(EasyDict(TINY_STAGE2))
    model.load_state_dict(oracle.state_dict(), strict=True)
    model.to(dev).train()
    pts = torch.from_numpy(clouds(4, TINY_B, TINY_N))
    rec = OL.Draws(record=True)
    lo = oracle(pts, rec)
    lo.backward()
    lg = model(pts.to(dev), draws=Draws(rec.table, device=dev))
    lg.backward()
    torch.cuda.synchronize()
    assert abs(lg.item() - lo.item()) <= 1e-4, (lg.item(), lo.item())
    go = dict(oracle.named_parameters())["ACT_encoder.blocks.blocks.0.attn.qkv.weight"].grad
    gg = dict(model.named_parameters())["ACT_encoder.blocks.blocks.0.attn.qkv.weight"].grad.cpu()
    assert (gg - go).abs().max() <= 1e-4 * max(1.0, go.abs().max().item())
    print(f"smoke: tiny Stage-II step loss {lg.item():.6f} (oracle {lo.item():.6f}), gradients within 1e-4")
