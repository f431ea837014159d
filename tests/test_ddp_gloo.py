"""CPU, world_size 2, gloo: the data-parallel wrapper of the runner (act_amd.tools.runner_pretrain.wrap_ddp) averages
gradients across ranks exactly like the mean of the per-shard gradients, keeps BatchNorm statistics per rank
(broadcast_buffers=False, the reference's non-SyncBN default) and needs no find_unused_parameters once the dead
heads are frozen.  The model under DDP here is the CPU oracle (the HIP path has no CPU mode)."""
import argparse
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    from tests.golden.fill import fill_module, TINY_STAGE2
    from oracle import models as OM
    torch.manual_seed(0)
    m = fill_module(OM.ACT_PointDistillation(OM.edict(TINY_STAGE2)), "ddp.").train()
    m.dvae_tokenizer.prompt_p = 0.0
    return m


def _draws(rank):
    from oracle.layers import Draws
    from oracle.models import rand_mask
    g = torch.Generator().manual_seed(100 + rank)
    noise = -torch.empty(2, 16, 64).exponential_(generator=g).log()
    return Draws({"mask": rand_mask(2, 16, 12, generator=g), "gumbel": noise})


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests.golden.fill import clouds
    from act_amd.tools.runner_pretrain import wrap_ddp, freeze_unused_heads
    from act_amd.utils import dist_utils
    model = _build()
    freeze_unused_heads(model)
    ddp = wrap_ddp(model, argparse.Namespace(local_rank=rank, use_gpu=False))
    pts = torch.from_numpy(clouds(50 + rank, 2, 128))
    loss = ddp(pts, _draws(rank))
    loss.backward()
    mean_loss = dist_utils.reduce_tensor(loss.detach(), argparse.Namespace(world_size=world))
    grads = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    bn = model.ACT_encoder.encoder.first_conv[1].running_mean.clone()
    torch.save({"grads": grads, "loss": loss.item(), "mean_loss": mean_loss.item(), "bn": bn}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_gradients_equal_mean_of_shard_gradients(tmp_path):
    from tests.golden.fill import clouds
    from act_amd.tools.runner_pretrain import freeze_unused_heads
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(tmp_path / f"r{i}.pt") for i in range(2)]
    # single-process reference: each shard alone, then the mean
    ref_grads, ref_loss, bns = [], [], []
    for rank in range(2):
        m = _build(); freeze_unused_heads(m)
        loss = m(torch.from_numpy(clouds(50 + rank, 2, 128)), _draws(rank))
        loss.backward()
        ref_grads.append({n: p.grad for n, p in m.named_parameters() if p.grad is not None})
        ref_loss.append(loss.item())
        bns.append(m.ACT_encoder.encoder.first_conv[1].running_mean.clone())
    assert set(r[0]["grads"]) == set(ref_grads[0])
    assert not any("lm_head" in n or "cls_head" in n for n in r[0]["grads"])
    for n in ref_grads[0]:
        want = 0.5 * (ref_grads[0][n] + ref_grads[1][n])
        for rank in range(2):
            assert torch.allclose(r[rank]["grads"][n], want, rtol=1e-5, atol=1e-7), n       # identical on both ranks
    for rank in range(2):
        assert abs(r[rank]["loss"] - ref_loss[rank]) < 1e-6
        assert abs(r[rank]["mean_loss"] - 0.5 * sum(ref_loss)) < 1e-6                        # logged loss all-reduce
        assert torch.allclose(r[rank]["bn"], bns[rank])                                      # BN buffers stay per rank
    assert not torch.allclose(r[0]["bn"], r[1]["bn"])
