"""process-group helpers (reference: utils/dist_utils.py:9-54).  backend 'nccl' IS RCCL on ROCm;
one process per GPU over xGMI.  'gloo' is accepted for the CPU multi-process tests."""
import os

import torch
from torch import distributed as dist


def init_dist(launcher, backend="nccl", **kwargs):
    if launcher != "pytorch":
        raise ValueError(f"Invalid launcher type: {launcher}")
    rank = int(os.environ["RANK"])
    if backend == "nccl":
        if not torch.cuda.is_available():
            raise RuntimeError("backend 'nccl' (RCCL) needs GPUs")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank % torch.cuda.device_count())))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group(backend=backend, **kwargs)
    print(f"init distributed in rank {dist.get_rank()}")


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def reduce_tensor(tensor, args):
    """mean over ranks (used for the logged loss, tools/runner_pretrain.py:159-167)."""
    rt = tensor.clone()
    dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    rt /= args.world_size
    return rt


def gather_tensor(tensor, args):
    output_tensors = [tensor.clone() for _ in range(args.world_size)]
    dist.all_gather(output_tensors, tensor)
    return torch.cat(output_tensors, dim=0)
