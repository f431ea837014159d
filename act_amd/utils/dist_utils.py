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


# ---- host-side placement of the ranks of one node ---------------------------------------------------------------------------------------
# Every rank runs its own enqueue loop (~5 ms of Python + ~500 launches per Stage-II step against a ~26 ms GPU step): eight of them on one host must not
# migrate between sockets or share cores.  Reference: main.py:44-67 starts one process per GPU and leaves placement to the OS.
def _parse_cpulist(text):
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_cpus(device_index):
    """CPUs of the NUMA node the GPU's PCIe function hangs off (sysfs), or None when the platform does not say (VM, container without sysfs, node -1)"""
    try:
        p = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return None
        return _parse_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read())
    except Exception:
        return None


def plan_affinity(allowed, local_rank, local_world, node_cpus=None, ranks_on_node=None, max_cores=None):
    """the CPU set of one rank: an equal, disjoint, contiguous slice of ``allowed`` (the CPUs this process may use); with ``node_cpus`` (the GPU's NUMA
    node) the slice is cut from allowed ∩ node_cpus among the ``ranks_on_node`` = (index of this rank among the ranks of that node, their count).
    Never returns an empty set: with fewer CPUs than ranks the ranks share round-robin."""
    allowed = sorted(allowed)
    pool, idx, cnt = allowed, local_rank, max(1, local_world)
    if node_cpus:
        inter = [c for c in allowed if c in set(node_cpus)]
        if inter and ranks_on_node:
            pool, (idx, cnt) = inter, ranks_on_node
    per = len(pool) // cnt
    if per == 0:
        return [pool[idx % len(pool)]]
    cores = pool[idx * per:(idx + 1) * per]
    return cores[:max_cores] if max_cores else cores


def pin_rank(local_rank, local_world, device_index=None, max_cores=None):
    """os.sched_setaffinity of THIS process to its slice (ACT_PIN_CORES=0 disables); returns a description for logs / the bench line.  With one rank per
    node nothing is pinned (the whole host is that rank's)."""
    info = {"pinned": False, "local_rank": local_rank, "local_world": local_world}
    if os.environ.get("ACT_PIN_CORES", "1") == "0" or local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        info["reason"] = "ACT_PIN_CORES=0" if os.environ.get("ACT_PIN_CORES", "1") == "0" else ("single rank" if local_world <= 1 else "no sched_setaffinity")
        return info
    allowed = sorted(os.sched_getaffinity(0))
    node, ron = None, None
    if device_index is not None and torch.cuda.is_available():
        # the NUMA pools of ALL local ranks (every rank sees every device of the node and does the same sysfs walk, so all ranks decide alike): the NUMA-aware
        # plan is used only when it works for every rank -- a cpuset that misses one GPU's node would otherwise mix plain and NUMA slices, which overlap
        ndev = max(1, torch.cuda.device_count())
        nodes = [gpu_numa_cpus(r % ndev) for r in range(local_world)]
        aset = set(allowed)
        if all(n and any(c in aset for c in n) for n in nodes):
            node = nodes[local_rank]
            same = [r for r in range(local_world) if nodes[r] == node]
            ron = (same.index(local_rank), len(same))
    cores = plan_affinity(allowed, local_rank, local_world, node, ron, max_cores)
    try:
        os.sched_setaffinity(0, cores)
    except OSError as e:
        info["reason"] = f"sched_setaffinity failed: {e}"
        return info
    # the intra-op pool of torch (CPU-side tensor bookkeeping only: every kernel runs on the GPU) must not oversubscribe the slice
    torch.set_num_threads(max(1, min(len(cores), 8)))
    info.update(pinned=True, cores=f"{cores[0]}-{cores[-1]}" if cores == list(range(cores[0], cores[-1] + 1)) else ",".join(map(str, cores)),
                n_cores=len(cores), numa_aware=bool(node and ron))
    return info
