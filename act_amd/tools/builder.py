"""dataset / model / optimizer / scheduler / checkpoint builders (reference: tools/builder.py:14-173)."""
import math
import os

import torch
import torch.optim as optim

from ..datasets import build_dataset_from_cfg
from ..models import build_model_from_cfg
from ..utils.logger import print_log
from ..utils.misc import worker_init_fn


def dataset_builder(args, config):
    """(sampler, DataLoader) of one dataset section of the YAML (tools/builder.py:14-33): the train subset is shuffled and drops its
    ragged last batch; under DDP every rank gets a DistributedSampler shard.  Host batches are pinned so the H2D copy of the next
    batch overlaps the current step."""
    dataset = build_dataset_from_cfg(config._base_, config.others)
    is_train = config.others.subset == 'train'
    sampler = torch.utils.data.distributed.DistributedSampler(dataset, shuffle=is_train) if args.distributed else None
    loader = torch.utils.data.DataLoader(
        dataset, batch_size=config.others.bs, sampler=sampler, shuffle=(is_train and sampler is None), drop_last=is_train,
        num_workers=int(args.num_workers), worker_init_fn=worker_init_fn, pin_memory=True)
    return sampler, loader


def model_builder(config):
    return build_model_from_cfg(config)


def add_weight_decay(model, weight_decay=1e-5, skip_list=()):
    """two AdamW groups (tools/builder.py:38-51): no decay for 1-D params, '.bias' and names containing 'token'."""
    decay, no_decay = [], []
    module = model.module if hasattr(model, "module") else model
    for name, param in module.named_parameters():
        if not param.requires_grad:
            continue
        if len(param.shape) == 1 or name.endswith(".bias") or 'token' in name or name in skip_list:
            no_decay.append(param)
        else:
            decay.append(param)
    return [{'params': no_decay, 'weight_decay': 0.}, {'params': decay, 'weight_decay': weight_decay}]


class CosineLRScheduler:
    """timm 0.5.4 CosineLRScheduler as configured in tools/builder.py:71-81 (t_in_epochs, cycle_limit=1):
    linear warm-up from warmup_lr_init over warmup_t epochs, then lr_min + 0.5 (lr - lr_min)(1 + cos(pi t / t_initial))."""

    def __init__(self, optimizer, t_initial, lr_min=0., warmup_t=0, warmup_lr_init=0., cycle_limit=1, t_in_epochs=True, **_):
        self.optimizer = optimizer
        self.t_initial, self.lr_min, self.warmup_t, self.warmup_lr_init = t_initial, lr_min, warmup_t, warmup_lr_init
        self.cycle_limit = cycle_limit
        for g in optimizer.param_groups:
            g.setdefault('initial_lr', g['lr'])
        self.base_values = [g['initial_lr'] for g in optimizer.param_groups]
        self.warmup_steps = [(v - warmup_lr_init) / warmup_t for v in self.base_values] if warmup_t else [1 for _ in self.base_values]
        if warmup_t:
            self._set(self._get_lr(0) if False else [warmup_lr_init for _ in self.base_values])

    def _get_lr(self, t):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * s for s in self.warmup_steps]
        i = t // self.t_initial
        t_curr = t - self.t_initial * i
        if i < self.cycle_limit:
            return [self.lr_min + 0.5 * (v - self.lr_min) * (1 + math.cos(math.pi * t_curr / self.t_initial)) for v in self.base_values]
        return [self.lr_min for _ in self.base_values]

    def _set(self, values):
        for g, v in zip(self.optimizer.param_groups, values):
            g['lr'] = v

    def step(self, epoch, metric=None):
        self._set(self._get_lr(epoch))

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != 'optimizer'}

    def load_state_dict(self, sd):
        self.__dict__.update(sd)


class FusedAdamW(optim.AdamW):
    """``torch.optim.AdamW(fused=True)`` -- the same two native calls per parameter group (``torch._foreach_add_`` on the step counters,
    ``torch._fused_adamw_``), the same state layout, ``state_dict`` and step hooks -- with the per-step Python bookkeeping of ``Adam._init_group`` /
    ``_fused_adam`` (state lookups per parameter, re-grouping by device and dtype: ~0.8 ms of host time per Stage-II step for 200 parameters)
    done once and cached.  Any configuration the cache does not describe (first step: lazy state creation; a parameter without gradient;
    amsgrad / maximize / capturable / tensor lr / grad scaler; a closure) takes ``torch.optim.AdamW.step`` unchanged."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self._fast = {}

    def add_param_group(self, group):
        self.__dict__.setdefault("_fast", {}).clear()
        return super().add_param_group(group)

    def load_state_dict(self, sd):
        self._fast.clear()
        return super().load_state_dict(sd)

    def _lists(self, gi, group):
        c = self._fast.get(gi)
        params = group['params']
        if c is not None and c[0] == len(params):
            # re-validate by identity: ``optimizer.state[p][...] = t`` or ``optimizer.state.clear()`` replaces moments without going
            # through load_state_dict / add_param_group; the first and the last parameter's entries are the cheap witness, a mismatch rebuilds the lists
            s0, s1 = self.state.get(params[0]), self.state.get(params[-1])
            if (s0 and s1 and params[0] is c[1][0] and params[-1] is c[1][-1] and s0.get('exp_avg') is c[2][0] and s1.get('exp_avg') is c[2][-1]
                    and s0.get('exp_avg_sq') is c[3][0] and s1.get('exp_avg_sq') is c[3][-1] and s0.get('step') is c[4][0] and s1.get('step') is c[4][-1]):
                return c
            self._fast.pop(gi, None)
        if not (group.get('fused') and not group['amsgrad'] and not group['maximize'] and not group['capturable'] and not group['differentiable']
                and group.get('decoupled_weight_decay', True)):
            return None
        st = [self.state.get(p) for p in params]
        if not params or any(not s for s in st):
            return None                                              # lazy state creation happens in the stock step
        dev, dt = params[0].device, params[0].dtype
        if dt != torch.float32 or any(p.device != dev or p.dtype != dt for p in params):
            return None
        c = self._fast[gi] = (len(params), list(params), [s['exp_avg'] for s in st], [s['exp_avg_sq'] for s in st], [s['step'] for s in st])
        return c

    def _stock_step(self, closure=None):
        """torch.optim.AdamW.step WITHOUT its hook wrapper (torch patches ``cls.step`` of every optimizer class it has instantiated with a wrapper
        that runs the step pre / post hooks; the wrapper around THIS class's step has already run them)"""
        f = optim.AdamW.step
        if getattr(f, "hooked", False):
            f = getattr(f, "__wrapped__", f)
        return f(self, closure)

    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None or getattr(self, "grad_scale", None) is not None or getattr(self, "found_inf", None) is not None:
            return self._stock_step(closure)
        plan = []
        for gi, group in enumerate(self.param_groups):
            c = self._lists(gi, group)
            if c is None or not isinstance(group['lr'], float):
                return self._stock_step()
            grads = [p.grad for p in c[1]]
            if any(g is None for g in grads):
                return self._stock_step()
            plan.append((group, c, grads))
        for group, (_, params, exp_avgs, exp_avg_sqs, steps), grads in plan:
            beta1, beta2 = group['betas']
            torch._foreach_add_(steps, 1)
            torch._fused_adamw_(params, grads, exp_avgs, exp_avg_sqs, [], steps, amsgrad=False, lr=group['lr'], beta1=beta1, beta2=beta2,
                                weight_decay=group['weight_decay'], eps=group['eps'], maximize=False, grad_scale=None, found_inf=None)
        return None

    def zero_grad(self, set_to_none=True):
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group['params']:
                p.grad = None


def build_opti_sche(base_model, config):
    opti_config = config.optimizer
    if opti_config.type == 'AdamW':
        param_groups = add_weight_decay(base_model, weight_decay=opti_config.kwargs.weight_decay)
        fused = any(p.is_cuda for g in param_groups for p in g['params'])
        optimizer = (FusedAdamW if fused else optim.AdamW)(param_groups, fused=fused, **opti_config.kwargs)
    elif opti_config.type == 'Adam':
        optimizer = optim.Adam(base_model.parameters(), **opti_config.kwargs)
    elif opti_config.type == 'SGD':
        optimizer = optim.SGD(base_model.parameters(), nesterov=True, momentum=0.9, **opti_config.kwargs)
    else:
        raise NotImplementedError()
    sche_config = config.scheduler
    if sche_config.type == 'CosLR':
        scheduler = CosineLRScheduler(optimizer, t_initial=sche_config.kwargs.epochs, lr_min=1e-7, warmup_lr_init=1e-6,
                                      warmup_t=sche_config.kwargs.initial_epochs, cycle_limit=1, t_in_epochs=True)
    elif sche_config.type == 'StepLR':
        scheduler = torch.optim.lr_scheduler.StepLR(optimizer, **sche_config.kwargs)
    elif sche_config.type == 'function':
        scheduler = None
    else:
        raise NotImplementedError()
    return optimizer, scheduler


# ---- checkpoints: the reference's container {'base_model', 'optimizer', 'epoch', 'metrics', 'best_metrics'} (tools/builder.py:138-144),
# keys optionally prefixed with 'module.' when the file was written from a DDP / DataParallel wrapper ---------------------------------
_LAST = 'ckpt-last.pth'


def _unwrap(model):
    return model.module if hasattr(model, "module") else model


def _weights(blob, *slots):
    """state_dict stored under the first present slot, DDP prefixes removed"""
    for slot in slots:
        sd = blob.get(slot)
        if sd is not None:
            return {k.replace("module.", ""): v for k, v in sd.items()}
    raise RuntimeError('mismatch of ckpt weight')


def _as_dict(metric):
    return metric if isinstance(metric, dict) else metric.state_dict()


def _read_last(args, what, logger):
    path = os.path.join(args.experiment_path, _LAST)
    if not os.path.exists(path):
        print_log(f'[RESUME INFO] no checkpoint file from path {path}...', logger=logger)
        return None
    print_log(f'[RESUME INFO] Loading {what} from {path}...', logger=logger)
    return torch.load(path, map_location='cpu')


def resume_model(base_model, args, logger=None):
    """-> (first epoch to run, best metrics dict); (0, 0) when there is nothing to resume (tools/builder.py:97-121)"""
    blob = _read_last(args, 'model weights', logger)
    if blob is None:
        return 0, 0
    _unwrap(base_model).load_state_dict(_weights(blob, 'base_model'), strict=True)
    best = _as_dict(blob['best_metrics'])
    print_log(f"[RESUME INFO] resume ckpts @ {blob['epoch']} epoch( best_metrics = {best})", logger=logger)
    return blob['epoch'] + 1, best


def resume_optimizer(optimizer, args, logger=None):
    """restore the optimizer state of ckpt-last.pth (tools/builder.py:123-132).  A checkpoint written by the reference lists the
    never-trained lm_head / cls_head parameters in its AdamW groups (they are frozen here, see runner_pretrain.freeze_unused_heads):
    such a state cannot be mapped onto this optimizer, which then starts from fresh moments -- said loudly, not silently."""
    blob = _read_last(args, 'optimizer', logger)
    if blob is None:
        return 0, 0, 0
    saved = blob['optimizer']
    have = [len(g['params']) for g in optimizer.param_groups]
    want = [len(g['params']) for g in saved['param_groups']]
    if have != want:
        print_log(f'[RESUME WARNING] optimizer state of the checkpoint has parameter groups of sizes {want}, this run has {have} '
                  '(frozen unused heads): AdamW moments are NOT restored', logger=logger)
        return None
    optimizer.load_state_dict(saved)


def save_checkpoint(base_model, optimizer, epoch, metrics, best_metrics, prefix, args, skip=False, logger=None):
    path = os.path.join(args.experiment_path, prefix + '.pth')
    if skip:
        print_log(f"Skipped saving checkpoint at {path}", logger=logger)
    elif args.local_rank == 0:
        blob = dict(base_model=_unwrap(base_model).state_dict(), optimizer=optimizer.state_dict(), epoch=epoch,
                    metrics=_as_dict(metrics) if metrics is not None else {},
                    best_metrics=_as_dict(best_metrics) if best_metrics is not None else {})
        torch.save(blob, path)
        print_log(f"Save checkpoint at {path}", logger=logger)


def load_model(base_model, ckpt_path, logger=None):
    """strict load of a released / saved checkpoint ('model' or 'base_model' slot; tools/builder.py:147-173)"""
    if not os.path.exists(ckpt_path):
        raise NotImplementedError('no checkpoint file from path %s...' % ckpt_path)
    print_log(f'Loading weights from {ckpt_path}...', logger=logger)
    blob = torch.load(ckpt_path, map_location='cpu')
    _unwrap(base_model).load_state_dict(_weights(blob, 'model', 'base_model'), strict=True)
    perf = _as_dict(blob['metrics']) if blob.get('metrics') is not None else 'No Metrics'
    print_log(f"ckpts @ {blob.get('epoch', -1)} epoch( performance = {perf})", logger=logger)
