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
    dataset = build_dataset_from_cfg(config._base_, config.others)
    shuffle = config.others.subset == 'train'
    if args.distributed:
        sampler = torch.utils.data.distributed.DistributedSampler(dataset, shuffle=shuffle)
        dataloader = torch.utils.data.DataLoader(dataset, batch_size=config.others.bs, num_workers=int(args.num_workers),
                                                 drop_last=config.others.subset == 'train', worker_init_fn=worker_init_fn,
                                                 sampler=sampler, pin_memory=True)
    else:
        sampler = None
        dataloader = torch.utils.data.DataLoader(dataset, batch_size=config.others.bs, shuffle=shuffle,
                                                 drop_last=config.others.subset == 'train', num_workers=int(args.num_workers),
                                                 worker_init_fn=worker_init_fn, pin_memory=True)
    return sampler, dataloader


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


def build_opti_sche(base_model, config):
    opti_config = config.optimizer
    if opti_config.type == 'AdamW':
        param_groups = add_weight_decay(base_model, weight_decay=opti_config.kwargs.weight_decay)
        fused = any(p.is_cuda for g in param_groups for p in g['params'])
        optimizer = optim.AdamW(param_groups, fused=fused, **opti_config.kwargs)
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


def _strip(sd):
    return {k.replace("module.", ""): v for k, v in sd.items()}


def resume_model(base_model, args, logger=None):
    ckpt_path = os.path.join(args.experiment_path, 'ckpt-last.pth')
    if not os.path.exists(ckpt_path):
        print_log(f'[RESUME INFO] no checkpoint file from path {ckpt_path}...', logger=logger)
        return 0, 0
    print_log(f'[RESUME INFO] Loading model weights from {ckpt_path}...', logger=logger)
    state_dict = torch.load(ckpt_path, map_location='cpu')
    base_model.load_state_dict(_strip(state_dict['base_model']), strict=True)
    start_epoch = state_dict['epoch'] + 1
    best_metrics = state_dict['best_metrics']
    if not isinstance(best_metrics, dict):
        best_metrics = best_metrics.state_dict()
    print_log(f'[RESUME INFO] resume ckpts @ {start_epoch - 1} epoch( best_metrics = {str(best_metrics):s})', logger=logger)
    return start_epoch, best_metrics


def resume_optimizer(optimizer, args, logger=None):
    ckpt_path = os.path.join(args.experiment_path, 'ckpt-last.pth')
    if not os.path.exists(ckpt_path):
        print_log(f'[RESUME INFO] no checkpoint file from path {ckpt_path}...', logger=logger)
        return 0, 0, 0
    print_log(f'[RESUME INFO] Loading optimizer from {ckpt_path}...', logger=logger)
    optimizer.load_state_dict(torch.load(ckpt_path, map_location='cpu')['optimizer'])


def save_checkpoint(base_model, optimizer, epoch, metrics, best_metrics, prefix, args, skip=False, logger=None):
    path = os.path.join(args.experiment_path, prefix + '.pth')
    if skip:
        print_log(f"Skipped saving checkpoint at {path}", logger=logger)
        return
    if args.local_rank == 0:
        module = base_model.module if hasattr(base_model, "module") else base_model
        torch.save({'base_model': module.state_dict(), 'optimizer': optimizer.state_dict(), 'epoch': epoch,
                    'metrics': metrics.state_dict() if metrics is not None else dict(),
                    'best_metrics': best_metrics.state_dict() if best_metrics is not None else dict()}, path)
        print_log(f"Save checkpoint at {path}", logger=logger)


def load_model(base_model, ckpt_path, logger=None):
    if not os.path.exists(ckpt_path):
        raise NotImplementedError('no checkpoint file from path %s...' % ckpt_path)
    print_log(f'Loading weights from {ckpt_path}...', logger=logger)
    state_dict = torch.load(ckpt_path, map_location='cpu')
    if state_dict.get('model') is not None:
        base_ckpt = _strip(state_dict['model'])
    elif state_dict.get('base_model') is not None:
        base_ckpt = _strip(state_dict['base_model'])
    else:
        raise RuntimeError('mismatch of ckpt weight')
    base_model.load_state_dict(base_ckpt, strict=True)
