"""Stage-I autoencoder training loop (reference: tools/runner_autoencoder.py:18-217), same ``run_net`` signature.

temperature: cosine 1 -> 0.0625 over ``temp.ntime`` iterations (:42-53); KL weight: 0 for the first 10k iterations, then
cosine ``kldweight.start`` -> ``kldweight.target`` (:18-40); loss = recon (CD-L1 coarse + fine) + kld_weight * KL.
Validation reports whole-cloud Chamfer L1/L2 x1000 (the reference additionally prints per-taxonomy tables and F-score,
which need open3d and the ShapeNet taxonomy files: out of scope)."""
import math
import time

import torch

from . import builder
from .runner_pretrain import wrap_ddp, _Single
from ..extensions.chamfer_dist import ChamferDistanceL1, ChamferDistanceL2
from ..utils import dist_utils
from ..utils.AverageMeter import AverageMeter
from ..utils.logger import get_logger, print_log


class Metrics:
    """minimal stand-in for utils/metrics.py Metrics: lower CDL1 is better."""

    def __init__(self, name="CDL1", values=None):
        self.name = name
        # a checkpoint written before the first validation stores an empty dict: that means "no best yet"
        self._values = dict(values) if isinstance(values, dict) and name in values else {"CDL1": float("inf"), "CDL2": float("inf")}

    def better_than(self, other):
        return other is None or self._values[self.name] < other._values[other.name]

    def state_dict(self):
        return dict(self._values)


def kld_weight(config, niter):
    start, target, ntime = config.kldweight.start, config.kldweight.target, config.kldweight.ntime
    _niter = niter - 10000
    if _niter > ntime:
        return target
    if _niter < 0:
        return 0.
    return target + (start - target) * (1. + math.cos(math.pi * float(_niter) / ntime)) / 2.


def compute_loss(loss_1, loss_2, config, niter, train_writer):
    w = kld_weight(config, niter)
    if train_writer is not None:
        train_writer.add_scalar('Loss/Batch/KLD_Weight', w, niter)
    return loss_1 + w * loss_2


def get_temp(config, niter):
    if config.get('temp') is None:
        return 0
    start, target, ntime = config.temp.start, config.temp.target, config.temp.ntime
    if niter > ntime:
        return target
    return target + (start - target) * (1. + math.cos(math.pi * float(niter) / ntime)) / 2.


def train_step(base_model, optimizer, points, config, n_itr, num_iter=1, train_writer=None, draws=None):
    temp = get_temp(config, n_itr)
    module = base_model.module
    ret = base_model(points, temperature=temp, hard=False, draws=draws)
    loss_1, loss_2 = module.get_loss(ret, points)
    loss = compute_loss(loss_1, loss_2, config, n_itr, train_writer)
    loss.backward()
    if num_iter == config.step_per_update:
        optimizer.step()
        optimizer.zero_grad(set_to_none=True)
    return loss_1.detach(), loss_2.detach(), temp


@torch.no_grad()
def validate(base_model, test_dataloader, epoch, cdl1, cdl2, args, config, device, logger=None, max_batches=None):
    base_model.eval()
    tot1 = tot2 = 0.0
    n = 0
    for idx, (_, _, data) in enumerate(test_dataloader):
        points = data.to(device)
        ret = base_model(points, temperature=1., hard=True)
        dense = ret[1]
        tot1 += cdl1(dense, points).item() * 1000 * points.shape[0]
        tot2 += cdl2(dense, points).item() * 1000 * points.shape[0]
        n += points.shape[0]
        if max_batches is not None and idx + 1 >= max_batches:
            break
    m = Metrics(config.consider_metric, {"CDL1": tot1 / max(n, 1), "CDL2": tot2 / max(n, 1)})
    print_log('[Validation] EPOCH: %d  Metrics = %s' % (epoch, m.state_dict()), logger=logger)
    return m


def run_net(args, config, train_writer=None, val_writer=None, max_steps=None, log_every=100):
    logger = get_logger(args.log_name)
    (train_sampler, train_dataloader), (_, test_dataloader) = builder.dataset_builder(args, config.dataset.train), \
        builder.dataset_builder(args, config.dataset.val)
    base_model = builder.model_builder(config.model)
    device = torch.device("cuda", args.local_rank % max(1, torch.cuda.device_count()))
    if args.use_gpu:
        torch.cuda.set_device(device)          # every launch goes to the current device's current stream
        base_model.to(device)
    start_epoch, best_metrics, metrics = 0, None, None
    if args.resume:
        start_epoch, best = builder.resume_model(base_model, args, logger=logger)
        best_metrics = Metrics(config.consider_metric, best if isinstance(best, dict) else None)
    elif args.start_ckpts is not None:
        builder.load_model(base_model, args.start_ckpts, logger=logger)
    base_model = wrap_ddp(base_model, args) if args.distributed else _Single(base_model)
    optimizer, scheduler = builder.build_opti_sche(base_model, config)
    cdl1, cdl2 = ChamferDistanceL1(), ChamferDistanceL2()
    if args.resume:
        builder.resume_optimizer(optimizer, args, logger=logger)

    base_model.zero_grad()
    steps, log = 0, []
    for epoch in range(start_epoch, config.max_epoch + 1):
        if args.distributed:
            train_sampler.set_epoch(epoch)
        base_model.train()
        epoch_start = batch_start = time.time()
        batch_time, data_time, losses = AverageMeter(), AverageMeter(), AverageMeter(['Loss1', 'Loss2'])
        num_iter, n_batches, pending = 0, len(train_dataloader), []
        for idx, (taxonomy_ids, model_ids, data) in enumerate(train_dataloader):
            num_iter += 1
            n_itr = epoch * n_batches + idx
            data_time.update(time.time() - batch_start)
            if config.dataset.train._base_.NAME != 'ShapeNet':
                raise NotImplementedError(f'Train phase do not support {config.dataset.train._base_.NAME}')
            points = data.to(device, non_blocking=True)
            l1, l2, temp = train_step(base_model, optimizer, points, config, n_itr, num_iter, train_writer)
            if num_iter == config.step_per_update:
                num_iter = 0
            if args.distributed:
                l1, l2 = dist_utils.reduce_tensor(l1, args), dist_utils.reduce_tensor(l2, args)
            pending.append(torch.stack((l1, l2)))
            steps += 1
            if idx % log_every == 0 or (max_steps is not None and steps >= max_steps):
                for v1, v2 in (torch.stack(pending) * 1000).tolist():          # one host sync per log interval
                    losses.update([v1, v2]); log.append((v1, v2))
                pending = []
                if train_writer is not None:
                    train_writer.add_scalar('Loss/Batch/Loss_1', losses.val(0), n_itr)
                    train_writer.add_scalar('Loss/Batch/Loss_2', losses.val(1), n_itr)
                    train_writer.add_scalar('Loss/Batch/Temperature', temp, n_itr)
                    train_writer.add_scalar('Loss/Batch/LR', optimizer.param_groups[0]['lr'], n_itr)
                batch_time.update(time.time() - batch_start)
                print_log('[Epoch %d/%d][Batch %d/%d] BatchTime = %.3f (s) DataTime = %.3f (s) Losses = %s lr = %.6f' %
                          (epoch, config.max_epoch, idx + 1, n_batches, batch_time.val(), data_time.val(),
                           ['%.4f' % l for l in losses.val()], optimizer.param_groups[0]['lr']), logger=logger)
            batch_start = time.time()
            if max_steps is not None and steps >= max_steps:
                break
        for v1, v2 in ((torch.stack(pending) * 1000).tolist() if pending else []):
            losses.update([v1, v2]); log.append((v1, v2))
        if config.scheduler.type != 'function' and scheduler is not None:
            scheduler.step(epoch)
        print_log('[Training] EPOCH: %d EpochTime = %.3f (s) Losses = %s lr = %.6f' %
                  (epoch, time.time() - epoch_start, ['%.4f' % l for l in losses.avg()], optimizer.param_groups[0]['lr']), logger=logger)
        if epoch % args.val_freq == 0 and epoch != 0:
            metrics = validate(base_model, test_dataloader, epoch, cdl1, cdl2, args, config, device, logger=logger)
            if metrics.better_than(best_metrics):
                best_metrics = metrics
                builder.save_checkpoint(base_model, optimizer, epoch, metrics, best_metrics, 'ckpt-best', args, logger=logger)
        builder.save_checkpoint(base_model, optimizer, epoch, metrics, best_metrics, 'ckpt-last', args, logger=logger)
        if max_steps is not None and steps >= max_steps:
            break
    return log
