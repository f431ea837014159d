"""Finetune / inference loop of the PointTransformer classifier (reference: tools/runner_finetune.py:65-470), same
``run_net`` / ``validate`` / ``validate_vote`` / ``test_net`` entry points and batch tuple
``(taxonomy_ids, model_ids, (points, label))``.

Device-first differences, none visible in the config surface:
  * the 8192 -> 1200 farthest-point subsampling, the random 1024-of-1200 choice, the rotation / scale-translate augmentation,
    the cross-entropy loss and the accuracy all run on the device; the training loop reads loss / accuracy back once per
    ``log_every`` steps instead of two ``.item()`` calls per step;
  * overall / class-balanced accuracy are computed with torch (the reference imports sklearn.metrics for two one-liners);
  * DDP as in runner_pretrain (no per-step BatchNorm buffer broadcast).
"""
import time

import numpy as np
import torch

from . import builder
from .runner_pretrain import wrap_ddp, _Single
from .. import kernels as K
from ..datasets import data_transforms
from ..pointnet2_ops import pointnet2_utils
from ..utils import dist_utils, misc
from ..utils.AverageMeter import AverageMeter
from ..utils.config import apply_fewshot_args
from ..utils.logger import get_logger, print_log

train_transforms = data_transforms.PointcloudRotate()
test_transforms = data_transforms.PointcloudScaleAndTranslate()


class Acc_Metric:
    def __init__(self, acc=0., acc_avg=0.):
        if isinstance(acc, dict):
            self.acc, self.acc_avg = acc['acc'], acc.get('acc_avg', 0.)
        elif isinstance(acc, Acc_Metric):
            self.acc, self.acc_avg = acc.acc, acc.acc_avg
        else:
            self.acc, self.acc_avg = acc, acc_avg

    def better_than(self, other):
        return self.acc > other.acc

    def state_dict(self):
        return {'acc': self.acc, 'acc_avg': self.acc_avg}


def point_all_for(npoints, train=True):
    """size of the FPS pool the random npoints-subset is drawn from (tools/runner_finetune.py:141-150, 311-318)."""
    table = {1024: 1200, 2048: 2400, 4096: 4800, 8192: 8192}
    if npoints not in table or (not train and npoints == 2048):
        raise NotImplementedError()
    return table[npoints]


def subsample(points, npoints, point_all, choice=None, fps_idx=None):
    """FPS to ``point_all`` then a random ``npoints``-subset of it (the same subset for every cloud of the batch, as the
    reference's ``fps_idx[:, np.random.choice(point_all, npoints, False)]``).  -> ([B,npoints,3], fps_idx)"""
    if points.size(1) < point_all:
        point_all = points.size(1)
    if fps_idx is None:
        fps_idx = pointnet2_utils.furthest_point_sample(points, point_all)               # [B, point_all] int32
    if choice is None:
        choice = torch.randperm(point_all, device=points.device)[:npoints]      # drawn on the device: a numpy choice would cost
    else:                                                                       # a blocking pageable H2D copy every step
        choice = torch.as_tensor(choice, device=points.device, dtype=torch.long)
    sel = fps_idx[:, choice].contiguous()
    out = pointnet2_utils.gather_operation(points.transpose(1, 2).contiguous(), sel).transpose(1, 2).contiguous()
    return out, fps_idx


def accuracy_scores(label, pred, num_classes=None):
    """overall accuracy and class-balanced accuracy in percent (sklearn accuracy_score / balanced_accuracy_score semantics:
    mean recall over the classes that occur in ``label``)."""
    label, pred = label.view(-1).long(), pred.view(-1).long()
    acc = (label == pred).float().mean().item() * 100.
    nc = int(num_classes or (max(label.max().item(), pred.max().item()) + 1))
    total = torch.bincount(label, minlength=nc).float()
    hit = torch.bincount(label[label == pred], minlength=nc).float()
    present = total > 0
    acc_avg = (hit[present] / total[present]).mean().item() * 100.
    return acc, acc_avg


def prepare_batch(points_raw, config, augment=True, choice=None, rot_u=None):
    """raw cloud batch -> network input: FPS pool, random npoints-subset, rotation (tools/runner_finetune.py:141-159)."""
    npoints = config.npoints
    pts, _ = subsample(points_raw, npoints, point_all_for(npoints), choice)
    return train_transforms(pts, rot_u) if augment else pts


def prefetch_batch(next_points_raw, config, augment=True):
    """Enqueue prepare_batch(next batch) on the auxiliary stream: the 8192 -> 1200 farthest-point sampling is a serial chain
    that keeps one workgroup per cloud busy for ~1.2 ms (32 of 256 CUs at B=32) -- it hides behind the current batch's backward.
    The result is attached to the raw tensor and picked up by the next train_step()."""
    dev = next_points_raw.device
    main, side = torch.cuda.current_stream(dev), K.side_stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        pts = prepare_batch(next_points_raw, config, augment)
        ev = torch.cuda.Event()
        ev.record(side)
    next_points_raw._act_prepared = (pts, ev)


def train_step(base_model, optimizer, points, label, config, num_iter=1, augment=True, draws=None, choice=None, rot_u=None,
               next_points=None):
    """one optimisation step on a raw device batch [B,N_raw,3]; -> (loss, acc%) detached device tensors (no host sync).
    ``next_points``: the next raw batch, whose preparation is started on the auxiliary stream before this backward."""
    pre = getattr(points, "_act_prepared", None)
    if pre is not None and choice is None and rot_u is None:
        points, ev = pre
        main = torch.cuda.current_stream(points.device)
        main.wait_event(ev)
        points.record_stream(main)
    else:
        points = prepare_batch(points, config, augment, choice, rot_u)
    ret = base_model(points, draws=draws) if draws is not None else base_model(points)
    loss, acc = base_model.module.get_loss_acc(ret, label)
    if next_points is not None and next_points.is_cuda:
        prefetch_batch(next_points, config, augment)
    loss.backward()
    if num_iter == config.step_per_update:
        if config.get('grad_norm_clip') is not None:
            plist = base_model.__dict__.get("_act_trainable")
            if plist is None:
                plist = base_model.__dict__["_act_trainable"] = [p for p in base_model.parameters() if p.requires_grad]
            torch.nn.utils.clip_grad_norm_([p for p in plist if p.grad is not None], config.grad_norm_clip, norm_type=2, foreach=True)
        optimizer.step()
        base_model.zero_grad(set_to_none=True)
    return loss.detach(), acc.detach()


def run_net(args, config, train_writer=None, val_writer=None, max_steps=None, log_every=20):
    logger = get_logger(args.log_name)
    apply_fewshot_args(args, config)                  # --way / --shot / --fold -> dataset sections (main.py:72-78); no-op when already applied
    (train_sampler, train_dataloader), (_, test_dataloader) = builder.dataset_builder(args, config.dataset.train), \
        builder.dataset_builder(args, config.dataset.val)
    base_model = builder.model_builder(config.model)
    misc.summary_parameters(base_model, logger)
    start_epoch, best_epoch = 0, 0
    best_metrics, best_metrics_vote, metrics = Acc_Metric(0., 0.), Acc_Metric(0., 0.), Acc_Metric(0., 0.)
    if args.resume:
        start_epoch, best_metric = builder.resume_model(base_model, args, logger=logger)
        best_metrics = Acc_Metric(best_metric)
    elif getattr(args, "ckpts", None) is not None:
        base_model.load_model_from_ckpt(args.ckpts)
    else:
        print_log('Training from scratch', logger=logger)
    device = torch.device("cuda", args.local_rank % max(1, torch.cuda.device_count()))
    if args.use_gpu:
        torch.cuda.set_device(device)          # every launch goes to the current device's current stream
        base_model.to(device)
    if args.distributed:
        if args.sync_bn:
            base_model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(base_model)
            print_log('Using Synchronized BatchNorm ...', logger=logger)
        base_model = wrap_ddp(base_model, args)
    else:
        base_model = _Single(base_model)
    optimizer, scheduler = builder.build_opti_sche(base_model, config)
    if args.resume:
        builder.resume_optimizer(optimizer, args, logger=logger)

    base_model.zero_grad()
    steps, log = 0, []
    for epoch in range(start_epoch, config.max_epoch + 1):
        if args.distributed:
            train_sampler.set_epoch(epoch)
        base_model.train()
        epoch_start_time = time.time()
        losses = AverageMeter(['loss', 'acc'])
        num_iter, pending = 0, []
        n_batches = len(train_dataloader)
        loader = iter(train_dataloader)
        nxt = next(loader, None)
        cur = (nxt[2][0].to(device, non_blocking=True), nxt[2][1].to(device, non_blocking=True)) if nxt is not None else None
        for idx in range(n_batches):
            if cur is None:
                break
            num_iter += 1
            n_itr = epoch * n_batches + idx
            points, label = cur
            nxt = next(loader, None)                         # one batch of look-ahead: its FPS / subset / rotation overlap this backward
            cur = (nxt[2][0].to(device, non_blocking=True), nxt[2][1].to(device, non_blocking=True)) if nxt is not None else None
            loss, acc = train_step(base_model, optimizer, points, label, config, num_iter,
                                   next_points=cur[0] if cur is not None else None)
            if num_iter == config.step_per_update:
                num_iter = 0
            if args.distributed:
                loss = dist_utils.reduce_tensor(loss, args)
                acc = dist_utils.reduce_tensor(acc, args)
            pending.append(torch.stack((loss.reshape(()), acc.reshape(()))))
            steps += 1
            last = max_steps is not None and steps >= max_steps
            if (idx + 1) % log_every == 0 or idx + 1 == n_batches or last:
                for l, a in torch.stack(pending).tolist():              # the only host sync, once per log interval
                    losses.update([l, a]); log.append((l, a))
                pending = []
                if train_writer is not None:
                    train_writer.add_scalar('Loss/Batch/Loss', log[-1][0], n_itr)
                    train_writer.add_scalar('Loss/Batch/TrainAcc', log[-1][1], n_itr)
                    train_writer.add_scalar('Loss/Batch/LR', optimizer.param_groups[0]['lr'], n_itr)
            if last:
                break
        if isinstance(scheduler, list):
            for item in scheduler:
                item.step(epoch)
        elif scheduler is not None:
            scheduler.step(epoch)
        if train_writer is not None:
            train_writer.add_scalar('Loss/Epoch/Loss', losses.avg(0), epoch)
        print_log('[Training] EPOCH: %d EpochTime : %.3f (s) [Loss,Acc] = %s lr = %e' %
                  (epoch, time.time() - epoch_start_time, ['%.4f' % l for l in losses.avg()], optimizer.param_groups[0]['lr']),
                  logger=logger)
        if epoch % args.val_freq == 0 and epoch != 0:
            metrics = validate(base_model, test_dataloader, epoch, val_writer, args, config, logger=logger)
            better = metrics.better_than(best_metrics)
            if better:
                best_metrics, best_epoch = metrics, epoch
                builder.save_checkpoint(base_model, optimizer, epoch, metrics, best_metrics, 'ckpt-best', args, logger=logger)
            if getattr(args, "vote", False):
                if metrics.acc > 92.1 or (better and metrics.acc > 91):
                    metrics_vote = validate_vote(base_model, test_dataloader, epoch, val_writer, args, config, logger=logger)
                    if metrics_vote.better_than(best_metrics_vote):
                        best_metrics_vote = metrics_vote
                        builder.save_checkpoint(base_model, optimizer, epoch, metrics, best_metrics_vote, 'ckpt-best_vote', args,
                                                logger=logger)
        builder.save_checkpoint(base_model, optimizer, epoch, metrics, best_metrics, 'ckpt-last', args, logger=logger)
        print_log('Best Val OA=%.4f  mAcc=%.4f, EPOCH: %d' % (best_metrics.acc, best_metrics.acc_avg, best_epoch), logger=logger)
        if max_steps is not None and steps >= max_steps:
            break
    print_log("[Training] Best OA=%.4f  mAcc=%.4f" % (best_metrics.acc, best_metrics.acc_avg), logger=logger)
    if train_writer is not None:
        train_writer.close()
    if val_writer is not None:
        val_writer.close()
    return log


def _collect(test_pred, test_label, args):
    test_pred, test_label = torch.cat(test_pred, dim=0), torch.cat(test_label, dim=0)
    if args.distributed:
        test_pred = dist_utils.gather_tensor(test_pred, args)
        test_label = dist_utils.gather_tensor(test_label, args)
    return accuracy_scores(test_label, test_pred)


def fps_ahead(points_raw, npoints):
    """misc.fps(points_raw, npoints) enqueued on the auxiliary stream -> (points, event): the 8192 -> 1024 sampling of the NEXT
    batch (a serial chain on one workgroup per cloud) overlaps the forward of the current one."""
    dev = points_raw.device
    main, side = torch.cuda.current_stream(dev), K.side_stream(dev)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        pts = misc.fps(points_raw, npoints)
        ev = torch.cuda.Event()
        ev.record(side)
    return pts, ev


def sampled_batches(test_dataloader, npoints, dev):
    """yields (fps-sampled points [B,npoints,3], labels) with one batch of look-ahead on the auxiliary stream"""
    it = iter(test_dataloader)
    nxt = next(it, None)
    pending = None
    if nxt is not None:
        pending = (fps_ahead(nxt[2][0].to(dev, non_blocking=True), npoints), nxt[2][1].to(dev, non_blocking=True))
    while pending is not None:
        (pts, ev), label = pending
        nxt = next(it, None)
        pending = None
        if nxt is not None:
            pending = (fps_ahead(nxt[2][0].to(dev, non_blocking=True), npoints), nxt[2][1].to(dev, non_blocking=True))
        main = torch.cuda.current_stream(dev)
        main.wait_event(ev)
        pts.record_stream(main)
        yield pts, label


def validate(base_model, test_dataloader, epoch, val_writer, args, config, logger=None):
    base_model.eval()
    test_pred, test_label = [], []
    npoints = config.npoints
    dev = next(base_model.parameters()).device
    with torch.no_grad():
        for points, label in sampled_batches(test_dataloader, npoints, dev):
            logits = base_model(points)
            test_pred.append(logits.argmax(-1).view(-1))
            test_label.append(label.view(-1))
        acc, acc_avg = _collect(test_pred, test_label, args)
        print_log('[Validation] EPOCH: %d  OA=%.4f  mAcc=%.4f' % (epoch, acc, acc_avg), logger=logger)
    if val_writer is not None:
        val_writer.add_scalar('Metric/ACC', acc, epoch)
    return Acc_Metric(acc, acc_avg)


def validate_vote(base_model, test_dataloader, epoch, val_writer, args, config, logger=None, times=10):
    """test-time voting: mean logits over ``times`` random (subset, scale-translate) views (tools/runner_finetune.py:300-366)."""
    print_log(f"[VALIDATION_VOTE] epoch {epoch}", logger=logger)
    base_model.eval()
    test_pred, test_label = [], []
    npoints = config.npoints
    dev = next(base_model.parameters()).device
    with torch.no_grad():
        for idx, (taxonomy_ids, model_ids, data) in enumerate(test_dataloader):
            points_raw = data[0].to(dev)
            point_all = point_all_for(npoints, train=False)
            fps_idx_raw = None
            votes = None
            for kk in range(times):
                points, fps_idx_raw = subsample(points_raw, npoints, point_all, fps_idx=fps_idx_raw)
                logits = base_model(test_transforms(points))
                votes = logits if votes is None else votes + logits
            test_pred.append((votes / times).argmax(-1).view(-1))
            test_label.append(data[1].to(dev).view(-1))
        acc, acc_avg = _collect(test_pred, test_label, args)
        print_log('[VALIDATION_VOTE] EPOCH: %d  OA=%.4f  mAcc=%.4f' % (epoch, acc, acc_avg), logger=logger)
    if val_writer is not None:
        val_writer.add_scalar('Metric/ACC_Vote', acc, epoch)
    return Acc_Metric(acc, acc_avg)


def test_net(args, config, vote_rounds=1):
    logger = get_logger(args.log_name)
    print_log('Tester start ... ', logger=logger)
    _, test_dataloader = builder.dataset_builder(args, config.dataset.test)
    base_model = builder.model_builder(config.model)
    builder.load_model(base_model, args.ckpts, logger=logger)
    if args.use_gpu:
        base_model.to(torch.device("cuda", args.local_rank % max(1, torch.cuda.device_count())))
    if args.distributed:
        raise NotImplementedError()
    return test(base_model, test_dataloader, args, config, logger=logger, vote_rounds=vote_rounds)


def test(base_model, test_dataloader, args, config, logger=None, vote_rounds=1):
    """plain test accuracy, then the best of ``vote_rounds`` voting passes (the reference hard-codes 299 rounds)."""
    m = validate(base_model, test_dataloader, 0, None, args, config, logger=logger)
    print_log('[TEST] OA=%.4f  mAcc=%.4f' % (m.acc, m.acc_avg), logger=logger)
    best = 0.
    for r in range(1, vote_rounds + 1):
        v = validate_vote(base_model, test_dataloader, 1, None, args, config, logger=logger, times=10)
        best = max(best, v.acc)
        print_log('[TEST_VOTE_time %d]  acc = %.4f, best acc = %.4f' % (r, v.acc, best), logger=logger)
    return m, best
