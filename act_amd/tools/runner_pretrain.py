"""Stage-II pretraining loop (reference: tools/runner_pretrain.py:53-226), same ``run_net`` signature.

Differences forced by the hardware-first design, none of them visible in the config surface:
  * augmentation, masking, token compaction and the loss run on the device without host synchronisation;
  * DDP is created with broadcast_buffers=False (BatchNorm statistics stay per rank, exactly as the
    reference's non-SyncBN default) and without find_unused_parameters: the never-used heads
    (lm_head / cls_head) are kept in the state_dict but frozen, so gradients are a single bucketed RCCL
    all-reduce overlapped with backward;
  * the logged loss is all-reduced every step but read back only every ``log_every`` steps.
"""
import time
import os
import weakref

import torch
import torch.nn as nn

from . import builder
from ..datasets.data_transforms import PointcloudScaleAndTranslate
from ..utils import dist_utils, misc
from ..utils.AverageMeter import AverageMeter
from ..utils.logger import get_logger, print_log

train_transforms = PointcloudScaleAndTranslate()


class Acc_Metric:
    def __init__(self, acc=0.):
        self.acc = acc.acc if hasattr(acc, "acc") else (acc['acc'] if isinstance(acc, dict) else acc)

    def better_than(self, other):
        return self.acc > other.acc

    def state_dict(self):
        return {'acc': self.acc}


def freeze_unused_heads(model):
    """lm_head / cls_head never receive gradients in ACT_PointDistillation.forward (models/act.py:190-196 vs
    :269-309); freezing them removes the need for DDP(find_unused_parameters=True)."""
    enc = getattr(model, "ACT_encoder", None)
    if enc is not None:
        for m in (enc.lm_head, enc.cls_head):
            for p in m.parameters():
                p.requires_grad = False


def set_stack_chunk(model, chunk):
    """blocks per composite host call for every TransformerEncoder / TransformerDecoder of ``model`` (None: process default, 0: whole stack)"""
    from act_amd.models.act import TransformerEncoder, TransformerDecoder
    for m in model.modules():
        if isinstance(m, (TransformerEncoder, TransformerDecoder)):
            m.stack_chunk = chunk


def wrap_ddp(base_model, args):
    device_ids = [args.local_rank % torch.cuda.device_count()] if torch.cuda.is_available() and args.use_gpu else None
    if torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1 and "ACT_BLOCK_STACK_CHUNK" not in os.environ:
        # a block stack hands its parameter gradients to DDP when its backward call returns: in chunks of 4 blocks the bucket all-reduces of the
        # deeper blocks start while the shallower ones are still being differentiated.  Set on THIS model's encoder / decoder modules (no process-wide
        # state); bit-identical to any other chunking, the folded gradient of the shared pos included (tests/test_gpu_composite.py)
        set_stack_chunk(base_model, 4)
    return nn.parallel.DistributedDataParallel(base_model, device_ids=device_ids, broadcast_buffers=False,
                                               gradient_as_bucket_view=True, bucket_cap_mb=25)


class _Single(nn.Module):
    """exposes ``.module`` like DataParallel/DDP so builder.build_opti_sche sees the same object shape."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)


class _Announced:
    """the batch that the previous train_step on THIS model already augmented (in place) and announced to the teacher prefetch: a weak
    reference plus the tensor version right after the augmentation, kept per model in a module-level WeakKeyDictionary (two models trained in one
    process do not see each other's batches; nothing is stored ON the module, so ``torch.save(model)`` / ``mp.spawn(args=(model,))`` still pickle
    it), so the same tensor handed back as ``points`` is not augmented twice"""
    _by_model = weakref.WeakKeyDictionary()

    @staticmethod
    def mark(model, t):
        _Announced._by_model[model] = (weakref.ref(t), t._version)

    @staticmethod
    def is_marked(model, t):
        ref, version = _Announced._by_model.get(model, (None, -1))
        return ref is not None and ref() is t and version == t._version


def train_step(base_model, optimizer, points, config, num_iter=1, augment=True, draws=None, next_points=None, next_draws=None):
    """one optimisation step on a device batch [B,N,3]; returns the detached loss tensor (no host sync).

    ``next_points`` (optional): the NEXT batch.  It is augmented here and announced to the model, which starts its grouping and
    frozen-teacher forward on the auxiliary stream while this batch's backward runs; pass that same tensor as ``points`` of the
    next call (it is not augmented twice).  ``draws`` / ``next_draws`` (parity tests): injected random draws of this step / of the next step's teacher."""
    inner = base_model.module if hasattr(base_model, "module") else base_model
    if augment and not _Announced.is_marked(inner, points):
        points = train_transforms(points)
    loss = base_model(points, draws=draws) if draws is not None else base_model(points)
    if isinstance(loss, tuple):                      # ACT_PointBERT returns (moco, dvae, cutmix): summed (tools/runner_pretrain.py:140-142)
        loss = loss[0] + loss[1] + loss[2]
    if next_points is not None:
        if augment:
            next_points = train_transforms(next_points)
            _Announced.mark(inner, next_points)
        if hasattr(inner, "prefetch_teacher"):
            inner.prefetch_teacher(next_points, next_draws) if next_draws is not None else inner.prefetch_teacher(next_points)
    loss.backward()
    if num_iter == config.step_per_update:
        optimizer.step()
        optimizer.zero_grad(set_to_none=True)
    return loss.detach()


def run_net(args, config, train_writer=None, val_writer=None, max_steps=None, log_every=100):
    logger = get_logger(args.log_name)
    (train_sampler, train_dataloader), (_, test_dataloader) = builder.dataset_builder(args, config.dataset.train), \
        builder.dataset_builder(args, config.dataset.val)
    base_model = builder.model_builder(config.model)
    freeze_unused_heads(base_model)
    device = torch.device("cuda", args.local_rank % max(1, torch.cuda.device_count()))
    if args.use_gpu:
        torch.cuda.set_device(device)          # every launch goes to the current device's current stream
        base_model.to(device)
    if args.distributed:                       # one enqueue loop per rank on one host: own cores, on the GPU's NUMA node where sysfs tells (ACT_PIN_CORES=0 disables)
        from act_amd.utils.dist_utils import pin_rank
        pin = pin_rank(args.local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", getattr(args, "world_size", 1))), device.index if args.use_gpu else None)
        print_log(f'[rank {args.local_rank}] host affinity: {pin}', logger=logger)
    start_epoch, best_metrics, metrics = 0, Acc_Metric(0.), Acc_Metric(0.)
    if args.resume:
        start_epoch, best_metric = builder.resume_model(base_model, args, logger=logger)
        best_metrics = Acc_Metric(best_metric)
    elif args.start_ckpts is not None:
        builder.load_model(base_model, args.start_ckpts, logger=logger)
    if args.distributed:
        if args.sync_bn:
            base_model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(base_model)
            print_log('Using Synchronized BatchNorm ...', logger=logger)
        base_model = wrap_ddp(base_model, args)
        print_log('Using Distributed Data parallel ...', logger=logger)
    else:
        base_model = _Single(base_model)
    optimizer, scheduler = builder.build_opti_sche(base_model, config)
    if args.resume:
        builder.resume_optimizer(optimizer, args, logger=logger)

    base_model.zero_grad()
    steps = 0
    losses_log = []
    for epoch in range(start_epoch, config.max_epoch + 1):
        if args.distributed:
            train_sampler.set_epoch(epoch)
        base_model.train()
        epoch_start_time = batch_start_time = time.time()
        batch_time, data_time, losses = AverageMeter(), AverageMeter(), AverageMeter(['Loss'])
        num_iter = 0
        n_batches = len(train_dataloader)
        pending = []
        npoints = config.dataset.train.others.npoints
        dataset_name = config.dataset.train._base_.NAME

        def to_points(data):
            if dataset_name == 'ShapeNet':
                pts = data.to(device, non_blocking=True)
            elif dataset_name == 'ModelNet':
                pts = misc.fps(data[0].to(device, non_blocking=True), npoints)
            else:
                raise NotImplementedError(f'Train phase do not support {dataset_name}')
            assert pts.size(1) == npoints
            return pts

        loader = iter(train_dataloader)
        nxt = next(loader, None)
        points = to_points(nxt[2]) if nxt is not None else None
        for idx in range(n_batches):
            if points is None:
                break
            num_iter += 1
            n_itr = epoch * n_batches + idx
            data_time.update(time.time() - batch_start_time)
            nxt = next(loader, None)                         # one batch of look-ahead: its teacher forward overlaps this backward
            next_points = to_points(nxt[2]) if nxt is not None else None
            loss = train_step(base_model, optimizer, points, config, num_iter, next_points=next_points)
            points = next_points
            if num_iter == config.step_per_update:
                num_iter = 0
            if args.distributed:
                loss = dist_utils.reduce_tensor(loss, args)
            pending.append(loss)
            steps += 1
            if idx % log_every == 0 or (max_steps is not None and steps >= max_steps):
                vals = torch.stack(pending).tolist()           # the only host sync, once per log interval
                pending = []
                for v in vals:
                    losses.update([v])
                    losses_log.append(v)
                if train_writer is not None:
                    train_writer.add_scalar('Loss/Batch/Loss', vals[-1], n_itr)
                    train_writer.add_scalar('Loss/Batch/LR', optimizer.param_groups[0]['lr'], n_itr)
                batch_time.update(time.time() - batch_start_time)
                print_log('[Epoch %d/%d][Batch %d/%d] BatchTime = %.3f (s) DataTime = %.3f (s) [Losses] = %s lr = %.6f' %
                          (epoch, config.max_epoch, idx + 1, n_batches, batch_time.val(), data_time.val(),
                           ['%.4f' % l for l in losses.val()], optimizer.param_groups[0]['lr']), logger=logger)
            batch_start_time = time.time()
            if max_steps is not None and steps >= max_steps:
                break
        if pending:
            for v in torch.stack(pending).tolist():
                losses.update([v]); losses_log.append(v)
        if scheduler is not None:
            scheduler.step(epoch)
        if train_writer is not None:
            train_writer.add_scalar('Loss/Epoch/Loss_1', losses.avg(0), epoch)
        print_log('[Training] EPOCH: %d EpochTime = %.3f (s) Losses = %s lr = %.6f' %
                  (epoch, time.time() - epoch_start_time, ['%.4f' % l for l in losses.avg()], optimizer.param_groups[0]['lr']),
                  logger=logger)
        builder.save_checkpoint(base_model, optimizer, epoch, metrics, best_metrics, 'ckpt-last', args, logger=logger)
        if epoch % 25 == 0 and epoch >= 250:
            builder.save_checkpoint(base_model, optimizer, epoch, metrics, best_metrics, f'ckpt-epoch-{epoch:03d}', args, logger=logger)
        if (config.max_epoch - epoch) < 3:
            builder.save_checkpoint(base_model, optimizer, epoch, metrics, best_metrics, f'ckpt-epoch-{epoch:03d}', args, logger=logger)
        if max_steps is not None and steps >= max_steps:
            break
    if train_writer is not None:
        train_writer.close()
    if val_writer is not None:
        val_writer.close()
    return losses_log
