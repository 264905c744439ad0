"""Classification fine-tuning loops with the reference's signatures and return dicts (reference
trainer/finetune_cls/ft_cls_trainer.py:15-108 `ft_train_one_epoch`, :110-192 `ft_val`): per-iteration LR schedule,
loss / accum_iter, optimizer step cadence with optional gradient clipping, top-1 / top-5 accuracy meters."""
import time

import torch

from ... import ops
from ...utils import misc
from ...utils.lr_sched import adjust_learning_rate


def _forward(args, model, x):
    out = model(x)
    return out, out[-2]          # (..., pred, attn) for every backbone type


def _topk_accuracy(pred, label, topk=(1,)):
    """timm.utils.accuracy: percentage of samples whose label is among the k largest logits (a metric, not on the
    training path)."""
    maxk = min(max(topk), pred.shape[1])
    _, idx = pred.detach().float().topk(maxk, 1, True, True)
    hit = idx.eq(label.view(-1, 1))
    return [hit[:, :min(k, maxk)].any(1).float().sum() * (100.0 / pred.shape[0]) for k in topk]


def ft_train_one_epoch(args, model, data_loader, optimizer, epoch, loss_scaler, log_writer=None, evrepsl_model=None):
    if evrepsl_model is not None or getattr(args, "use_evrepsl", False):
        raise NotImplementedError("EvRepSL preprocessing is out of scope (SURVEY.md 2)")
    model.train(True)
    logger = misc.MetricLogger(delimiter="  ")
    logger.add_meter("lr", misc.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    header = "Epoch: [{}]".format(epoch + 1)
    optimizer.zero_grad()
    if log_writer is not None:
        print("log_dir: {}".format(log_writer.log_dir))
    n_iter = len(data_loader)
    for it, (events_voxel_grid, label, image_name) in enumerate(logger.log_every(args, data_loader, args.print_freq, header)):
        if it % args.accum_iter == 0:
            adjust_learning_rate(optimizer, it / n_iter + epoch, args)
        events_voxel_grid = events_voxel_grid.to(args.device, non_blocking=True)
        label = label.to(args.device, non_blocking=True)
        _, pred = _forward(args, model, events_voxel_grid)
        # nn.CrossEntropyLoss, or timm's LabelSmoothingCrossEntropy when args.smoothing > 0 (reference :63-66)
        loss_cls = ops.CrossEntropyFn.apply(pred, label, float(getattr(args, "smoothing", 0) or 0))
        if args.backward:
            loss_cls = loss_cls / args.accum_iter
            step_now = (it + 1) % args.accum_iter == 0
            loss_scaler(loss_cls, optimizer, clip_grad=getattr(args, "clip_grad", None), parameters=model.parameters(),
                        update_grad=step_now)
            if step_now:
                optimizer.zero_grad()
        if str(args.device).startswith("cuda"):
            torch.cuda.synchronize()
        logger.update(loss_cls=loss_cls.item())
        lr = optimizer.param_groups[0]["lr"]
        logger.update(lr=lr)
        reduced = misc.all_reduce_mean(loss_cls.item())
        if log_writer is not None and (it + 1) % args.log_freq == 0 and (it + 1) % args.accum_iter == 0:
            x = int((it / n_iter + epoch) * 1000)
            log_writer.add_scalar("loss_cls", reduced, x)
            log_writer.add_scalar("lr", lr, x)
    logger.synchronize_between_processes()
    print("Averaged stats:", logger)
    return {k: m.global_avg for k, m in logger.meters.items()}


@torch.no_grad()
def ft_val(args, model, data_loader, epoch, dataset_name="origin", evrepsl_model=None):
    if evrepsl_model is not None or getattr(args, "use_evrepsl", False):
        raise NotImplementedError("EvRepSL preprocessing is out of scope (SURVEY.md 2)")
    model.eval()
    logger = misc.MetricLogger(delimiter="  ")
    infer_time = 0.0
    for events_voxel_grid, label, image_name in logger.log_every(args, data_loader, args.print_freq, "Test:"):
        events_voxel_grid = events_voxel_grid.to(args.device, non_blocking=True)
        label = label.to(args.device, non_blocking=True)
        t0 = time.time()
        _, pred = _forward(args, model, events_voxel_grid)
        infer_time += time.time() - t0
        loss_cls = ops.CrossEntropyFn.apply(pred, label)
        logger.update(loss_cls=loss_cls.item())
        if getattr(args, "dataset_type", "") != "n-cars":
            acc1, acc5 = _topk_accuracy(pred, label, topk=(1, 5))
            logger.update(acc1=acc1.item())
            logger.update(acc5=acc5.item())
        else:
            (acc1,) = _topk_accuracy(pred, label, topk=(1,))
            logger.update(acc1=acc1.item())
    logger.synchronize_between_processes()
    if "acc5" in logger.meters:
        print("* Acc@1 {:.3f} Acc@5 {:.3f} loss_cls {:.3f}".format(logger.acc1.global_avg, logger.acc5.global_avg, logger.loss_cls.global_avg))
    else:
        print("* Acc@1 {:.3f} loss_cls {:.3f}".format(logger.acc1.global_avg, logger.loss_cls.global_avg))
    print("average inference time (ms): %.2f" % (infer_time / max(len(data_loader), 1) * 1.e3))
    return {k: m.global_avg for k, m in logger.meters.items()}
