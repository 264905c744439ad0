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


def auto_ft_step_executor(args, model, optimizer, loss_scaler, batch_tensors):
    """The fine-tuning step (dense backbone -> token mean -> head -> cross entropy -> backward -> FusedAdamW) captured once as a HIP
    graph, as trainer.pretrain.pr_trainer.auto_step_executor does for the pre-training loops. The stochastic
    regularisers of the recipe replay correctly: the DropPath draws and the dropout keys come from torch's device generator, which
    advances under replay (ops.draw_block_drop / draw_drop_seed). Returns None where the captured form cannot stand in: gradient
    accumulation, backward off, gradient clipping (its coefficient is a host decision on the gradient norm), an optimizer that is
    not FusedAdamW, a CPU device, or `args.graph_step = False`."""
    from ...engine import GraphedStep
    from ...optim import FusedAdamW
    if not getattr(args, "graph_step", True) or args.accum_iter != 1 or not args.backward or getattr(args, "clip_grad", None) is not None:
        return None
    if not str(args.device).startswith("cuda") or not isinstance(optimizer, FusedAdamW) or "forward" in vars(model):
        return None
    x, y = batch_tensors
    key = (id(optimizer), "loss_cls", tuple(x.shape), tuple(y.shape), str(ops.get_compute_dtype()))
    cached = getattr(model, "_evp_auto_executor", None)
    if cached is not None and (cached[0] == key or (cached[0][:2] == key[:2] and cached[0][4] == key[4])):
        return cached[1]
    smoothing = float(getattr(args, "smoothing", 0) or 0)
    fwd = lambda m, x_, y_, noise: (ops.CrossEntropyFn.apply(m(x_)[-2], y_, smoothing),)
    ex = GraphedStep(model, optimizer, fwd, [x.clone(), y.clone()], reducer=getattr(loss_scaler, "reducer", None))
    model._evp_auto_executor = (key, ex)
    return ex


def ft_train_one_epoch(args, model, data_loader, optimizer, epoch, loss_scaler, log_writer=None, evrepsl_model=None):
    """One fine-tuning epoch (reference ft_cls_trainer.py:15-108). On the GPU the steps run as HIP-graph replays (auto_ft_step_executor),
    the losses stay on the device between log points and batch i + 1 is uploaded while step i runs; `args.graph_step = False` or
    `args.sync_every_step = True` give the eager / per-step-synchronised loop."""
    if evrepsl_model is not None or getattr(args, "use_evrepsl", False):
        raise NotImplementedError("EvRepSL preprocessing is out of scope (SURVEY.md 2)")
    on_gpu = str(args.device).startswith("cuda")
    if on_gpu and not getattr(args, "sync_every_step", False):
        done = _graphed_epoch(args, model, data_loader, optimizer, epoch, loss_scaler, log_writer)
        if done is not None:
            return done
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


def _graphed_epoch(args, model, data_loader, optimizer, epoch, loss_scaler, log_writer):
    """The epoch through the step executor; None when the first batch shows that the captured form does not apply (the caller then
    runs the eager loop on the same loader from its start)."""
    from ..pretrain.pr_trainer import _DeferredLosses, _DevicePrefetcher
    model.train(True)
    cached = getattr(model, "_evp_auto_executor", None)
    if cached is not None and cached[0][:2] == (id(optimizer), "loss_cls") and cached[0][4] == str(ops.get_compute_dtype()):
        ex = cached[1]                     # built by an earlier epoch: no need to look at a batch first
        if auto_ft_step_executor(args, model, optimizer, loss_scaler, tuple(ex.inputs)) is not ex:
            return None
    else:
        it0 = iter(data_loader)
        try:
            first = next(it0)
        except StopIteration:
            return None
        del it0
        if not isinstance(first, dict):
            return None
        x0 = first["events_voxel_grid"].to(args.device, non_blocking=True)
        y0 = first["label"].to(args.device, non_blocking=True)
        ex = auto_ft_step_executor(args, model, optimizer, loss_scaler, (x0, y0))
        if ex is None:
            return None
    x0 = ex.inputs[0]
    logger = misc.MetricLogger(delimiter="  ")
    logger.add_meter("lr", misc.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    header = "Epoch: [{}]".format(epoch + 1)
    optimizer.zero_grad()
    if log_writer is not None:
        print("log_dir: {}".format(log_writer.log_dir))
    n_iter = len(data_loader)
    deferred = _DeferredLosses(n_iter, x0.device)
    loader = _DevicePrefetcher(data_loader, args.device) if getattr(args, "prefetch_to_device", True) else data_loader
    for it, (events_voxel_grid, label, image_name) in enumerate(logger.log_every(args, loader, args.print_freq, header)):
        adjust_learning_rate(optimizer, it / n_iter + epoch, args)
        events_voxel_grid = events_voxel_grid.to(args.device, non_blocking=True)
        label = label.to(args.device, non_blocking=True)
        if tuple(events_voxel_grid.shape) == tuple(ex.inputs[0].shape):
            loss = ex.step(events_voxel_grid, label)
        else:                              # the short last batch of an epoch
            loss = ex.eager_step_with(events_voxel_grid, label)
        deferred.push(loss)
        lr = optimizer.param_groups[0]["lr"]
        logger.update(lr=lr)
        log_now = (it + 1) % args.log_freq == 0
        if log_now or (it + 1) % args.print_freq == 0 or it + 1 == n_iter:
            newest = deferred.flush(logger, "loss_cls")
            if log_now:
                reduced = misc.all_reduce_mean(newest)
                if log_writer is not None:
                    x = int((it / n_iter + epoch) * 1000)
                    log_writer.add_scalar("loss_cls", reduced, x)
                    log_writer.add_scalar("lr", lr, x)
    deferred.flush(logger, "loss_cls")
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
