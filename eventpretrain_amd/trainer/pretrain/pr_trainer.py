"""One-epoch loops of the pre-training stages with the reference's signatures and return dicts
(reference trainer/pretrain/pr_trainer.py:9-89,91-155): per-iteration LR schedule, loss / accum_iter, optimizer step
cadence, metric all-reduce. The matplotlib visualisation the reference calls from inside the loop is an optional
`vis_hook` (default off)."""
import torch

from ...utils import misc
from ...utils.lr_sched import adjust_learning_rate


def _loop(args, model, data_loader, optimizer, epoch, loss_scaler, log_writer, loss_name, forward, vis_hook, step_executor=None):
    model.train(True)
    logger = misc.MetricLogger(delimiter="  ")
    logger.add_meter("lr", misc.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    header = "Epoch: [{}]".format(epoch + 1)
    optimizer.zero_grad()
    if log_writer is not None:
        print("log_dir: {}".format(log_writer.log_dir))
    n_iter = len(data_loader)
    last = None
    for it, batch in enumerate(logger.log_every(args, data_loader, args.print_freq, header)):
        if it % args.accum_iter == 0:
            adjust_learning_rate(optimizer, it / n_iter + epoch, args)
        events_voxel_grid = batch[0].to(args.device, non_blocking=True)
        supp = batch[1].to(args.device, non_blocking=True)
        if step_executor is not None:
            # HIP-graph replay of forward + backward + optimizer step (eventpretrain_amd/engine.py); the lr set above
            # reaches the graph through the optimizer's staged scalars
            if args.accum_iter != 1 or not args.backward:
                raise ValueError("step_executor runs one optimizer step per batch (accum_iter=1, backward=True)")
            loss = step_executor.step(events_voxel_grid, supp)
            logger.update(**{loss_name: loss.item()})
            step_now = True
        else:
            outputs = forward(events_voxel_grid, supp)
            loss = outputs[0]
            last = (events_voxel_grid, supp, outputs, batch[-1])
            if vis_hook is not None and args.test_experiment and args.visualize:
                vis_hook(args, *last, epoch)
            logger.update(**{loss_name: loss.item()})
            loss = loss / args.accum_iter
            step_now = (it + 1) % args.accum_iter == 0
            if args.backward:
                loss_scaler(loss, optimizer, parameters=model.parameters(), update_grad=step_now)
                if step_now:
                    optimizer.zero_grad()
        if str(args.device).startswith("cuda"):
            torch.cuda.synchronize()
        lr = optimizer.param_groups[0]["lr"]
        logger.update(lr=lr)
        reduced = misc.all_reduce_mean(loss.item())
        if log_writer is not None and (it + 1) % args.log_freq == 0 and step_now:
            x = int((it / n_iter + epoch) * 1000)        # "epoch_1000x" axis
            log_writer.add_scalar(loss_name, reduced, x)
            log_writer.add_scalar("lr", lr, x)
    if vis_hook is not None and args.visualize and (epoch + 1) % args.vis_train_freq == 0 and last is not None:
        vis_hook(args, *last, epoch)
    logger.synchronize_between_processes()
    print("Averaged stats:", logger)
    return {k: m.global_avg for k, m in logger.meters.items()}


def pr_rec_one_epoch(args, model, data_loader, optimizer, epoch, loss_scaler, log_writer=None, vis_hook=None,
                     step_executor=None):
    """Masked-modeling epoch: model(events_voxel_grid, sub_frame, is_rec=True). `step_executor` (an
    eventpretrain_amd.engine.GraphedStep built on this model / optimizer) replaces the eager forward / backward /
    optimizer calls by a HIP-graph replay."""
    return _loop(args, model, data_loader, optimizer, epoch, loss_scaler, log_writer, "reconstruct_loss",
                 lambda x, y: model(x, y, is_rec=True), vis_hook, step_executor)


def pr_con_one_epoch(args, model, data_loader, optimizer, epoch, loss_scaler, log_writer=None, vis_hook=None,
                     step_executor=None):
    """Contrastive / transfer epoch: model(events_voxel_grid, clip_emb)."""
    return _loop(args, model, data_loader, optimizer, epoch, loss_scaler, log_writer, "contrastive_loss",
                 lambda x, y: model(x, y), vis_hook, step_executor)


def pr_con_n_one_epoch(args, model, preprocess, clip_model, data_loader, optimizer, epoch, loss_scaler, log_writer=None,
                       vis_hook=None):
    """Contrastive epoch with the CLIP image branch evaluated on the fly (reference trainer/pretrain/pr_trainer.py:158-223):
    batches carry the pre-processed RGB image instead of stored CLIP tokens, `clip_model.encode_image(image)` supplies the
    (B, 197, 512) token tensor. The CLIP encoder itself is the caller's frozen module (out of scope here: SURVEY.md 8c takes the
    CLIP branch as an input tensor); `preprocess` is accepted for signature compatibility and unused, as in the reference."""
    def forward(x, image):
        with torch.no_grad():
            clip_emb = clip_model.encode_image(image).to(args.device, non_blocking=True).float()
        return model(x, clip_emb)
    return _loop(args, model, data_loader, optimizer, epoch, loss_scaler, log_writer, "contrastive_loss", forward, vis_hook)


def pr_rec_and_con_one_epoch(args, model, data_loader, optimizer, epoch, loss_scaler, log_writer=None, vis_hook=None):
    """Joint epoch (reference trainer/pretrain/pr_trainer.py:225-304): one masked-modeling forward and one contrastive
    forward per batch, the two losses summed before the single backward."""
    model.train(True)
    logger = misc.MetricLogger(delimiter="  ")
    logger.add_meter("lr", misc.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    header = "Epoch: [{}]".format(epoch + 1)
    optimizer.zero_grad()
    if log_writer is not None:
        print("log_dir: {}".format(log_writer.log_dir))
    n_iter = len(data_loader)
    for it, (events_voxel_grid, sub_frame, clip_emb, image_name) in enumerate(
            logger.log_every(args, data_loader, args.print_freq, header)):
        if it % args.accum_iter == 0:
            adjust_learning_rate(optimizer, it / n_iter + epoch, args)
        events_voxel_grid = events_voxel_grid.to(args.device, non_blocking=True)
        sub_frame = sub_frame.to(args.device, non_blocking=True)
        clip_emb = clip_emb.to(args.device, non_blocking=True)
        rec = model(events_voxel_grid, sub_frame, is_rec=True)
        con = model(events_voxel_grid, clip_emb)
        if vis_hook is not None and args.test_experiment and args.visualize:
            vis_hook(args, events_voxel_grid, (sub_frame, clip_emb), (rec, con), image_name, epoch)
        logger.update(reconstruct_loss=rec[0].item())
        logger.update(contrastive_loss=con[0].item())
        loss_total = (rec[0] + con[0]) / args.accum_iter
        step_now = (it + 1) % args.accum_iter == 0
        if args.backward:
            loss_scaler(loss_total, optimizer, parameters=model.parameters(), update_grad=step_now)
            if step_now:
                optimizer.zero_grad()
        if str(args.device).startswith("cuda"):
            torch.cuda.synchronize()
        lr = optimizer.param_groups[0]["lr"]
        logger.update(lr=lr)
        r_red = misc.all_reduce_mean(rec[0].item())
        c_red = misc.all_reduce_mean(con[0].item())
        if log_writer is not None and (it + 1) % args.log_freq == 0 and step_now:
            x = int((it / n_iter + epoch) * 1000)
            log_writer.add_scalar("reconstruct_loss", r_red, x)
            log_writer.add_scalar("contrastive_loss", c_red, x)
            log_writer.add_scalar("lr", lr, x)
    logger.synchronize_between_processes()
    print("Averaged stats:", logger)
    return {k: m.global_avg for k, m in logger.meters.items()}
