"""One-epoch loops of the pre-training stages with the reference's signatures and return dicts
(reference trainer/pretrain/pr_trainer.py:9-89,91-155): per-iteration LR schedule, loss / accum_iter, optimizer step
cadence, metric all-reduce. The matplotlib visualisation the reference calls from inside the loop is an optional
`vis_hook` (default off)."""
import torch

from ...utils import misc
from ...utils.lr_sched import adjust_learning_rate


def auto_step_executor(args, model, optimizer, loss_scaler, batch_tensors, loss_name, vis_hook=None):
    """The step executor the epoch loops build by themselves on their first batch (and keep on the model): forward + backward +
    FusedAdamW captured once as a HIP graph and replayed per batch -- an eager step of this path is ~600 launches from Python and
    runs host-bound at 2.5-3x the device time. Returns None where the captured form cannot stand in for the
    eager loop: gradient accumulation (accum_iter > 1), backward off, visualisation inside the loop, an optimizer that is not
    FusedAdamW, a CPU device, a forward replaced on the instance, or `args.graph_step = False` (the opt-out). A data-parallel run
    takes the executor's multi-GPU form with the scaler's reducer -- the contrastive stage with its key all-gather between two
    captured graphs, the Swin backbone with a collective per-step verdict on its window plan."""
    from ...engine import GraphedStep
    from ...optim import FusedAdamW
    if not getattr(args, "graph_step", True) or args.accum_iter != 1 or not args.backward:
        return None
    if not str(args.device).startswith("cuda") or not isinstance(optimizer, FusedAdamW) or "forward" in vars(model):
        return None
    if vis_hook is not None and args.visualize:
        return None
    x, y = batch_tensors
    key = (id(optimizer), loss_name, tuple(x.shape), tuple(y.shape), str(ops_dtype()))
    cached = getattr(model, "_evp_auto_executor", None)
    if cached is not None and cached[0] == key:
        return cached[1]
    if cached is not None and cached[0][:2] == key[:2] and cached[0][4] == key[4]:
        return cached[1]               # same model / optimizer / phase, another batch shape: the loop steps it eagerly through the executor
    reducer = getattr(loss_scaler, "reducer", None)
    is_rec = loss_name == "reconstruct_loss"
    noise_shape = step_prepare = None
    if is_rec:
        fwd = lambda m, x_, y_, noise: m(x_, y_, is_rec=True, noise=noise)
        if getattr(args, "masking_strategy", "random") == "random":
            noise_shape = (x.shape[0], model.backbone.num_patches)
            if getattr(model, "backbone_type", "") == "swin":
                # (data-parallel too: the ranks agree per step whether every pattern fits the captured shape, engine.GraphedStep._vote)
                step_prepare = model.backbone.enable_static_plan(x.device)
    else:
        fwd = lambda m, x_, y_, noise: m(x_, y_)
    seed = int(torch.empty((), dtype=torch.int64).random_().item())      # follows torch.manual_seed like the eager draw would
    ex = GraphedStep(model, optimizer, fwd, [x.clone(), y.clone()], noise_shape=noise_shape,
                     generator=torch.Generator(device=x.device).manual_seed(seed), reducer=reducer, step_prepare=step_prepare,
                     host_generator=torch.Generator().manual_seed(seed))
    model._evp_auto_executor = (key, ex)
    return ex


def ops_dtype():
    from ... import ops
    return ops.get_compute_dtype()


class _DevicePrefetcher:
    """Batch i + 1's host -> device copies on a side stream while step i runs (the loop used to issue them in front of the replay on
    the step's own stream: 64 MB of voxel grids per ViT-Base batch = 1.3 ms of a 10.7 ms step even from pinned memory). Wraps any
    loader of dict batches; tensors already on the device pass through. The consumer's stream waits on the copy's event."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        self.stream = torch.cuda.Stream(self.device)

    def __len__(self):
        return len(self.loader)

    def _move(self, batch):
        if not isinstance(batch, dict):          # (the reference's loaders yield dicts; anything else passes through untouched)
            return batch, None
        moved, ev = {}, None
        with torch.cuda.stream(self.stream):
            for k, v in batch.items():
                if torch.is_tensor(v) and v.device != self.device:
                    moved[k] = v.to(self.device, non_blocking=True)
                else:
                    moved[k] = v
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return moved, ev

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._move(next(it))
        except StopIteration:
            return
        while nxt is not None:
            cur, ev = nxt
            try:
                nxt = self._move(next(it))      # queued behind the previous copy, ahead of the step the consumer is about to run
            except StopIteration:
                nxt = None
            if ev is not None:
                torch.cuda.current_stream(self.device).wait_event(ev)
                for v in cur.values():
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(torch.cuda.current_stream(self.device))
            yield cur


class _DeferredLosses:
    """Per-step losses kept ON THE DEVICE. The reference reads the loss back every step (`loss.item()`,
    pr_trainer.py:46,64) and all-reduces it for the log (`utils/misc.py:406-414`): with a replayed step that is three host syncs
    per iteration and the host can never queue step N+1 while step N runs. Here every step's loss is copied into a device array
    (one tiny copy behind the replay); the meters receive the values -- one update per step, in order, so `global_avg`, the
    smoothed window and the returned dict are what per-step updates give -- when something is about to be printed or logged and
    at the end of the epoch. `args.sync_every_step = True` restores the reference's per-step read-back."""

    def __init__(self, n_iter, device):
        self.buf = torch.zeros(max(int(n_iter), 1), dtype=torch.float32, device=device)
        self.done = 0          # steps already handed to the meters
        self.n = 0

    def push(self, loss):
        if self.n >= self.buf.numel():        # a loader that yields more batches than len() promised
            self.buf = torch.cat([self.buf, torch.zeros_like(self.buf)])
        self.buf[self.n].copy_(loss.detach().reshape(()), non_blocking=True)
        self.n += 1

    def flush(self, logger, name):
        """-> the newest loss value (a host float) or None when nothing was pending; ONE device read-back."""
        if self.done == self.n:
            return None
        vals = self.buf[self.done:self.n].tolist()
        self.done = self.n
        for v in vals:
            logger.update(**{name: v})
        return vals[-1]


def _loop(args, model, data_loader, optimizer, epoch, loss_scaler, log_writer, loss_name, forward, vis_hook, step_executor=None, auto=True):
    model.train(True)
    logger = misc.MetricLogger(delimiter="  ")
    logger.add_meter("lr", misc.SmoothedValue(window_size=1, fmt="{value:.6f}"))
    header = "Epoch: [{}]".format(epoch + 1)
    optimizer.zero_grad()
    if log_writer is not None:
        print("log_dir: {}".format(log_writer.log_dir))
    n_iter = len(data_loader)
    last = None
    auto = auto and step_executor is None
    on_gpu = str(args.device).startswith("cuda")
    deferred = None            # _DeferredLosses once a step executor runs the steps and per-step syncs are not asked for
    if on_gpu and getattr(args, "prefetch_to_device", True) and not getattr(args, "sync_every_step", False) \
            and not getattr(data_loader, "yields_device_batches", False):
        # (a loader whose batches are already device tensors it will overwrite -- dataset.pretrain.gpu_event_loader -- must not be
        # asked for batch i + 1 before batch i has been consumed)
        data_loader = _DevicePrefetcher(data_loader, args.device)
    for it, batch in enumerate(logger.log_every(args, data_loader, args.print_freq, header)):
        if it % args.accum_iter == 0:
            adjust_learning_rate(optimizer, it / n_iter + epoch, args)
        events_voxel_grid = batch[0].to(args.device, non_blocking=True)
        supp = batch[1].to(args.device, non_blocking=True)
        if step_executor is None and auto:
            # the fast path is the default path: the loop captures its own step executor on the first batch (auto_step_executor)
            step_executor = auto_step_executor(args, model, optimizer, loss_scaler, (events_voxel_grid, supp), loss_name, vis_hook)
            auto = step_executor is not None
        lr = optimizer.param_groups[0]["lr"]
        log_now = (it + 1) % args.log_freq == 0
        if step_executor is not None:
            # HIP-graph replay of forward + backward + optimizer step (eventpretrain_amd/engine.py); the lr set above
            # reaches the graph through the optimizer's staged scalars
            if args.accum_iter != 1 or not args.backward:
                raise ValueError("step_executor runs one optimizer step per batch (accum_iter=1, backward=True)")
            if tuple(events_voxel_grid.shape) == tuple(step_executor.inputs[0].shape) and tuple(supp.shape) == tuple(step_executor.inputs[1].shape):
                loss = step_executor.step(events_voxel_grid, supp)
            else:                          # the short last batch of an epoch: one eager step that leaves the graph usable
                loss = step_executor.eager_step_with(events_voxel_grid, supp)
            step_now = True
            if on_gpu and not getattr(args, "sync_every_step", False):
                if deferred is None:
                    deferred = _DeferredLosses(n_iter, events_voxel_grid.device)
                deferred.push(loss)
                logger.update(lr=lr)
                # the host reads the device only when a value is needed: a progress line is due, a log point is due (the reduction of
                # the logged value is collective: every rank is here at the same iterations), or the epoch ends
                if log_now or (it + 1) % args.print_freq == 0 or it + 1 == n_iter:
                    newest = deferred.flush(logger, loss_name)
                    if log_now:
                        reduced = misc.all_reduce_mean(newest)
                        if log_writer is not None:
                            x = int((it / n_iter + epoch) * 1000)        # "epoch_1000x" axis
                            log_writer.add_scalar(loss_name, reduced, x)
                            log_writer.add_scalar("lr", lr, x)
                continue
            logger.update(**{loss_name: loss.item()})
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
        if on_gpu:
            torch.cuda.synchronize()
        logger.update(lr=lr)
        reduced = misc.all_reduce_mean(loss.item())
        if log_writer is not None and log_now and step_now:
            x = int((it / n_iter + epoch) * 1000)        # "epoch_1000x" axis
            log_writer.add_scalar(loss_name, reduced, x)
            log_writer.add_scalar("lr", lr, x)
    if deferred is not None:
        deferred.flush(logger, loss_name)
    if vis_hook is not None and args.visualize and (epoch + 1) % args.vis_train_freq == 0 and last is not None:
        vis_hook(args, *last, epoch)
    logger.synchronize_between_processes()
    print("Averaged stats:", logger)
    return {k: m.global_avg for k, m in logger.meters.items()}


def pr_rec_one_epoch(args, model, data_loader, optimizer, epoch, loss_scaler, log_writer=None, vis_hook=None,
                     step_executor=None):
    """Masked-modeling epoch: model(events_voxel_grid, sub_frame, is_rec=True). The forward / backward / optimizer calls of a
    batch run as one HIP-graph replay: the loop builds its executor on the first batch (auto_step_executor; opt out with
    args.graph_step = False) or takes the `step_executor` (an eventpretrain_amd.engine.GraphedStep on this model / optimizer)
    it is given."""
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
    # (the CLIP encoder runs inside this loop's forward: no captured step here)
    return _loop(args, model, data_loader, optimizer, epoch, loss_scaler, log_writer, "contrastive_loss", forward, vis_hook, auto=False)


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
