"""Host glue of the training loop with the reference's names (reference utils/misc.py): meters, the dict->tuple
batch adapter of MetricLogger.log_every, distributed helpers, the loss-scaler/optimizer-step object, grad norm,
checkpoint dict layout. Only the pieces trainer/pretrain calls are restated."""
import datetime
import os
import time
from collections import defaultdict, deque
from pathlib import Path

import torch
import torch.distributed as dist


class SmoothedValue:
    """Windowed median/avg plus global average of a scalar series (reference utils/misc.py:24-98)."""

    def __init__(self, window_size=20, fmt=None):
        self.deque = deque(maxlen=window_size)
        self.total = 0.0
        self.count = 0
        self.fmt = fmt or "{median:.4f} ({global_avg:.4f})"

    def update(self, value, n=1):
        self.deque.append(value)
        self.count += n
        self.total += value * n

    def synchronize_between_processes(self):
        """Sum count/total over ranks (float64 all_reduce after a barrier); the window is left local."""
        if not is_dist_avail_and_initialized():
            return
        dev = "cuda" if torch.cuda.is_available() and dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([self.count, self.total], dtype=torch.float64, device=dev)
        dist.barrier()
        dist.all_reduce(t)
        self.count, self.total = int(t[0].item()), t[1].item()

    @property
    def median(self):
        return torch.tensor(list(self.deque)).median().item()

    @property
    def avg(self):
        return torch.tensor(list(self.deque), dtype=torch.float32).mean().item()

    @property
    def global_avg(self):
        return self.total / self.count

    @property
    def max(self):
        return max(self.deque)

    @property
    def value(self):
        return self.deque[-1]

    def __str__(self):
        return self.fmt.format(median=self.median, avg=self.avg, global_avg=self.global_avg, max=self.max, value=self.value)


_BATCH_KEYS = {
    "rec": ("events_voxel_grid", "sub_frame", "image_name"), "rec-n": ("events_voxel_grid", "sub_frame", "image_name"),
    "adj": ("events_voxel_grid", "clip_emb", "image_name"), "_adj": ("events_voxel_grid", "clip_emb", "image_name"),
    "con": ("events_voxel_grid", "clip_emb", "image_name"),
    "adj-n": ("events_voxel_grid", "image", "image_name"), "con-n": ("events_voxel_grid", "image", "image_name"),
    "rec+con": ("events_voxel_grid", "sub_frame", "clip_emb", "image_name"),
}


class MetricLogger:
    def __init__(self, delimiter="\t"):
        self.meters = defaultdict(SmoothedValue)
        self.delimiter = delimiter

    def update(self, **kwargs):
        for k, v in kwargs.items():
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                v = v.item()
            self.meters[k].update(v)

    def __getattr__(self, attr):
        if attr in self.meters:
            return self.meters[attr]
        if attr in self.__dict__:
            return self.__dict__[attr]
        raise AttributeError(f"'{type(self).__name__}' object has no attribute '{attr}'")

    def __str__(self):
        return self.delimiter.join(f"{n}: {m}" for n, m in self.meters.items())

    def synchronize_between_processes(self):
        for m in self.meters.values():
            m.synchronize_between_processes()

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def log_every(self, args, iterable, print_freq, header=None):
        """Yields each batch dict as the tuple the phase's trainer unpacks (reference utils/misc.py:144-167) and
        prints progress every print_freq iterations."""
        if args.phase == "finetune_cls":
            keys = ("events_voxel_grid", "label", "image_name")
        elif args.phase == "pretrain" and args.pr_phase in _BATCH_KEYS:
            keys = _BATCH_KEYS[args.pr_phase]
        else:
            raise ValueError((args.phase, getattr(args, "pr_phase", None)))
        header = header or ""
        n = len(iterable)
        start = end = time.time()
        iter_time, data_time = SmoothedValue(fmt="{avg:.4f}"), SmoothedValue(fmt="{avg:.4f}")
        for i, obj in enumerate(iterable):
            data_time.update(time.time() - end)
            yield tuple(obj[k] for k in keys)
            iter_time.update(time.time() - end)
            if (i + 1) % print_freq == 0 or i + 1 == n:
                eta = str(datetime.timedelta(seconds=int(iter_time.global_avg * (n - i))))
                msg = [header, f"[{i + 1}/{n}]", f"eta: {eta}", str(self), f"time: {iter_time}", f"data: {data_time}"]
                if torch.cuda.is_available():
                    msg.append("max mem: %.0f" % (torch.cuda.max_memory_allocated() / (1024.0 * 1024.0)))
                print(self.delimiter.join(msg))
            end = time.time()
        total = time.time() - start
        print("{} Total time: {} ({:.4f} s / it)".format(header, str(datetime.timedelta(seconds=int(total))), total / max(n, 1)))


# ----------------------------------------------------------------------------------------------------- distributed
def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(*a, **k):
    if is_main_process():
        torch.save(*a, **k)


def init_distributed_mode(args):
    """One process per GPU; reads RANK / WORLD_SIZE / LOCAL_RANK from the launcher (torchrun). backend 'nccl' is RCCL
    over xGMI on ROCm; 'gloo' when no GPU is visible (CPU rehearsal of the control flow only)."""
    if "RANK" not in os.environ or "WORLD_SIZE" not in os.environ:
        args.distributed = False
        return
    args.rank, args.world_size = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    args.gpu = int(os.environ.get("LOCAL_RANK", 0))
    args.distributed = True
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(args.gpu)
    dist.init_process_group(backend=backend, world_size=args.world_size, rank=args.rank)
    dist.barrier()


def all_reduce_mean(x):
    """Mean of a python scalar over ranks (reference utils/misc.py:406-414)."""
    world = get_world_size()
    if world <= 1:
        return x
    dev = "cuda" if torch.cuda.is_available() and dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor(x, device=dev)
    dist.all_reduce(t)
    return (t / world).item()


# ----------------------------------------------------------------------------------------------------- scaler / grad norm
def get_grad_norm_(parameters, norm_type: float = 2.0, optimizer=None):
    """2-norm of the per-parameter gradient 2-norms (reference utils/misc.py:303-315), computed by
    evp_grad_norm_multi when the FusedAdamW that owns the parameters is given."""
    if norm_type != 2.0:
        raise NotImplementedError("only the 2-norm is used by the pre-training loop")
    if optimizer is not None and hasattr(optimizer, "grad_norm"):
        return optimizer.grad_norm()
    from ..optim import FusedAdamW
    params = [p for p in ([parameters] if isinstance(parameters, torch.Tensor) else parameters) if p.grad is not None]
    if not params:
        return torch.tensor(0.0)
    return FusedAdamW([{"params": params}], lr=0.0).grad_norm()


class NativeScalerWithGradNormCount:
    """Callable with the reference's signature (utils/misc.py:274-300). The reference wraps a fp16 GradScaler; this
    path computes in bf16 / f32 with f32 accumulation, which needs no loss scaling, so the object only sequences
    backward -> (data-parallel gradient all-reduce) -> grad norm -> optimizer step. state_dict() keeps the key layout of
    a disabled GradScaler.

    `reducer` (a parallel.BucketedGradReducer, here or per call) takes the place of the reference's DDP wrap
    (main_pretrain.py:319): after the LAST accumulation backward of an optimizer step it SUM-all-reduces the gradients;
    the mean (1 / world) rides on FusedAdamW.grad_scale. The returned norm and the clip coefficient are those of the
    MEAN gradient, as under DDP."""
    state_dict_key = "amp_scaler"

    def __init__(self, reducer=None):
        self._state = {}
        self.reducer = reducer

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True, reducer=None):
        loss.backward(create_graph=create_graph)
        if not update_grad:
            return None
        reducer = reducer if reducer is not None else self.reducer
        from ..parallel import ensure_mean_grad_scale
        gs = ensure_mean_grad_scale(optimizer, reducer, "NativeScalerWithGradNormCount")
        if reducer is not None:
            reducer.finish()
        norm = get_grad_norm_(parameters, optimizer=optimizer)
        norm = norm * gs                     # the norm of the gradient the optimizer applies (the mean over ranks; 1 without a reducer)
        if clip_grad is not None:
            # torch.nn.utils.clip_grad_norm_(parameters, clip_grad) (reference utils/misc.py:289-290): every gradient times
            # min(1, clip / (total_norm + 1e-6)). The factor rides on FusedAdamW's grad_scale for this step (the kernel
            # multiplies each gradient by it as it reads it) instead of a pass over the gradients; the returned norm is
            # the pre-clip total norm, as clip_grad_norm_ returns.
            coef = min(1.0, float(clip_grad) / (float(norm) + 1e-6))
            if hasattr(optimizer, "grad_scale"):
                saved = optimizer.grad_scale
                optimizer.grad_scale = saved * coef
                try:
                    optimizer.step()
                finally:
                    optimizer.grad_scale = saved
                return norm
            torch.nn.utils.clip_grad_norm_(parameters, clip_grad)
        optimizer.step()
        return norm

    def state_dict(self):
        return dict(self._state)

    def load_state_dict(self, state_dict):
        self._state = dict(state_dict)


# ----------------------------------------------------------------------------------------------------- checkpoints
_PHASE_DIR = {"rec": "rec_dir", "adj": "adj_dir", "adj-n": "adj_n_dir", "_adj": "_adj_dir", "con": "con_dir",
              "con-n": "con_n_dir", "rec+con": "rec_and_con_dir"}


def checkpoint_dir(args):
    sub = getattr(args, _PHASE_DIR.get(getattr(args, "pr_phase", ""), ""), "") if args.phase == "pretrain" else ""
    return Path(args.output_dir) / sub / "checkpoints"


def save_model(args, epoch, model, model_without_ddp, optimizer, loss_scaler):
    """{'model','optimizer','epoch','scaler','args'} -> checkpoints/checkpoint_NN.pth on rank 0
    (dict layout of reference utils/misc.py:353-359)."""
    d = checkpoint_dir(args)
    if is_main_process():
        d.mkdir(parents=True, exist_ok=True)
    path = d / ("checkpoint_%02d.pth" % (epoch + 1))
    save_on_master({"model": model_without_ddp.state_dict(), "optimizer": optimizer.state_dict(), "epoch": epoch,
                    "scaler": loss_scaler.state_dict(), "args": args}, path)
    return path


def load_model(args, model_without_ddp, optimizer, loss_scaler):
    """Resume from args.resume (a local path; the reference's https branch needs network and is not built)."""
    if not getattr(args, "resume", ""):
        return
    if str(args.resume).startswith("https"):
        raise NotImplementedError("resuming from a URL needs network access")
    ckpt = torch.load(args.resume, map_location="cpu", weights_only=False)
    model_without_ddp.load_state_dict(ckpt["model"])
    print("Resume checkpoint %s" % args.resume)
    if "optimizer" in ckpt and "epoch" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer"])
        args.start_epoch = ckpt["epoch"] + 1
        if "scaler" in ckpt:
            loss_scaler.load_state_dict(ckpt["scaler"])
        print("With optim & sched!")


def remap_stage_checkpoint(state_dict, pr_phase):
    """Key renames of the stage hand-off (reference main_pretrain.py:265-279): older checkpoints call the backbone's
    final norm `norm_l_h` (MM stage) or `norm_h`; both become `norm_layer`."""
    old = "norm_l_h" if pr_phase in ("rec", "adj", "_adj", "adj-n") else ("norm_h" if pr_phase == "con" else None)
    if old is None:
        return state_dict
    out = {}
    for k, v in state_dict.items():
        parts = k.split(".")
        out[".".join("norm_layer" if s == old else s for s in parts)] = v
    return out
