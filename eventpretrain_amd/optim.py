"""FusedAdamW: torch.optim.AdamW semantics (the optimizer main_pretrain.py:341-343 builds) executed by ONE HIP launch
over every parameter (csrc/optim.hip), which also refreshes the bf16 weight shadows the MFMA GEMMs read.

Subclasses torch.optim.Optimizer only for the param_groups / state_dict plumbing; the arithmetic is evp_adamw_multi.
Per-group `lr` (already multiplied by `lr_scale` by utils.lr_sched.adjust_learning_rate) and `weight_decay` are read
every step. Gradient pointers are re-read every step (autograd may hand out new buffers); under HIP-graph capture use
`refresh()` outside the graph and `launch()` inside it.
"""
import math

import numpy as np
import torch

from . import _lib
from ._lib import call, stream_ptr

CHUNK = 16384


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        b = {tuple(g["betas"]) for g in self.param_groups}
        e = {g["eps"] for g in self.param_groups}
        if len(b) != 1 or len(e) != 1:
            raise ValueError("FusedAdamW needs one (betas, eps) for all groups")
        self.grad_scale = float(grad_scale)
        self._step = 0
        self._tabs = None
        self._sig = None

    # ------------------------------------------------------------------------------------------------ tables
    def _active(self):
        out = []
        for gi, g in enumerate(self.param_groups):
            for p in g["params"]:
                if p.grad is not None:
                    out.append((gi, p))
        return out

    def _build(self, act):
        dev = act[0][1].device
        n = len(act)
        for _, p in act:
            if p.dtype != torch.float32 or not p.is_contiguous() or not p.is_cuda:
                raise _lib.EvpError("FusedAdamW: parameters must be contiguous float32 tensors in device memory")
            st = self.state[p]
            if "exp_avg" not in st:
                st["step"] = torch.tensor(float(self._step))      # informational; the kernel uses self._step
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        numel = np.array([p.numel() for _, p in act], dtype=np.int64)
        chunk_t, chunk_o = [], []
        for t, ne in enumerate(numel):
            offs = np.arange(0, ne, CHUNK, dtype=np.int64)
            chunk_t.append(np.full(offs.shape, t, dtype=np.int32))
            chunk_o.append(offs)
        T = dict(
            n=n, dev=dev,
            params=torch.tensor([p.data_ptr() for _, p in act], dtype=torch.int64, device=dev),
            m=torch.tensor([self.state[p]["exp_avg"].data_ptr() for _, p in act], dtype=torch.int64, device=dev),
            v=torch.tensor([self.state[p]["exp_avg_sq"].data_ptr() for _, p in act], dtype=torch.int64, device=dev),
            numel=torch.from_numpy(numel).to(dev),
            chunk_t=torch.from_numpy(np.concatenate(chunk_t)).to(dev),
            chunk_o=torch.from_numpy(np.concatenate(chunk_o)).to(dev),
            grads=torch.zeros(n, dtype=torch.int64, device=dev),
            lp=torch.zeros(n, dtype=torch.int64, device=dev),
            wd=torch.zeros(n, dtype=torch.float32, device=dev),
            lr=torch.zeros(n, dtype=torch.float32, device=dev),
            hyper=torch.zeros(4, dtype=torch.float32, device=dev),
            norm_ws=torch.empty(sum(len(c) for c in chunk_t), dtype=torch.float32, device=dev),
            norm_out=torch.empty(1, dtype=torch.float32, device=dev),
        )
        T["n_chunks"] = int(T["chunk_t"].numel())
        # pinned staging for the per-step refresh
        T["h_grads"] = torch.zeros(n, dtype=torch.int64).pin_memory()
        T["h_lp"] = torch.zeros(n, dtype=torch.int64).pin_memory()
        T["h_wd"] = torch.zeros(n, dtype=torch.float32).pin_memory()
        T["h_lr"] = torch.zeros(n, dtype=torch.float32).pin_memory()
        T["h_hyper"] = torch.zeros(4, dtype=torch.float32).pin_memory()
        for k in ("grads", "lp", "wd", "lr", "hyper"):       # numpy views of the pinned tables: cheap element writes
            T["n_" + k] = T["h_" + k].numpy()
        T["group_of"] = np.array([gi for gi, _ in act], dtype=np.int64)
        return T

    def refresh(self, advance=True, scalars=True):
        """Host side of a step: (re)build tables if the set of parameters with gradients changed, bump the step
        counter, stage gradient pointers / lr / weight decay / bias corrections into pinned host tables and enqueue
        their (tiny) H2D copies. Called inside a HIP-graph capture, the copies become graph nodes that re-read the
        pinned tables at every replay; `stage_scalars()` then updates lr / bias corrections before each replay.
        `scalars=False`: pointer tables only (the scalar tables travel in a graph of their own, `upload_scalars`)."""
        from . import ops
        ops.flush_deferred_grads()      # normally already done by the end-of-backward callback
        act = self._active()
        if not act:
            return False
        sig = tuple(id(p) for _, p in act)
        if sig != self._sig:
            self._tabs, self._sig = self._build(act), sig
        T = self._tabs
        self._act = act
        ng, nl = T["n_grads"], T["n_lp"]
        for i, (gi, p) in enumerate(act):
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous():
                raise _lib.EvpError("FusedAdamW: gradients must be contiguous float32")
            ng[i] = g.data_ptr()
            sh = getattr(p, "_evp_lp", None)
            nl[i] = sh.data_ptr() if (sh is not None and getattr(p, "_evp_lp_version", -1) == p._version) else 0
        if scalars:
            self.stage_scalars(advance)
        for k in ("grads", "lp") + (("wd", "lr", "hyper") if scalars else ()):
            T[k].copy_(T["h_" + k], non_blocking=True)
        return True

    def upload_scalars(self):
        """The H2D copies of the per-step scalar tables (lr, weight decay, bias corrections) on their own, for a step executor
        that captures them as a separate graph in front of the step (`refresh(scalars=False)` then leaves them out)."""
        T = self._tabs
        if T is None:
            raise _lib.EvpError("upload_scalars: no tables yet (run refresh() once)")
        for k in ("wd", "lr", "hyper"):
            T[k].copy_(T["h_" + k], non_blocking=True)

    def stage_scalars(self, advance=True):
        """Write this step's lr / weight decay / bias corrections into the pinned host tables (no device work)."""
        T = self._tabs
        if advance:
            self._step += 1
        b1, b2 = self.param_groups[0]["betas"]
        go = T["group_of"]
        T["n_wd"][:] = np.array([g["weight_decay"] for g in self.param_groups], dtype=np.float32)[go]
        T["n_lr"][:] = np.array([g["lr"] for g in self.param_groups], dtype=np.float32)[go]
        step = max(self._step, 1)
        T["n_hyper"][:] = (1.0 - b1 ** step, math.sqrt(1.0 - b2 ** step), self.grad_scale, 1.0)

    # ------------------------------------------------------------------------------------------------ checkpoints
    def state_dict(self):
        """torch.optim.AdamW layout (step / exp_avg / exp_avg_sq per parameter), so checkpoints interchange with the
        reference's optimizer (utils/misc.py:353-359). The kernel keeps ONE step counter; it is written into every
        parameter's `step` entry here."""
        for st in self.state.values():
            if "exp_avg" in st:
                st["step"] = torch.tensor(float(self._step))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        steps = [float(st["step"]) for st in self.state.values() if "step" in st]
        self._step = int(max(steps)) if steps else 0
        for st in self.state.values():           # the kernel wants contiguous f32 moments
            for k in ("exp_avg", "exp_avg_sq"):
                if k in st and (st[k].dtype != torch.float32 or not st[k].is_contiguous()):
                    st[k] = st[k].float().contiguous()
        self._tabs, self._sig = None, None       # the moments are new tensors: rebuild the pointer tables

    @torch.no_grad()
    def reset_state(self):
        """Zero the moments IN PLACE and restart the step counter (the captured graphs keep pointing at these buffers)."""
        for st in self.state.values():
            if "exp_avg" in st:
                st["exp_avg"].zero_()
                st["exp_avg_sq"].zero_()
        self._step = 0

    def launch(self):
        """Device side of a step (graph-capturable): one evp_adamw_multi over all chunks."""
        T = self._tabs
        b1, b2 = self.param_groups[0]["betas"]
        call("evp_adamw_multi", T["params"].data_ptr(), T["grads"].data_ptr(), T["m"].data_ptr(), T["v"].data_ptr(),
             T["lp"].data_ptr(), T["numel"].data_ptr(), T["wd"].data_ptr(), T["lr"].data_ptr(), T["chunk_t"].data_ptr(),
             T["chunk_o"].data_ptr(), T["n_chunks"], CHUNK, 1.0, float(b1), float(b2), float(self.param_groups[0]["eps"]),
             max(self._step, 1), self.grad_scale, T["hyper"].data_ptr(), stream_ptr())

    def build_parts(self, param_lists):
        """Cut the update into parts (data-parallel plan: one per all-reduced gradient buffer, so a part's update runs
        while the next buffer is still on the wire). `param_lists`: disjoint lists of parameters; parameters with a
        gradient that appear in none form a last part. Call after refresh(); returns the number of parts."""
        T = self._tabs
        index = {id(p): i for i, (_, p) in enumerate(self._act)}
        numel = T["numel"].cpu().numpy()
        taken, parts = set(), []

        def tables(ts):
            ct = np.concatenate([np.full((int(numel[t]) + CHUNK - 1) // CHUNK, t, dtype=np.int32) for t in ts])
            co = np.concatenate([np.arange(0, int(numel[t]), CHUNK, dtype=np.int64) for t in ts])
            return torch.from_numpy(ct).to(T["dev"]), torch.from_numpy(co).to(T["dev"]), int(ct.shape[0])

        for plist in list(param_lists) + [None]:
            if plist is None:
                ts = [t for t in range(T["n"]) if t not in taken]
            else:
                ts = [index[id(p)] for p in plist if id(p) in index and index[id(p)] not in taken]
            taken.update(ts)
            parts.append(tables(ts) if ts else None)
        self._parts = parts
        return len(parts)

    def launch_part(self, i):
        """Device side of the update for part i of build_parts() (same kernel, that part's chunk table)."""
        part = self._parts[i]
        if part is None:
            return
        T = self._tabs
        b1, b2 = self.param_groups[0]["betas"]
        call("evp_adamw_multi", T["params"].data_ptr(), T["grads"].data_ptr(), T["m"].data_ptr(), T["v"].data_ptr(),
             T["lp"].data_ptr(), T["numel"].data_ptr(), T["wd"].data_ptr(), T["lr"].data_ptr(), part[0].data_ptr(),
             part[1].data_ptr(), part[2], CHUNK, 1.0, float(b1), float(b2), float(self.param_groups[0]["eps"]),
             max(self._step, 1), self.grad_scale, T["hyper"].data_ptr(), stream_ptr())

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self.refresh():
            self.launch()
        return loss

    @torch.no_grad()
    def grad_norm(self):
        """Global 2-norm of all gradients (= norm of per-parameter norms, utils/misc.py:303-315) as a device scalar.
        Uses the tables of the last refresh()."""
        if not self.refresh(advance=False):     # gradient buffers can move between steps: always re-read pointers
            raise _lib.EvpError("grad_norm: no parameter has a gradient")
        T = self._tabs
        call("evp_grad_norm_multi", T["grads"].data_ptr(), T["numel"].data_ptr(), T["chunk_t"].data_ptr(),
             T["chunk_o"].data_ptr(), T["n_chunks"], CHUNK, T["norm_ws"].data_ptr(), T["norm_out"].data_ptr(), stream_ptr())
        return T["norm_out"]
