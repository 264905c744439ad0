"""Data-parallel gradient reduction for one process per GPU: RCCL (torch.distributed backend "nccl" on ROCm) over
xGMI. Replaces the reference's DistributedDataParallel wrap (main_pretrain.py:319; collective C1 in SURVEY.md 2.2).

Where the gradients live decides the shape of the collective:
  * weight / bias gradients are written by the deferred grouped launches (ops._DeferredGrads) straight into a few
    large flat f32 buffers (`param.grad` are views) -> they are SUM-all-reduced IN PLACE, no packing pass;
  * the remaining small gradients (LayerNorm / BatchNorm affine, mask token) are packed into one small flat bucket
    with a multi-tensor copy and reduced with a single call.
Few large messages suit xGMI (point-to-point links, a ring step is per-link bound) better than DDP's 25 MB buckets.
The 1/world mean is folded into FusedAdamW's grad_scale. Buffers (MoCo queue, BN statistics) are NOT broadcast each
forward: the reference's accidental per-forward DDP buffer broadcast is deliberately not
reproduced. `make_static_plan()` freezes the buffer set for HIP-graph replay (gradient addresses are then static).
"""
import torch
import torch.distributed as dist


def ensure_mean_grad_scale(optimizer, reducer, who="data-parallel step"):
    """The reducers SUM gradients over ranks; the MEAN that DistributedDataParallel applies (main_pretrain.py:319) rides on
    FusedAdamW.grad_scale. Set it when the caller left the default 1.0, refuse anything else than 1 / world -- a silent
    world-times-larger update otherwise. Returns the scale in force. Shared by NativeScalerWithGradNormCount and
    engine.GraphedStep."""
    gs = float(getattr(optimizer, "grad_scale", 1.0))
    if reducer is None:
        return gs
    world = int(getattr(reducer, "world_size", 1))
    if hasattr(optimizer, "grad_scale"):
        if gs == 1.0 and world > 1:
            optimizer.grad_scale = gs = 1.0 / world
        elif abs(gs * world - 1.0) > 1e-6:
            raise ValueError("%s: with a gradient reducer the optimizer's grad_scale must be 1 / world_size (= %g), got %g"
                             % (who, 1.0 / world, gs))
    elif world > 1:
        raise ValueError("%s: a gradient reducer needs an optimizer with a grad_scale (FusedAdamW)" % who)
    return gs


class _Plan:
    def __init__(self, flats, others, bucket, views, group):
        self.flats, self.others, self.bucket, self.views, self.group = flats, others, bucket, views, group
        self.srcs = [p.grad for p in others]        # where backward writes the small gradients
        for p, v in zip(others, views):              # the optimizer reads the reduced bucket views
            p.grad = v

    def run(self):
        works = [dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.group, async_op=True) for f in self.flats]
        if self.others:
            torch._foreach_copy_(self.views, self.srcs)
            works.append(dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()                                  # stream wait, no host sync


class OverlappedPlan:
    """HIP-graph data-parallel step, gradient part: the deferred weight-gradient work was kept OUT of the captured
    forward+backward graph and cut into chunks (ops.build_deferred_plan); run() launches chunk c and immediately issues the
    asynchronous all-reduce of the flat buffer chunk c-... wrote, so RCCL (on its own stream, ordered after the chunk by
    the event torch.distributed records at the call) moves chunk c over xGMI while chunk c+1's GEMMs run. Only the last
    chunk's all-reduce is exposed. The small gradients autograd produced inside the graph go through the packed bucket
    first."""

    def __init__(self, steps, others, bucket, views, group, two_streams=True, early_steps=()):
        """`early_steps`: weight-gradient launches whose operands are complete before the rest of the backward has run -- the
        DECODER's, once the step executor has cut the backward at the encoder / decoder boundary (engine.BackwardCut). run_early()
        launches them and their all-reduces on a side stream; the caller then replays the encoder's backward on the main stream,
        so these buffers cross xGMI under ~3 ms of compute instead of competing for the step's tail."""
        self.steps, self.others, self.bucket, self.views, self.group = steps, others, bucket, views, group
        self.early_steps = list(early_steps)
        self._early_works, self._early_done = [], False
        self.srcs = [p.grad for p in others]
        for p, v in zip(others, views):
            p.grad = v
        # Chunks of one round write disjoint gradients: they alternate between two side streams, so the ramp-down of
        # chunk c (its last, partly filled round of workgroups) is filled by the first workgroups of chunk c+1 instead
        # of idling at a kernel boundary. Each chunk's all-reduce is issued from the stream the chunk ran on.
        self.streams = (torch.cuda.Stream(), torch.cuda.Stream()) if (two_streams and torch.cuda.is_available()) else None
        self.opt, self.update_stream = None, None
        self.timing = None          # a list: run() appends (event at "all chunks computed", event at "all reduced + updated")

    def attach_optimizer(self, opt, params):
        """Run the optimizer update in parts, each right after its gradient buffer has been all-reduced (on a third
        stream), instead of one launch after the last all-reduce: part 0 = the small-gradient bucket, part 1 + i = the
        gradients step i completes, last part = anything else. Needs the two-stream form and opt.refresh() done."""
        if self.streams is None:
            return False

        def owners(flats):
            spans = [(f.data_ptr(), f.data_ptr() + f.numel() * f.element_size()) for f in flats]
            return [p for p in params if p.grad is not None and any(lo <= p.grad.data_ptr() < hi for lo, hi in spans)]

        lists = [list(self.others)] + [owners(st.flats) for st in self.early_steps] + [owners(st.flats) for st in self.steps]
        self.n_parts = opt.build_parts(lists)
        self.opt, self.update_stream = opt, torch.cuda.Stream()
        return True

    def run_early(self):
        """After the forward + decoder-backward graph: the early weight-gradient launches and their all-reduces (and, with the
        optimizer attached, the update of their parameters) on side streams. Returns at once; run() joins."""
        self._early_works, self._early_done = [], True
        if not self.early_steps:
            return
        if self.streams is None:
            for st in self.early_steps:
                st.run()
                for f in st.flats:
                    self._early_works.append(dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        cur = torch.cuda.current_stream()
        side, upd = self.streams[1], self.update_stream
        side.wait_stream(cur)
        if upd is not None:
            upd.wait_stream(cur)                      # this step's lr / bias-correction tables are on the device
        for k, st in enumerate(self.early_steps):
            mine = []
            with torch.cuda.stream(side):
                st.run()
                for f in st.flats:
                    mine.append(dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self._early_works += mine
            if upd is not None:
                with torch.cuda.stream(upd):
                    for w in mine:
                        w.wait()
                    if not mine:
                        upd.wait_stream(side)
                    self.opt.launch_part(1 + k)

    def run(self):
        if self.early_steps and not self._early_done:
            self.run_early()                          # a caller without a split backward: everything after the one graph
        self._early_done = False
        works = list(self._early_works)
        self._early_works = []
        n_early = len(self.early_steps)
        bucket_work = None
        if self.others:
            torch._foreach_copy_(self.views, self.srcs)
            bucket_work = dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            works.append(bucket_work)
        if self.streams is None:
            for st in self.steps:
                st.run()
                for f in st.flats:
                    works.append(dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            cur = torch.cuda.current_stream()
            for s in self.streams:
                s.wait_stream(cur)                    # the forward+backward graph has finished
            upd = self.update_stream
            if upd is not None:
                upd.wait_stream(cur)                  # this step's lr / bias-correction tables are on the device
                if bucket_work is not None:
                    with torch.cuda.stream(upd):
                        bucket_work.wait()
                        self.opt.launch_part(0)
            prev_round, k = None, n_early
            for st in self.steps:
                r = getattr(st, "round", 0)
                if prev_round is not None and r != prev_round:      # a later round accumulates onto the earlier one's results
                    self.streams[0].wait_stream(self.streams[1])
                    self.streams[1].wait_stream(self.streams[0])
                prev_round = r
                mine = []
                with torch.cuda.stream(self.streams[k & 1]):
                    st.run()
                    for f in st.flats:
                        mine.append(dist.all_reduce(f, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                works += mine
                if upd is not None:
                    with torch.cuda.stream(upd):
                        for w in mine:
                            w.wait()                  # the update stream waits for this buffer only
                        if not mine:
                            upd.wait_stream(self.streams[k & 1])
                        self.opt.launch_part(1 + k)
                k += 1
            for s in self.streams:
                cur.wait_stream(s)
            if self.timing is not None:
                t_a = torch.cuda.Event(enable_timing=True)
                t_a.record(cur)
            if upd is not None:
                with torch.cuda.stream(upd):
                    if not self.others:
                        self.opt.launch_part(0)
                    self.opt.launch_part(self.n_parts - 1)
                cur.wait_stream(upd)
        for w in works:
            w.wait()                                  # stream wait, no host sync
        if self.timing is not None and self.streams is not None:
            t_b = torch.cuda.Event(enable_timing=True)
            t_b.record(torch.cuda.current_stream())
            self.timing.append((t_a, t_b))


class BucketedGradReducer:
    def __init__(self, params, bucket_mb=64.0, process_group=None, buffers=None, broadcast=True):
        """`params`: the model's parameters; `buffers`: its buffers (BatchNorm statistics, MoCo queue), broadcast once
        together with the parameters. With `broadcast` (default) every rank starts from rank 0's weights, which is what
        DistributedDataParallel's constructor does for the reference (main_pretrain.py:319, seeds differ per rank,
        main_pretrain.py:174); the bf16 weight shadows are refreshed afterwards."""
        from . import ops
        self.group = process_group
        params = list(params)
        self.params = [p for p in params if p.requires_grad]
        self.bucket_mb = bucket_mb                    # kept for API compatibility; flat buffers are reduced whole
        self._bucket, self._sig = None, None
        ops.track_deferred_flat_buffers(self)         # this object consumes them in _collect / make_overlapped_plan
        if broadcast and dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
            src = dist.get_global_rank(process_group, 0) if process_group is not None else 0
            with torch.no_grad():
                for t in params + list(buffers or []):
                    dist.broadcast(t.data, src=src, group=process_group)
            if params and params[0].is_cuda:
                ops.refresh_lp_shadows(params)

    @classmethod
    def for_module(cls, module, **kw):
        """Reducer over a module's parameters with its buffers included in the initial broadcast."""
        return cls(module.parameters(), buffers=module.buffers(), **kw)

    @property
    def world_size(self):
        return dist.get_world_size(self.group) if (dist.is_available() and dist.is_initialized()) else 1

    def _collect(self):
        from . import ops
        ops.flush_deferred_grads()
        flats = ops.take_deferred_flat_buffers()
        spans = [(f.data_ptr(), f.data_ptr() + f.numel() * f.element_size()) for f in flats]

        def in_flat(t):
            a = t.data_ptr()
            return any(lo <= a < hi for lo, hi in spans)

        others = [p for p in self.params if p.grad is not None and not in_flat(p.grad)]
        bucket = views = None
        if others:
            sig = tuple((id(p), p.numel()) for p in others)
            if sig != self._sig:
                sizes = [(p.numel() + 63) // 64 * 64 for p in others]
                self._bucket = torch.zeros(sum(sizes), dtype=torch.float32, device=others[0].device)
                self._offs = [sum(sizes[:i]) for i in range(len(sizes))]
                self._sig = sig
            bucket = self._bucket
            views = [bucket[o:o + p.numel()].view_as(p) for o, p in zip(self._offs, others)]
        return flats, others, bucket, views

    def finish(self):
        """Eager mode: call after every backward. All-reduces (SUM) every gradient, waits on the current stream."""
        _Plan(*self._collect(), self.group).run()

    def make_overlapped_plan(self, n_chunks=4, early_steps=()):
        """HIP-graph mode with overlap: call once after a forward+backward captured under ops.hold_deferred_grads(True).
        `early_steps`: ops.build_deferred_plan(...) of what an earlier part of a split backward queued (OverlappedPlan)."""
        from . import ops
        steps = ops.build_deferred_plan(n_chunks)
        if any(getattr(st, "round", 0) != 0 for st in list(steps) + list(early_steps)):
            # a parameter with several contributions per backward (rec+con) would be all-reduced before its later rounds
            raise RuntimeError("overlapped data-parallel plan: parameters with more than one gradient contribution per step "
                               "are not supported; use the eager reducer (BucketedGradReducer.finish)")
        flats = ops.take_deferred_flat_buffers()
        spans = [(f.data_ptr(), f.data_ptr() + f.numel() * f.element_size()) for f in flats]
        others = [p for p in self.params if p.grad is not None and not any(lo <= p.grad.data_ptr() < hi for lo, hi in spans)]
        bucket = views = None
        if others:
            sizes = [(p.numel() + 63) // 64 * 64 for p in others]
            bucket = torch.zeros(sum(sizes), dtype=torch.float32, device=others[0].device)
            offs = [sum(sizes[:i]) for i in range(len(sizes))]
            views = [bucket[o:o + p.numel()].view_as(p) for o, p in zip(offs, others)]
        return OverlappedPlan(steps, others, bucket, views, self.group, early_steps=early_steps)

    def make_static_plan(self):
        """HIP-graph mode: call once after the captured backward; the returned plan's run() reduces the same buffers
        after every replay."""
        return _Plan(*self._collect(), self.group)
