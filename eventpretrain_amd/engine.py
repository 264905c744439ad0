"""Step executor: the pre-training step (forward + backward + FusedAdamW) captured once in HIP graphs and replayed.

An eager step of this path is ~600 kernel launches from Python -- launch-bound at about twice the device time. The
executor runs a few eager steps (allocator and autograd warm-up, on the capture stream), captures

    1 rank : [forward + backward + AdamW]                                               one graph
    N ranks: [forward + backward] -> weight-gradient GEMMs in chunks, each chunk's flat buffer all-reduced (RCCL, in
             place) while the next chunk computes -> [AdamW]

and from then on a step is: copy the batch into the static input buffers, draw the mask noise into its static buffer,
stage the optimizer's per-step scalars (lr, bias corrections) into the pinned table the graph's own H2D node re-reads,
replay. Collectives stay outside the graphs. If capture fails the executor says so and keeps stepping eagerly.

Replaces nothing in the reference (its loop is eager PyTorch, trainer/pretrain/pr_trainer.py:20-76); it is the
MI355X-side answer to "launch-bound inner loop -> HIP graph". `trainer.pretrain.pr_trainer.pr_rec_one_epoch(...,
step_executor=...)` uses it when given one."""
import os

import torch

from . import ops

_DP_BACKWARD_CUT_DEFAULT = "1"


class BackwardCut:
    """Splits one backward pass at a tensor: forward calls cut(t) where the model hands `t` from one part to the next, and gets a
    detached leaf in its place; loss.backward() then stops at the leaf (gradients of everything AFTER the cut, and the leaf's own),
    resume() runs the rest: t.backward(leaf.grad). Numerically the same backward, as two autograd calls -- which is what lets the
    data-parallel executor capture them as two HIP graphs and put a collective between them."""

    def __init__(self):
        self.src = self.leaf = None

    def __call__(self, t):
        self.src = t
        self.leaf = t.detach().requires_grad_(True)
        return self.leaf

    def resume(self):
        self.src.backward(self.leaf.grad)
        self.src = self.leaf = None


class ForwardCollectives:
    """The collectives INSIDE a model's forward (the contrastive stage's key all-gather, pr_hub_model.py:170-188,248-259; the
    reference-faithful per-forward buffer broadcast), taken out of the captured HIP graphs. The step executor hands this object to
    the model for the capture (`model.set_collective_hook`); the forward then calls, in place of the collective:

      gather(t)             the all-gather's RESULT feeds the rest of the forward (in-batch InfoNCE): the capture is SPLIT here --
                            the running graph ends, the next one begins -- and the call returns the static gathered buffer; at
                            replay the executor runs all_gather_into_tensor(out, t) between the two graphs;
      gather(t, then=fn)    the result only feeds `fn(gathered)`, which nothing else in this step depends on (the MoCo queue's
                            enqueue): nothing is split; at replay the all-gather is issued asynchronously right after the forward
                            graph, crosses xGMI under the backward graph, and `fn` runs after it;
      pre_forward(fn)       `fn()` (a broadcast of module buffers) runs eagerly before the first graph of every step.
    Eager steps (warm-up, fall-back) run with the hook removed, i.e. with the model's own collectives."""

    def __init__(self, seq):
        self.seq = seq            # the _GraphSeq being captured
        self.pre, self.post = [], []

    def pre_forward(self, fn):
        self.pre.append(fn)

    def gather(self, t, then=None):
        import torch.distributed as dist
        world = dist.get_world_size()
        t = t.contiguous()
        out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        if then is not None:
            self.post.append((t, out, then))
            return None
        self.seq.split(lambda: dist.all_gather_into_tensor(out, t))
        return out


class _GraphSeq:
    """A captured step as a SEQUENCE of HIP graphs sharing one memory pool, with eager work (a collective) between two of them:
    split(fn) ends the graph being captured, runs fn() once now (every rank meets the collective during capture as it will at
    replay) and begins the next graph; replay() = graph, fn, graph, ..."""

    def __init__(self, stream):
        self.stream, self.graphs, self.between, self.cur, self.pool = stream, [], [], None, None

    def begin(self):
        g = torch.cuda.CUDAGraph()
        kw = {} if self.pool is None else {"pool": self.pool}
        g.capture_begin(capture_error_mode="thread_local", **kw)      # thread_local: RCCL's watchdog thread polls its events meanwhile
        self.cur = g

    def end(self):
        self.cur.capture_end()
        self.graphs.append(self.cur)
        self.between.append(None)
        if self.pool is None:
            self.pool = self.cur.pool()
        self.cur = None

    def abort(self):
        if self.cur is not None:
            try:
                self.cur.capture_end()
            except Exception:
                pass
            self.cur = None

    def split(self, fn):
        self.end()
        fn()
        self.between[-1] = fn
        self.begin()

    def replay(self):
        for g, fn in zip(self.graphs, self.between):
            g.replay()
            if fn is not None:
                fn()


class GraphedStep:
    def __init__(self, model, optimizer, forward, static_inputs, noise_shape=None, generator=None, reducer=None,
                 use_graph=True, warmup=2, wgrad_chunks=4, step_prepare=None, host_generator=None, backward_cut=None):
        """forward(model, *static_inputs, noise) -> tuple whose first item is the loss. `static_inputs`: device tensors
        with the batch's shapes (overwritten by `step(...)` when new data is passed). `noise_shape`: (B, L) of the
        masking noise, or None when the model draws none (density masking, contrastive stage).
        `step_prepare(noise_cpu) -> bool` (models whose launch geometry depends on the noise: the Swin backbone's window
        plan, `SwinTransformer.enable_static_plan`): the noise is then drawn on the HOST (`host_generator`), handed to the
        hook before anything is launched and copied to the static device buffer; False = the captured graph cannot serve
        this step, which then runs eagerly with the same noise."""
        self.model, self.opt, self.forward, self.reducer = model, optimizer, forward, reducer
        # data-parallel form only: cut the backward at the encoder / decoder boundary (two graphs, the decoder's gradients all-reduced
        # under the encoder's backward). None = the environment's EVP_DP_BACKWARD_CUT (see _capture for the default and why).
        self.backward_cut = (os.environ.get("EVP_DP_BACKWARD_CUT", _DP_BACKWARD_CUT_DEFAULT) != "0") if backward_cut is None else bool(backward_cut)
        if reducer is not None:
            # the reducer SUMs over ranks: the mean must be in force before anything is captured or stepped (ADVICE r3 -- the epoch
            # loops hand their executor the scaler's reducer without ever calling the scaler)
            from .parallel import ensure_mean_grad_scale
            ensure_mean_grad_scale(optimizer, reducer, "GraphedStep")
        # step_prepare + reducer (the data-parallel Swin step): the hook decides per step, from THIS rank's noise, whether the captured
        # launch shape fits. A rank that fell back to an eager step while the others replay would leave the all-reduces unmatched, so
        # the decision is made COLLECTIVELY (_vote: a host-side MIN over ranks of one int per step, on a gloo group -- no device sync)
        # and the fall-back has a data-parallel form (_eager_fallback): either every rank replays or every rank steps eagerly.
        self._vote_group = None
        if step_prepare is not None and reducer is not None and reducer.world_size > 1:
            import torch.distributed as dist
            self._vote_group = dist.group.WORLD if dist.get_backend() == "gloo" else dist.new_group(backend="gloo")
        self.inputs = [t for t in static_inputs]
        dev = self.inputs[0].device
        self.gen = generator if generator is not None else torch.Generator(device=dev)
        self.noise = torch.empty(*noise_shape, device=dev) if noise_shape is not None else None
        self.step_prepare = step_prepare
        self.host_gen = host_generator if host_generator is not None else torch.Generator()
        if step_prepare is not None:
            if self.noise is None:
                raise ValueError("step_prepare needs noise_shape")
            self._noise_pins = [torch.empty(*noise_shape).pin_memory() for _ in range(3)]
            self._noise_events, self._noise_turn, self._noise_cpu = [None] * 3, 0, None
        self.eager_fallbacks = 0
        self.noise_feed = None         # optional iterator of (B, L) noise tensors used instead of a draw (tests: a given noise sequence)
        self.graph = self.graph0 = self.graph2 = self.graphB = self.plan = self.fc = None
        self._static_grads, self._ptr_tables = [], None
        self.parts = False
        self.loss = None
        self._tables_read = None       # event: the last replay's H2D nodes have read the pinned scalar tables
        self.guard_tables = True       # False only for A/B timing of the guard itself (tools/ab_step.py)
        self.note = "eager"
        self.multi = reducer is not None
        self.wgrad_chunks = int(wgrad_chunks)
        if use_graph and self._forward_has_collective() and not hasattr(model, "set_collective_hook"):
            # collectives stay outside the captured graphs (module docstring); a model that cannot hand them over (ForwardCollectives)
            # is not captured
            use_graph = False
            self.note = "eager (the forward holds a collective and the model has no set_collective_hook)"
        if use_graph:
            self._capture(max(2, warmup))

    def _forward_has_collective(self):
        """True when the model's forward talks to other ranks (PrHubModel.forward_has_collective)."""
        f = getattr(self.model, "forward_has_collective", None)
        return bool(f()) if callable(f) else False

    # ------------------------------------------------------------------------------------------------ eager form
    def _draw_noise(self):
        """-> True when the (static-shape) launch sequence can serve this step's noise."""
        if self.noise is None:
            return True
        if self.step_prepare is None:
            if self.noise_feed is not None:
                self.noise.copy_(next(self.noise_feed).to(self.noise.device, non_blocking=True))
            else:
                self.noise.copy_(torch.rand(self.noise.shape, device=self.noise.device, generator=self.gen))
            return True
        t = self._noise_turn
        self._noise_turn = (t + 1) % len(self._noise_pins)
        if self._noise_events[t] is not None:
            self._noise_events[t].synchronize()        # the H2D copy that last read this pinned buffer has run
        pin = self._noise_pins[t]
        torch.rand(pin.shape, generator=self.host_gen, out=pin)
        self._noise_cpu = pin
        ok = bool(self.step_prepare(pin))
        self.noise.copy_(pin, non_blocking=True)
        ev = self._noise_events[t] or torch.cuda.Event()
        ev.record()
        self._noise_events[t] = ev
        return ok

    def eager_step(self):
        ok = self._draw_noise()
        # not ok: the model's own host path plans for exactly this noise (a CPU tensor tells it to)
        out = self.forward(self.model, *self.inputs, self.noise if ok else self._noise_cpu.clone())
        out[0].backward()
        if self.reducer is not None:
            self.reducer.finish()
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        return out[0]

    # ------------------------------------------------------------------------------------------------ capture
    def _snapshot(self):
        """Everything the warm-up steps of _capture() would otherwise leave changed: weights, module buffers (BatchNorm
        statistics, MoCo queue and pointer), optimizer moments + step counter, and the mask-noise generator."""
        params = [p for p in self.model.parameters()]
        snap = dict(params=[p.detach().clone() for p in params],
                    buffers=[(b, b.detach().clone()) for b in self.model.buffers()],
                    step=self.opt._step, gen=self.gen.get_state(), host_gen=self.host_gen.get_state(), moments={})
        for p in params:
            st = self.opt.state.get(p, {})
            if "exp_avg" in st:
                snap["moments"][p] = (st["exp_avg"].clone(), st["exp_avg_sq"].clone())
        return snap

    @torch.no_grad()
    def _restore(self, snap):
        """In place: the captured graphs keep pointing at these tensors."""
        for p, v in zip(self.model.parameters(), snap["params"]):
            p.copy_(v)
        for b, v in snap["buffers"]:
            b.copy_(v)
        for p in self.model.parameters():
            st = self.opt.state.get(p, {})
            if "exp_avg" in st:
                if p in snap["moments"]:
                    st["exp_avg"].copy_(snap["moments"][p][0])
                    st["exp_avg_sq"].copy_(snap["moments"][p][1])
                else:                     # moments born during the warm-up: back to their initial zeros
                    st["exp_avg"].zero_()
                    st["exp_avg_sq"].zero_()
        self.opt._step = snap["step"]
        self.gen.set_state(snap["gen"])
        self.host_gen.set_state(snap["host_gen"])
        ops.refresh_lp_shadows(self.model.parameters())

    def _capture(self, warmup):
        # The warm-up runs real optimizer steps on whatever the static inputs hold (allocator / autograd warm-up, and the
        # optimizer's state tensors must exist before capture). Training must nevertheless start from the weights,
        # moments, buffers and noise stream the caller handed over -- a resumed checkpoint in particular -- so all of it is
        # snapshotted here and put back, in place, once the graphs exist.
        snap = self._snapshot()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):       # warm up on the capture stream so every AccumulateGrad node is born there
            for _ in range(warmup):
                self.eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        try:
            if self.step_prepare is not None:
                # the capture needs a step the fixed launch shape can serve: draw until one fits (the noise stream is put back)
                with torch.cuda.stream(side):
                    for _ in range(64):
                        if self._draw_noise():
                            break
                    else:
                        raise RuntimeError("step_prepare refused 64 noise draws in a row: nothing to capture")
                side.synchronize()
            seq = _GraphSeq(side)
            fc = ForwardCollectives(seq) if (self._forward_has_collective() and hasattr(self.model, "set_collective_hook")) else None
            ops.hold_deferred_grads(self.multi)      # N ranks: the grouped weight-gradient launches stay out of the graph
            # N ranks: the backward is cut (BackwardCut) -- masked modeling at the encoder / decoder boundary, the contrastive stage at
            # the loss -- and captured as two graphs, so that what is complete after the first -- the decoder's weight gradients, the
            # contrastive keys -- crosses xGMI on side streams while the second (the encoder's backward, ~3 ms) replays: the window
            # that hides the collective grows from the step's last ~2 ms to ~5 ms (unmeasured on more than one GPU)
            cut = BackwardCut() if (self.multi and self.backward_cut and hasattr(self.model, "set_backward_cut")) else None
            early_steps = ()
            import gc
            try:
                if cut is not None:
                    self.model.set_backward_cut(cut)
                if fc is not None:
                    self.model.set_collective_hook(fc)
                torch.cuda.synchronize()
                gc.collect()
                torch.cuda.empty_cache()
                with torch.cuda.stream(side):
                    seq.begin()
                    try:
                        out = self.forward(self.model, *self.inputs, self.noise)     # (a splitting gather ends / begins graphs in here)
                        out[0].backward()
                        if not self.multi:
                            self.opt.refresh(scalars=False)
                            self.opt.launch()
                        self.loss = out[0].detach()
                        del out
                        seq.end()
                    except BaseException:
                        seq.abort()
                        raise
                g1 = seq.graphs[0]
                if cut is not None and cut.src is not None:        # the forward used the cut: second half of the backward
                    with torch.cuda.stream(side):
                        early_steps = ops.build_deferred_plan(1)   # the decoder side's queued weight / bias gradients
                    gB = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gB, stream=side, pool=g1.pool(), capture_error_mode="thread_local"):
                        cut.resume()
                    self.graphB = gB
            finally:
                ops.hold_deferred_grads(False)
                if cut is not None:
                    self.model.set_backward_cut(None)
                if fc is not None:
                    self.model.set_collective_hook(None)
            self.fc = fc
            if not self.multi:
                # the per-step scalar tables (lr, weight decay, bias corrections) travel in a graph of their own, replayed in
                # FRONT of the step: the host may then prepare step N+1 as soon as step N has started (see step())
                g0 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g0, stream=side, capture_error_mode="thread_local"):
                    self.opt.upload_scalars()
                self.graph0 = g0
            self.graph, self.note = seq, "hip-graph"
            if self.multi:
                # chunked weight-gradient launches interleaved with their all-reduces (parallel.OverlappedPlan)
                with torch.cuda.stream(side):
                    self.plan = self.reducer.make_overlapped_plan(self.wgrad_chunks, early_steps=early_steps)
                # AdamW in parts, each launched as soon as its gradient buffer is reduced (the graph then only holds the
                # H2D copies of the per-step scalar tables); one launch after the last all-reduce otherwise
                self.parts = self.plan.streams is not None
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, stream=side, capture_error_mode="thread_local"):
                    self.opt.refresh()
                    if not self.parts:
                        self.opt.launch()
                if self.parts:
                    with torch.cuda.stream(side):
                        self.plan.attach_optimizer(self.opt, [p for p in self.model.parameters() if p.requires_grad])
                side.synchronize()
                self.graph2 = g2
                self.note = (("hip-graph (fwd + decoder bwd) + decoder weight gradients all-reduced under hip-graph (encoder bwd) + " if self.graphB is not None
                              else "hip-graph (fwd+bwd) + ") +
                             "%d weight-gradient chunks on two streams, each all-reduced (RCCL) while the next "
                             "computes, AdamW per reduced buffer" % self.wgrad_chunks) if self.parts else \
                            ("hip-graph (fwd+bwd) + %d weight-gradient chunks overlapped with RCCL all-reduce + hip-graph (AdamW)" % self.wgrad_chunks)
                if fc is not None and (fc.post or fc.pre or len(seq.graphs) > 1):
                    self.note += " [forward collectives outside the graphs: %d splitting key all-gather(s), %d overlapped with the backward, " \
                                 "%d buffer broadcast(s)]" % (len(seq.graphs) - 1, len(fc.post), len(fc.pre))
            # what an eager fall-back step must put back: the gradient tensors the captured graphs (and the reducer's plan) write and
            # the optimizer reads, and the optimizer's pointer tables
            self._static_grads = [(p, p.grad) for p in self.model.parameters() if p.grad is not None]
            T = self.opt._tabs
            self._ptr_tables = (T, T["n_grads"].copy(), T["n_lp"].copy()) if T is not None else None
        except Exception as e:               # keep training; say what happened
            import os
            if os.environ.get("EVP_RAISE_CAPTURE"):
                import traceback
                traceback.print_exc()
            self.graph = self.graph0 = self.graph2 = self.graphB = self.plan = self.fc = None
            self.note = "eager (graph capture failed: %r)" % (e,)
            self.opt.zero_grad(set_to_none=True)
            ops.flush_deferred_grads()
            dev = self.inputs[0].device
            seed = self.gen.initial_seed()
            self.gen = torch.Generator(device=dev).manual_seed(seed)      # a failed capture can poison the old generator
            snap["gen"] = self.gen.get_state()
        torch.cuda.synchronize()
        self._restore(snap)
        torch.cuda.synchronize()

    def resync_weights(self):
        """Call after the weights were changed behind the optimizer's back (load_state_dict, manual edits): refreshes the
        bf16 weight shadows in place so the captured graphs see the new values."""
        ops.refresh_lp_shadows(self.model.parameters())

    # ------------------------------------------------------------------------------------------------ stepping
    def step(self, *new_inputs):
        """One optimizer step. With arguments, the batch is first copied into the static buffers (same shapes).
        Returns the loss as a 0-dim device tensor (the graph's static output: read it before the next step)."""
        for dst, src in zip(self.inputs, new_inputs):
            if src is not dst:
                dst.copy_(src, non_blocking=True)
        if self.graph is None:
            return self.eager_step()
        if not self._vote(self._draw_noise()):
            return self._eager_fallback()
        # The captured H2D nodes read the optimizer's pinned scalar tables when the REPLAY runs, not when it is queued: a host
        # that is a step ahead would hand step N the learning rate / bias corrections of step N+1. Wait until the replay that
        # read them last has passed that point. One GPU: the tables are read by graph0 at the START of a step, so the host
        # still queues step N+1 while step N runs; N GPUs: by graph2, behind forward + backward.
        if self._tables_read is not None and self.guard_tables:
            self._tables_read.synchronize()
        self.opt.stage_scalars()
        if self.graph0 is not None:
            self.graph0.replay()
            self._mark_tables_read()
        if self.plan is not None and self.parts:
            self.graph2.replay()             # this step's lr / bias-correction tables -> device, ahead of everything that updates
            self._mark_tables_read()
        fc = self.fc
        if fc is not None:
            for fn in fc.pre:
                fn()                         # (reference-faithful mode: rank 0's buffers -> every rank, before the forward)
        self.graph.replay()                  # forward (+ backward); a splitting key all-gather runs between its graphs
        post = []
        if fc is not None and fc.post:
            import torch.distributed as dist
            # keys of this step -> every rank, asynchronously on RCCL's stream (ordered after the forward graph): under the backward
            post = [(dist.all_gather_into_tensor(out, t, async_op=True), out, then) for t, out, then in fc.post]
        if self.plan is not None:
            if self.graphB is not None:
                self.plan.run_early()        # decoder weight gradients + their all-reduce (+ update) on side streams ...
                self.graphB.replay()         # ... under the encoder's backward
            for w, out, then in post:
                w.wait()                     # stream wait
                then(out)                    # the enqueue of the gathered keys (one small launch)
            if self.parts:
                self.plan.run()              # weight gradients, all-reduces and the update, part by part
            else:
                self.plan.run()
                self.graph2.replay()
                self._mark_tables_read()
        else:
            for w, out, then in post:
                w.wait()
                then(out)
        return self.loss

    def _vote(self, ok):
        """All ranks replay, or all ranks fall back: MIN over ranks of this rank's verdict (host-side, gloo: no device sync; the
        previous replay is still running on the GPU while the ranks agree on this one)."""
        if self._vote_group is None:
            return ok
        import torch.distributed as dist
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self._vote_group)
        return bool(t.item())

    def eager_step_with(self, *inputs):
        """One optimizer step on inputs whose shapes the captured graph cannot serve (the short last batch of an epoch), leaving
        the graph usable for the batches after it. Single-process executors only."""
        noise = None
        if self.noise is not None:
            B = inputs[0].shape[0]
            noise = torch.rand(B, *self.noise.shape[1:], device=self.noise.device, generator=self.gen)
        if self.graph is None:
            # nothing captured (capture failed or was refused): a plain eager step on the given inputs, data-parallel included --
            # every rank meets the short batch at the same iteration, so the reducer's collectives match
            out = self.forward(self.model, *inputs, noise)
            out[0].backward()
            if self.reducer is not None:
                self.reducer.finish()
            self.opt.step()
            self.opt.zero_grad(set_to_none=True)
            return out[0].detach()
        return self._eager_fallback(list(inputs), noise)

    def _eager_fallback(self, inputs=None, noise=None):
        """One step outside the captured graph (the step's launch geometry does not fit it), leaving the graph usable: the
        eager backward must not accumulate into the captured gradient tensors, and the optimizer's eager refresh() rewrites
        the pinned pointer tables the graph's H2D nodes re-read -- both are put back."""
        self.eager_fallbacks += 1
        torch.cuda.current_stream().synchronize()
        self.opt.zero_grad(set_to_none=True)
        if inputs is None:
            out = self.forward(self.model, *self.inputs, self._noise_cpu.clone())
        else:
            out = self.forward(self.model, *inputs, noise)
        out[0].backward()
        if self.reducer is not None:
            # data-parallel form: every rank is here (the verdict was collective, or every rank met the same short batch), so the
            # reducer's all-reduces match; the mean rides on the optimizer's grad_scale as in the captured step
            self.reducer.finish()
        self.opt.step()
        self.opt.zero_grad(set_to_none=True)
        torch.cuda.current_stream().synchronize()
        for p, g in self._static_grads:
            p.grad = g
        if self._ptr_tables is not None and self.opt._tabs is self._ptr_tables[0]:
            T, ng, nl = self._ptr_tables
            T["n_grads"][:] = ng
            T["n_lp"][:] = nl
        else:
            raise RuntimeError("the optimizer rebuilt its tables during an eager fall-back step; re-capture the executor")
        self.loss.copy_(out[0].detach())
        return self.loss

    def _mark_tables_read(self):
        if self._tables_read is None:
            self._tables_read = torch.cuda.Event()
        self._tables_read.record()
