"""N>1 control flow on CPU with the gloo backend, world_size 2: the bucketed gradient reducer, the scalar/meter
reductions of the trainer, the key all-gather and the rank-offset labels of the in-batch InfoNCE (checked with the
oracle's loss, since the HIP kernels need a GPU)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, resq):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from eventpretrain_amd.model.pretrain.pr_hub_model import concat_all_gather
        from eventpretrain_amd.parallel import BucketedGradReducer
        from eventpretrain_amd.utils import misc
        from oracle import model_oracle as mo
        out = {}
        # --- bucketed reducer: several buckets, one parameter without a gradient
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.zeros(s)) for s in [(300, 70), (17,), (1, 1, 64), (4000,), (33, 5)]]
        red = BucketedGradReducer(params)
        loss = sum(((rank + 1) * (i + 1)) * p.sum() for i, p in enumerate(params[:-1]))   # last param unused
        loss.backward()
        red.finish()
        out["grads"] = [None if p.grad is None else p.grad.flatten()[:3].tolist() for p in params]
        # second step: a different set of parameters has gradients -> the small bucket is rebuilt
        for p in params:
            p.grad = None
        sum(p.sum() for p in params).backward()
        red.finish()
        out["grads2"] = [p.grad.flatten()[0].item() for p in params]
        # --- trainer-side reductions
        out["mean"] = misc.all_reduce_mean(float(rank + 1))
        m = misc.SmoothedValue()
        m.update(float(rank), n=1)
        m.update(10.0 * (rank + 1), n=2)
        m.synchronize_between_processes()
        out["meter"] = (m.count, m.total)
        assert misc.get_world_size() == 2 and misc.get_rank() == rank
        # --- key all-gather + rank-offset labels of the in-batch InfoNCE (pr_hub_model.py:170-188)
        g = torch.Generator().manual_seed(5)
        qa, ka = torch.randn(4, 3, 8, generator=g), torch.randn(4, 3, 8, generator=g)     # global batch 4 = 2 ranks x 2
        q, k = qa[2 * rank:2 * rank + 2], ka[2 * rank:2 * rank + 2]
        k_all = concat_all_gather(k)
        assert torch.equal(k_all, ka)
        out["nce"] = mo.info_nce_inbatch(q, k_all, 0.07, rank=rank).item()
        full = mo.info_nce_inbatch(qa, ka, 0.07, rank=0).item()
        out["nce_full"] = full
        out.update(_queue_policies(rank))
        out.update(_overlapped_plan_streamless(rank))
        out.update(_epoch_with_reducer(rank))
        resq.put((rank, out))
    finally:
        dist.destroy_process_group()


def _queue_policies(rank):
    """default "all_gather" = every rank enqueues the keys of all ranks (identical queues, no broadcast);
    opt-in "rank0_broadcast" = the reference's DDP buffer broadcast (rank 0's queue / pointer / BN statistics overwrite the
    others at every forward). The enqueue kernel needs a GPU, so its CPU restatement (oracle) stands in for it here."""
    from eventpretrain_amd import ops
    from eventpretrain_amd.model.pretrain import pr_hub_model as hub
    from eventpretrain_amd.testing import make_args
    from oracle import model_oracle as mo

    def cpu_enqueue(queue, keys, queue_ptr):
        q, p = mo.enqueue(queue, int(queue_ptr), keys)
        queue.copy_(q)
        queue_ptr[0] = p

    real = ops.enqueue_keys_dev
    ops.enqueue_keys_dev = cpu_enqueue
    out = {}
    try:
        for pol in ("all_gather", "rank0_broadcast", "local"):
            a = make_args(model_size="tiny", pr_phase="con", use_queue=True, mask_ratio=0.0, distributed=True)
            a.queue_policy = pol
            torch.manual_seed(7)                                   # same initial queue on both ranks
            m = hub.pretrain_hub_model_tiny_patch16_64(a, emb_frames_dim=512, queue_length=8, T=0.07)
            assert m.queue_policy() == pol
            # the step executor keeps collectives out of captured graphs: it must know which forwards hold one
            assert m.forward_has_collective() == (pol != "local")
            q0 = m.queue.clone()
            keys = torch.full((2, 16, 192), float(rank + 1)) + torch.arange(2).view(2, 1, 1)      # rank r, sample i -> r + 1 + i
            if pol == "rank0_broadcast":
                m.queue.add_(float(rank))                          # ranks have drifted apart ...
                m._sync_buffers_from_rank0()                       # ... the per-forward broadcast pulls them back to rank 0's
                assert torch.equal(m.queue, q0), "buffers not taken from rank 0"
            m._dequeue_and_enqueue(keys)
            out[pol] = (int(m.queue_ptr), m.queue[0, 0, :].tolist())
        # a per-step launch-geometry hook (Swin window plan) together with a reducer: the replay-or-fall-back decision is collective
        # (engine.GraphedStep._vote, a host-side MIN on a gloo group) -- one rank whose plan does not fit takes every rank to the eager
        # data-parallel step, so the all-reduces stay matched (the CUDA form of the whole control flow: tests/dp_cuda_worker.py "swin")
        from types import SimpleNamespace
        from eventpretrain_amd.engine import GraphedStep
        stub = SimpleNamespace(_vote_group=dist.group.WORLD)
        out["vote"] = [GraphedStep._vote(stub, True), GraphedStep._vote(stub, rank != 1), GraphedStep._vote(stub, rank == 1),
                       GraphedStep._vote(SimpleNamespace(_vote_group=None), rank == 1)]
    finally:
        ops.enqueue_keys_dev = real
    return {"queue_" + k: v for k, v in out.items()}


def _overlapped_plan_streamless(rank):
    """parallel.OverlappedPlan.run() without side streams (the branch a CPU rehearsal or a stream-less caller takes): the
    small-gradient bucket goes first, then every chunk runs and its flat buffer is all-reduced, rounds in order."""
    from types import SimpleNamespace
    from eventpretrain_amd.parallel import OverlappedPlan
    log = []
    flats = [torch.zeros(8), torch.zeros(5)]

    def mk(i, rnd):
        def run():
            log.append(i)
            flats[i].add_(float((rank + 1) * (i + 1)))
        return SimpleNamespace(run=run, flats=[flats[i]], round=rnd)

    p1, p2 = torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(2, 2))
    p1.grad, p2.grad = torch.full((3,), float(rank + 1)), torch.full((2, 2), 10.0 * (rank + 1))
    bucket = torch.zeros(64 + 64)
    views = [bucket[0:3].view_as(p1), bucket[64:68].view_as(p2)]
    plan = OverlappedPlan([mk(0, 0), mk(1, 0)], [p1, p2], bucket, views, None, two_streams=False)
    assert plan.streams is None
    plan.run()
    out = {"plan_log": list(log), "plan_flats": [f[0].item() for f in flats], "plan_small": [p1.grad[0].item(), p2.grad[0, 0].item()],
           "plan_grad_is_view": p1.grad.data_ptr() == bucket.data_ptr()}
    # split backward (engine.BackwardCut): an EARLY step -- the decoder's weight gradients -- is launched and all-reduced by
    # run_early() before the rest of the backward; run() then does the remaining chunks and must not repeat it. A caller that
    # never calls run_early() still gets every step exactly once.
    early = torch.zeros(4)

    def early_run():
        log.append("e")
        early.add_(float(rank + 1) * 7.0)

    for call_early in (True, False):
        del log[:]
        for f in flats:
            f.zero_()
        early.zero_()
        q1 = torch.nn.Parameter(torch.zeros(3))
        q1.grad = torch.full((3,), float(rank + 1))
        b2 = torch.zeros(64)
        plan2 = OverlappedPlan([mk(0, 0), mk(1, 0)], [q1], b2, [b2[0:3].view_as(q1)], None, two_streams=False,
                               early_steps=[SimpleNamespace(run=early_run, flats=[early], round=0)])
        if call_early:
            plan2.run_early()
            log.append("|")                 # ... where the encoder's backward graph replays
        plan2.run()
        out["plan_early_%d" % int(call_early)] = ("".join(str(x) for x in log), early[0].item(), [f[0].item() for f in flats], q1.grad[0].item())
    return out


def _epoch_with_reducer(rank):
    """The reference's epoch loop (pr_rec_one_epoch) with the scaler carrying a BucketedGradReducer in DDP's place: ranks start
    from different seeds (main_pretrain.py:174) -> the reducer's constructor broadcasts rank 0's weights; each rank sees its own
    batches; after the epoch the weights are identical on both ranks and equal a single process stepping on the MEAN gradient;
    the clip coefficient uses the mean gradient's norm."""
    from eventpretrain_amd.parallel import BucketedGradReducer
    from eventpretrain_amd.testing import make_args
    from eventpretrain_amd.trainer.pretrain.pr_trainer import pr_rec_one_epoch
    from eventpretrain_amd.utils.misc import NativeScalerWithGradNormCount

    class Toy(torch.nn.Module):
        def __init__(self, seed):
            super().__init__()
            torch.manual_seed(seed)
            self.w = torch.nn.Linear(6, 4)
            self.register_buffer("stat", torch.full((2,), float(seed)))

        def forward(self, x, y, is_rec=True):
            return (((self.w(x) - y) ** 2).mean(),)

    class ScaledSGD(torch.optim.SGD):          # FusedAdamW's interface: grad_scale multiplies every gradient, grad_norm()
        grad_scale = 0.5

        def grad_norm(self):
            return torch.sqrt(sum((p.grad ** 2).sum() for g in self.param_groups for p in g["params"] if p.grad is not None))

        @torch.no_grad()
        def step(self):
            for g in self.param_groups:
                for p in g["params"]:
                    if p.grad is not None:
                        p.add_(p.grad, alpha=-g["lr"] * self.grad_scale)

    a = make_args(lr=0.1, min_lr=0.1, warmup_epochs=0, epochs=1, accum_iter=1, device="cpu", print_freq=100, backward=True)
    m = Toy(seed=11 + rank)
    red = BucketedGradReducer.for_module(m)
    assert float(m.stat[0]) == 11.0                                 # buffers came from rank 0
    opt = ScaledSGD(m.parameters(), lr=0.1)
    g = torch.Generator().manual_seed(100)
    data = [(torch.randn(2, 5, 6, generator=g), torch.randn(2, 5, 4, generator=g)) for _ in range(3)]     # [rank] per step
    loader = [dict(events_voxel_grid=xb[rank], sub_frame=yb[rank], image_name=["n"]) for xb, yb in data]
    norms = []
    scaler = NativeScalerWithGradNormCount(reducer=red)
    real_call = scaler.__class__.__call__

    def spy(self, *aa, **kk):
        n = real_call(self, *aa, **kk)
        norms.append(float(n))
        return n

    scaler.__class__.__call__ = spy
    try:
        pr_rec_one_epoch(a, m, loader, opt, 0, scaler)
    finally:
        scaler.__class__.__call__ = real_call
    # single-process reference on the mean gradient of both ranks' batches
    ref = Toy(seed=11)
    ropt = torch.optim.SGD(ref.parameters(), lr=0.1)
    rnorms = []
    for xb, yb in data:
        ropt.zero_grad()
        (0.5 * (ref(xb[0], yb[0])[0] + ref(xb[1], yb[1])[0])).backward()
        rnorms.append(float(torch.sqrt(sum((p.grad ** 2).sum() for p in ref.parameters()))))
        ropt.step()
    return {"epoch_w": m.w.weight.detach().flatten().tolist(), "epoch_ref_w": ref.w.weight.detach().flatten().tolist(),
            "epoch_norms": norms, "epoch_ref_norms": rnorms}


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=60) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in (0, 1):
        o = res[r]
        # SUM over ranks of (rank+1)*(i+1): (1+2)*(i+1); the 1/world mean is applied by FusedAdamW's grad_scale
        assert o["grads"][:4] == [[3.0 * (i + 1)] * 3 for i in range(4)][:4] or all(
            abs(o["grads"][i][0] - 3.0 * (i + 1)) < 1e-6 for i in range(4))
        assert o["grads"][4] is None
        assert all(abs(v - 2.0) < 1e-6 for v in o["grads2"])
        assert o["mean"] == pytest.approx(1.5)
        assert o["meter"] == (6, pytest.approx(0.0 + 1.0 + 20.0 + 40.0))
    # mean of the two ranks' losses equals the single-process loss over the global batch
    assert 0.5 * (res[0]["nce"] + res[1]["nce"]) == pytest.approx(res[0]["nce_full"], rel=1e-6)
    # queue policies: all_gather -> both ranks hold [r0s0, r0s1, r1s0, r1s1] = [1, 2, 2, 3] in slots 0..3, pointer 4
    for r in (0, 1):
        ptr_, row = res[r]["queue_all_gather"]
        assert ptr_ == 4 and row[:4] == [1.0, 2.0, 2.0, 3.0]
        ptr_, row = res[r]["queue_local"]
        assert ptr_ == 2 and row[:2] == [1.0 + r, 2.0 + r]
        ptr_, row = res[r]["queue_rank0_broadcast"]
        assert ptr_ == 2 and row[:2] == [1.0 + r, 2.0 + r]          # own keys on rank 0's (broadcast) queue
    assert res[0]["queue_all_gather"] == res[1]["queue_all_gather"]
    assert res[0]["queue_vote"] == [True, False, False, False] and res[1]["queue_vote"] == [True, False, False, True]
    # overlapped plan without streams: chunks in order, flats summed over ranks ((1+2)*(i+1)), small gradients through the bucket
    for r in (0, 1):
        assert res[r]["plan_log"] == [0, 1]
        assert res[r]["plan_flats"] == [3.0, 6.0]
        assert res[r]["plan_small"] == [3.0, 30.0] and res[r]["plan_grad_is_view"]
        # early step first (before the point where the second backward graph replays), every buffer reduced once over the 2 ranks
        assert tuple(res[r]["plan_early_1"]) == ("e|01", 21.0, [3.0, 6.0], 3.0), res[r]["plan_early_1"]
        assert tuple(res[r]["plan_early_0"]) == ("e01", 21.0, [3.0, 6.0], 3.0), res[r]["plan_early_0"]
    # epoch loop with the reducer: identical weights on both ranks, equal to the single-process mean-gradient run
    assert res[0]["epoch_w"] == pytest.approx(res[1]["epoch_w"], abs=0)
    assert res[0]["epoch_w"] == pytest.approx(res[0]["epoch_ref_w"], rel=1e-5, abs=1e-6)
    assert res[0]["epoch_norms"] == pytest.approx(res[0]["epoch_ref_norms"], rel=1e-5)


def _worker_n(rank, world, port, resq):
    """The rank-count-dependent parts at world sizes the scaling run uses (4, 8): reducer sums, mean scale, key gather order and
    rank-offset labels, meter reduction, the collective vote."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        from types import SimpleNamespace
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from eventpretrain_amd.engine import GraphedStep
        from eventpretrain_amd.model.pretrain.pr_hub_model import concat_all_gather
        from eventpretrain_amd.parallel import BucketedGradReducer, ensure_mean_grad_scale
        from eventpretrain_amd.utils import misc
        from oracle import model_oracle as mo
        out = {}
        torch.manual_seed(rank)                                     # ranks start apart: the reducer's constructor broadcasts rank 0's values
        params = [torch.nn.Parameter(torch.randn(s)) for s in [(129, 33), (7,), (2048,), (5, 5)]]
        red = BucketedGradReducer(params)
        out["p0"] = [p.detach().flatten()[0].item() for p in params]
        sum(((rank + 1) * (i + 1)) * p.sum() for i, p in enumerate(params)).backward()
        red.finish()
        out["grads"] = [p.grad.flatten()[-1].item() for p in params]
        opt = SimpleNamespace(grad_scale=1.0)
        out["scale"] = ensure_mean_grad_scale(opt, red)
        try:
            ensure_mean_grad_scale(SimpleNamespace(grad_scale=0.5 if world != 2 else 0.25), red)
            out["bad_scale_refused"] = False
        except ValueError:
            out["bad_scale_refused"] = True
        out["mean"] = misc.all_reduce_mean(float(rank))
        m = misc.SmoothedValue()
        m.update(float(rank), n=rank + 1)
        m.synchronize_between_processes()
        out["meter"] = (m.count, m.total)
        g = torch.Generator().manual_seed(5)
        per = 3
        qa, ka = torch.randn(world * per, 2, 8, generator=g), torch.randn(world * per, 2, 8, generator=g)
        q, k = qa[per * rank:per * (rank + 1)], ka[per * rank:per * (rank + 1)]
        k_all = concat_all_gather(k)
        out["gather_ok"] = bool(torch.equal(k_all, ka))
        out["nce"] = mo.info_nce_inbatch(q, k_all, 0.07, rank=rank).item()
        out["nce_full"] = mo.info_nce_inbatch(qa, ka, 0.07, rank=0).item()
        stub = SimpleNamespace(_vote_group=dist.group.WORLD)
        out["vote"] = [GraphedStep._vote(stub, True), GraphedStep._vote(stub, rank != world - 1)]
        resq.put((rank, out))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [4, 8])
def test_world_size_n_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_n, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tri = world * (world + 1) // 2
    for r in range(world):
        o = res[r]
        assert o["p0"] == res[0]["p0"]                                        # rank 0's initial values everywhere
        assert o["grads"] == pytest.approx([float(tri * (i + 1)) for i in range(4)])      # SUM over ranks; the mean rides on grad_scale
        assert o["scale"] == pytest.approx(1.0 / world) and o["bad_scale_refused"]
        assert o["mean"] == pytest.approx((world - 1) / 2)
        assert o["meter"] == (tri, pytest.approx(sum(float(k) * (k + 1) for k in range(world))))
        assert o["gather_ok"] and o["vote"] == [True, False]
    assert sum(res[r]["nce"] for r in range(world)) / world == pytest.approx(res[0]["nce_full"], rel=1e-6)
