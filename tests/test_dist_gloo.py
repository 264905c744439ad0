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
        resq.put((rank, out))
    finally:
        dist.destroy_process_group()


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
