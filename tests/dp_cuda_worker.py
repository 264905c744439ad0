"""Worker of tests/test_gpu_round4.py::test_two_rank_epoch_loop_on_the_device: one rank of a 2-rank data-parallel run of the DEFAULT
epoch loop (pr_rec_one_epoch -> auto_step_executor -> the multi-GPU GraphedStep: split backward, chunked weight gradients, AdamW in
parts) with its tensors on the GPU. Both ranks share the ONE card of the test box, so the process group is gloo (RCCL refuses two
ranks on one device); everything but the transport -- capture, plan order, reducer, grad_scale, the loop's bookkeeping -- is the
code an 8-GPU run executes. Started by torch.distributed.run; rank 0 writes the result file."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def build(dtype, lr):
    from eventpretrain_amd import ops
    from eventpretrain_amd.model.pretrain import pr_hub_model as hub
    from eventpretrain_amd.optim import FusedAdamW
    from eventpretrain_amd.testing import det_fill_module_, make_args
    from eventpretrain_amd.utils import lr_decay as lrd
    ops.set_compute_dtype(torch.bfloat16 if dtype == "bf16" else torch.float32)
    a = make_args(model_size="tiny", pr_phase="rec", patch_size=16, device="cuda", input_size=64)
    a.batch_size, a.epochs, a.warmup_epochs, a.accum_iter, a.lr, a.min_lr = 2, 4, 1, 1, lr, 1e-6
    m = hub.pretrain_hub_model_tiny_patch16_64(a, emb_frames_dim=512, queue_length=8, T=0.07)
    det_fill_module_(m)
    m = m.cuda().train()
    opt = FusedAdamW(lrd.param_groups_lrd(a, m, a.weight_decay, layer_decay=1), lr=a.lr, betas=(0.9, 0.95))
    return a, m, opt


def batch_of(rank, step):
    from eventpretrain_amd.testing import det_normalish, det_uniform
    x = det_normalish(f"dp.voxels.{rank}.{step}", (2, 5, 64, 64)) * 0.5
    y = det_normalish(f"dp.sub_frame.{rank}.{step}", (2, 1, 64, 64))
    noise = det_uniform(f"dp.noise.{rank}.{step}", (2, 16), 0.0, 1.0)
    return x, y, noise


def scenario_graph_vs_eager(kind, dtype, steps, rank, world):
    """The data-parallel CAPTURED step against the data-parallel EAGER step (the model's own collectives, reducer.finish()) from the
    same start, data and noise: contrastive stage with the gathered queue / in-batch InfoNCE / rank-0 buffer broadcast, and the Swin
    masked step with a launch shape so tight that some steps fall back (decided collectively)."""
    from eventpretrain_amd import ops
    from eventpretrain_amd.engine import GraphedStep
    from eventpretrain_amd.model.pretrain import pr_hub_model as hub
    from eventpretrain_amd.optim import FusedAdamW
    from eventpretrain_amd.parallel import BucketedGradReducer
    from eventpretrain_amd.testing import det_fill_module_, det_normalish, make_args
    from eventpretrain_amd.utils import lr_decay as lrd
    from helpers import checksums
    ops.set_compute_dtype(torch.bfloat16 if dtype == "bf16" else torch.float32)
    res = {}
    for mode in ("eager", "graph"):
        if kind == "swin":
            a = make_args(model_size="tiny", pr_phase="rec", backbone_type="swin", device="cuda", distributed=True)
            m = hub.pretrain_hub_model_swin_tiny_patch16(a, emb_frames_dim=512, queue_length=1024, T=0.07)
            B, xs, ys = 2, (2, 5, 224, 224), (2, 1, 224, 224)
        else:
            a = make_args(model_size="tiny", pr_phase="con", patch_size=16, device="cuda", input_size=64, distributed=True,
                          use_queue=(kind != "con_inbatch"), mask_ratio=0.0)
            if kind == "con_bcast":
                a.queue_policy = "rank0_broadcast"
            m = hub.pretrain_hub_model_tiny_patch16_64(a, emb_frames_dim=512, queue_length=8, T=0.07)
            B, xs, ys = 2, (2, 5, 64, 64), (2, 17, 512)
        det_fill_module_(m)
        m = m.cuda().train()
        opt = FusedAdamW(lrd.param_groups_lrd(a, m, 0.05, layer_decay=1), lr=1e-3, betas=(0.9, 0.95))
        red = BucketedGradReducer.for_module(m)
        x = (det_normalish(f"dp2.x.{kind}.{rank}", xs) * 0.5).cuda()
        y = det_normalish(f"dp2.y.{kind}.{rank}", ys).cuda()
        if kind == "swin":
            fwd = lambda mm, xx, yy, noise: mm(xx, yy, is_rec=True, noise=noise)
            prepare = m.backbone.enable_static_plan("cuda", slack=1.05) if mode == "graph" else (lambda n: False)
            ex = GraphedStep(m, opt, fwd, [x, y], noise_shape=(B, 49), use_graph=(mode == "graph"), warmup=2, reducer=red,
                             step_prepare=prepare, host_generator=torch.Generator().manual_seed(77 + rank))
        else:
            fwd = lambda mm, xx, yy, noise: mm(xx, yy)
            ex = GraphedStep(m, opt, fwd, [x, y], use_graph=(mode == "graph"), warmup=2, reducer=red)
        if mode == "graph":
            assert ex.note.startswith("hip-graph"), ex.note
        losses = []
        for s_ in range(steps):
            x2 = (det_normalish(f"dp2.x.{kind}.{rank}.{s_}", xs) * 0.5).cuda()
            losses.append(float(ex.step(x2, y).item()))
        torch.cuda.synchronize()
        sums = {k: float(checksums(p)[2]) for k, p in m.named_parameters()}
        bufs = {k: float(checksums(b.float())[2]) for k, b in m.named_buffers()}
        # identical across ranks: the weights always; the queue and its pointer too unless the policy is the reference's rank-0
        # broadcast (there the ranks' queues differ between a step's enqueue and the next forward's broadcast)
        same_q = kind != "con_bcast"
        mine = torch.tensor([sums[k] for k in sorted(sums)] + [bufs[k] for k in sorted(bufs) if "queue" in k and same_q], dtype=torch.float64)
        both = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        res[mode] = dict(losses=losses, wsums=sums, bufs=bufs, scale={k: float(p.detach().abs().sum()) for k, p in m.named_parameters()},
                         note=ex.note, fallbacks=ex.eager_fallbacks, ranks_equal=bool(torch.allclose(both[0], both[1], rtol=0, atol=0)),
                         n_graphs=(len(ex.graph.graphs) if ex.graph is not None else 0), split=ex.graphB is not None,
                         n_post=(len(ex.fc.post) if ex.fc is not None else 0), n_pre=(len(ex.fc.pre) if ex.fc is not None else 0))
        del ex, m, opt, red
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenario", default="rec")
    ap.add_argument("--out", required=True)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--lr", type=float, default=1e-3)
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if args.scenario != "rec":
        res = scenario_graph_vs_eager(args.scenario, args.dtype, args.steps, rank, world)
        if rank == 0:
            json.dump(res, open(args.out, "w"))
        dist.barrier()
        dist.destroy_process_group()
        return
    from eventpretrain_amd.parallel import BucketedGradReducer
    from eventpretrain_amd.trainer.pretrain import pr_trainer
    from eventpretrain_amd.utils.misc import NativeScalerWithGradNormCount
    from helpers import checksums
    a, m, opt = build(args.dtype, args.lr)
    assert opt.grad_scale == 1.0                      # the default: the executor has to set 1 / world itself
    scaler = NativeScalerWithGradNormCount(reducer=BucketedGradReducer.for_module(m))
    data = [batch_of(rank, s) for s in range(args.steps)]
    x0, y0 = data[0][0].cuda(), data[0][1].cuda()
    ex = pr_trainer.auto_step_executor(a, m, opt, scaler, (x0, y0), "reconstruct_loss")
    assert ex is not None and ex.multi and ex.graph is not None, getattr(ex, "note", None)
    assert abs(opt.grad_scale - 1.0 / world) < 1e-12
    ex.noise_feed = iter([d[2] for d in data])
    loader = [dict(events_voxel_grid=d[0], sub_frame=d[1], image_name=["i"] * 2) for d in data]
    stats = pr_trainer.pr_rec_one_epoch(a, m, loader, opt, 0, scaler)
    assert m._evp_auto_executor[1] is ex
    torch.cuda.synchronize()
    sums = {k: checksums(p)[2] for k, p in m.named_parameters()}
    mine = torch.tensor([sums[k] for k in sorted(sums)], dtype=torch.float64)
    both = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    if rank == 0:
        json.dump(dict(note=ex.note, stats=stats, wsums=sums, ranks_equal=bool(torch.equal(both[0], both[1])), parts=bool(ex.parts),
                       split=ex.graphB is not None, grad_scale=opt.grad_scale), open(args.out, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
