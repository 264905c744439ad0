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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--lr", type=float, default=1e-3)
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
