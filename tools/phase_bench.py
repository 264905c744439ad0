#!/usr/bin/env python3
"""Step time of the other pre-training configurations (BASELINE.json configs 3-5: contrastive stage, ConvViT, Swin-T) on
the same executor as bench.py -- secondary numbers, not the headline metric.
usage: phase_bench.py {con|convvit|swin|rec+con} [--batch 64] [--steps 10] [--eager]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eventpretrain_amd import ops  # noqa: E402
from eventpretrain_amd.engine import GraphedStep  # noqa: E402
from eventpretrain_amd.model.pretrain import pr_hub_model as hub  # noqa: E402
from eventpretrain_amd.optim import FusedAdamW  # noqa: E402
from eventpretrain_amd.testing import make_args  # noqa: E402
from eventpretrain_amd.utils import lr_decay as lrd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["con", "convvit", "swin"])
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--eager", action="store_true")
    ns = ap.parse_args()
    B = ns.batch
    ops.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(0)
    x = torch.randn(B, 5, 224, 224, device="cuda") * 0.5
    if ns.what == "con":
        a = make_args(model_size="base", pr_phase="con", use_queue=True, mask_ratio=0.0, device="cuda")
        m = hub.pretrain_hub_model_base_patch16(a, emb_frames_dim=512, queue_length=1024, T=0.07)
        supp = torch.randn(B, 197, 512, device="cuda")
        fwd, noise_shape = (lambda mm, xx, ss, noise: mm(xx, ss)), None
    elif ns.what == "convvit":
        a = make_args(model_size="base", pr_phase="rec", backbone_type="convvit", device="cuda")
        m = hub.pretrain_hub_model_base_patch16(a, emb_frames_dim=512, queue_length=1024, T=0.07)
        supp = torch.randn(B, 1, 224, 224, device="cuda")
        fwd, noise_shape = (lambda mm, xx, ss, noise: mm(xx, ss, is_rec=True, noise=noise)), (B, 196)
    else:
        a = make_args(model_size="tiny", pr_phase="rec", backbone_type="swin", device="cuda")
        m = hub.pretrain_hub_model_swin_tiny_patch16(a, emb_frames_dim=512, queue_length=1024, T=0.07)
        supp = torch.randn(B, 1, 224, 224, device="cuda")
        fwd, noise_shape = (lambda mm, xx, ss, noise: mm(xx, ss, is_rec=True, noise=noise)), (B, 49)
    m = m.cuda().train()
    opt = FusedAdamW(lrd.param_groups_lrd(a, m, 0.05, layer_decay=1), lr=1e-4, betas=(0.9, 0.95))
    # the Swin step reads the mask back to plan its windows: not capturable, runs eagerly
    use_graph = not ns.eager and ns.what != "swin"
    ex = GraphedStep(m, opt, fwd, [x, supp], noise_shape=noise_shape, generator=torch.Generator(device="cuda").manual_seed(1),
                     use_graph=use_graph, warmup=3)
    for _ in range(3):
        ex.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(ns.steps):
        loss = ex.step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / ns.steps
    print(f"{ns.what}: {dt * 1e3:.2f} ms/step, {B / dt:.0f} samples/s, B={B}, mode={ex.note}, loss={loss.item():.4f}, "
          f"params={sum(p.numel() for p in m.parameters()) / 1e6:.1f}M", flush=True)


if __name__ == "__main__":
    main()
