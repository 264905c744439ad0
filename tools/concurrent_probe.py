#!/usr/bin/env python3
"""Probe (not part of the product): how much does the chip gain when TWO independent step chains run concurrently?
Two ViT-Base models at batch B/2, each captured in its own HIP graph, replayed on two streams at the same time, against one
model at batch B. If the pair finishes a B-sample step markedly faster, interleaving two micro-batches inside one step is
worth building (one chain's store-bound epilogues / partly filled launch tails overlap the other chain's MFMA phases).
usage: concurrent_probe.py [--batch 64] [--steps 20]"""
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


def build(B, seed):
    a = make_args(model_size="base", pr_phase="rec", device="cuda", batch_size=B)
    torch.manual_seed(seed)
    m = hub.pretrain_hub_model_base_patch16(a, emb_frames_dim=512, queue_length=1024, T=0.07).cuda().train()
    opt = FusedAdamW(lrd.param_groups_lrd(a, m, 0.05, layer_decay=1), lr=1e-4, betas=(0.9, 0.95))
    x = torch.randn(B, 5, 224, 224, device="cuda") * 0.5
    y = torch.randn(B, 1, 224, 224, device="cuda")
    fwd = lambda mm, xx, yy, noise: mm(xx, yy, is_rec=True, noise=noise)
    ex = GraphedStep(m, opt, fwd, [x, y], noise_shape=(B, 196), generator=torch.Generator(device="cuda").manual_seed(seed), warmup=3)
    assert ex.note.startswith("hip-graph"), ex.note
    return ex


def timeit(fn, steps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=20)
    ns = ap.parse_args()
    ops.set_compute_dtype(torch.bfloat16)
    B = ns.batch
    one = build(B, 1)
    t_one = timeit(one.step, ns.steps)
    print(f"one chain,  B={B}: {t_one:.2f} ms/step = {B / t_one * 1e3:.0f} samples/s", flush=True)
    del one
    torch.cuda.empty_cache()
    a, b = build(B // 2, 2), build(B // 2, 3)
    t_half = timeit(a.step, ns.steps)
    print(f"one chain,  B={B // 2}: {t_half:.2f} ms/step = {B / 2 / t_half * 1e3:.0f} samples/s", flush=True)
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()

    def both():
        cur = torch.cuda.current_stream()
        s0.wait_stream(cur)
        s1.wait_stream(cur)
        with torch.cuda.stream(s0):
            a.step()
        with torch.cuda.stream(s1):
            b.step()
        cur.wait_stream(s0)
        cur.wait_stream(s1)

    t_two = timeit(both, ns.steps)
    print(f"two chains, B={B // 2} each, concurrent: {t_two:.2f} ms per pair = {B / t_two * 1e3:.0f} samples/s "
          f"({t_one / t_two:.2f}x one chain at B={B}; serial pair would be {2 * t_half:.2f} ms)", flush=True)


if __name__ == "__main__":
    main()
