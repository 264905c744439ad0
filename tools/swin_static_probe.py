#!/usr/bin/env python3
"""Debug probe: eager Swin-T steps (no graph) with the static plan at several slacks against the pattern-sized plan, over random
patterns: loss difference and non-finite gradients."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eventpretrain_amd import ops  # noqa: E402
from eventpretrain_amd.model.pretrain import pr_hub_model as hub  # noqa: E402
from eventpretrain_amd.testing import det_fill_module_, make_args  # noqa: E402


def main():
    ops.set_compute_dtype(torch.float32)
    a = make_args(model_size="tiny", pr_phase="rec", backbone_type="swin", device="cuda")
    m = hub.pretrain_hub_model_swin_tiny_patch16(a, emb_frames_dim=512, queue_length=1024, T=0.07)
    det_fill_module_(m)
    m = m.cuda().train()
    B = 2
    x = torch.randn(B, 5, 224, 224, device="cuda") * 0.5
    y = torch.randn(B, 1, 224, 224, device="cuda")
    gen = torch.Generator().manual_seed(77)
    noises = [torch.rand(B, 49, generator=gen) for _ in range(10)]
    ref = []
    for nz in noises:
        m.zero_grad(set_to_none=True)
        out = m(x, y, is_rec=True, noise=nz)
        out[0].backward()
        ref.append((out[0].item(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    for slack in (1.05, 1.1, 1.25):
        prepare = m.backbone.enable_static_plan("cuda", slack=slack)
        for i, nz in enumerate(noises):
            ok = prepare(nz)
            if not ok:
                print(f"slack {slack} pattern {i}: overflow")
                continue
            m.zero_grad(set_to_none=True)
            out = m(x, y, is_rec=True, noise=nz.cuda())
            out[0].backward()
            torch.cuda.synchronize()
            bad = [n for n, p in m.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
            worst = max(((p.grad - ref[i][1][n]).abs().max().item() / (ref[i][1][n].abs().max().item() + 1e-12), n)
                        for n, p in m.named_parameters() if p.grad is not None and n not in bad)
            sp = m.backbone._static_plan
            print(f"slack {slack} pattern {i}: loss {out[0].item():.6f} (pattern plan {ref[i][0]:.6f}) non-finite grads {len(bad)} {bad[:2]} worst rel grad diff {worst[0]:.2e} {worst[1]}",
                  flush=True)
        m.backbone._static = None


if __name__ == "__main__":
    main()
