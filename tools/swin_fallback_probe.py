#!/usr/bin/env python3
"""Debug probe: graphed Swin step with a tight static plan (some steps fall back to eager); prints per step whether it was a
replay or a fall-back, the loss, and whether parameters / moments are finite."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch  # noqa: E402

from eventpretrain_amd import ops  # noqa: E402
from eventpretrain_amd.engine import GraphedStep  # noqa: E402
from eventpretrain_amd.model.pretrain import pr_hub_model as hub  # noqa: E402
from eventpretrain_amd.optim import FusedAdamW  # noqa: E402
from eventpretrain_amd.testing import det_fill_module_, make_args  # noqa: E402
from eventpretrain_amd.utils import lr_decay as lrd  # noqa: E402


def patch_fallback(ex, mode):
    """bisect what part of an eager fall-back step disturbs the following replay"""
    if mode == "full":
        return
    def fb():
        ex.eager_fallbacks += 1
        torch.cuda.synchronize()
        if mode == "noop":
            return ex.loss
        ex.opt.zero_grad(set_to_none=True)
        if mode == "fwdonly":
            with torch.no_grad():
                out = ex.forward(ex.model, *ex.inputs, ex._noise_cpu.clone())
        else:
            out = ex.forward(ex.model, *ex.inputs, ex._noise_cpu.clone())
            out[0].backward()
            if mode == "flush":
                ops.flush_deferred_grads()
        ex.opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        for p, g in ex._static_grads:
            p.grad = g
        ex.loss.copy_(out[0].detach())
        return ex.loss
    ex._eager_fallback = fb


def main():
    slack = float(sys.argv[1]) if len(sys.argv) > 1 else 1.05
    mode = sys.argv[2] if len(sys.argv) > 2 else "full"
    ops.set_compute_dtype(torch.bfloat16 if os.environ.get("PROBE_BF16") else torch.float32)
    a = make_args(model_size="tiny", pr_phase="rec", backbone_type="swin", device="cuda")
    m = hub.pretrain_hub_model_swin_tiny_patch16(a, emb_frames_dim=512, queue_length=1024, T=0.07)
    det_fill_module_(m)
    m = m.cuda().train()
    B = 2
    x = torch.randn(B, 5, 224, 224, device="cuda") * 0.5
    y = torch.randn(B, 1, 224, 224, device="cuda")
    opt = FusedAdamW(lrd.param_groups_lrd(a, m, 0.05, layer_decay=1), lr=1e-3, betas=(0.9, 0.95))
    prepare = m.backbone.enable_static_plan("cuda", slack=slack)
    fwd = lambda mm, xx, yy, noise: mm(xx, yy, is_rec=True, noise=noise)
    ex = GraphedStep(m, opt, fwd, [x, y], noise_shape=(B, 49), warmup=2, step_prepare=prepare, host_generator=torch.Generator().manual_seed(77))
    print(ex.note, mode, flush=True)
    patch_fallback(ex, mode)
    if mode == "prefilter":
        # the executor's plan never sees a pattern that does not fit: a second plan object screens the noise first
        import numpy as np
        from eventpretrain_amd.model.backbone.swin import StaticPatternPlan
        from eventpretrain_amd.model.sub_module.swin_block import PlanOverflow
        screen = StaticPatternPlan(m.backbone, "cuda", 24, slack)
        real = ex.step_prepare
        def filtered(noise_cpu):
            while True:
                n0 = noise_cpu[0].numpy()
                vis = np.zeros(49, dtype=bool)
                vis[np.argsort(n0, kind="stable")[:24]] = True
                try:
                    screen.load(vis)
                    break
                except PlanOverflow:
                    print("   (screened out an overflowing pattern, redrawing)")
                    torch.rand(noise_cpu.shape, generator=ex.host_gen, out=noise_cpu)
            return real(noise_cpu)
        ex.step_prepare = filtered
    for i in range(10):
        fb = ex.eager_fallbacks
        loss = ex.step().item()
        torch.cuda.synchronize()
        bad_p = [n for n, p in m.named_parameters() if not torch.isfinite(p).all()]
        bad_m = sum(1 for st in opt.state.values() if "exp_avg" in st and not (torch.isfinite(st["exp_avg"]).all() and torch.isfinite(st["exp_avg_sq"]).all()))
        T = opt._tabs
        dev_ptrs = T["grads"].cpu().numpy()
        same = int((dev_ptrs == ex._ptr_tables[1]).sum())
        names = {id(p_): n for n, p_ in m.named_parameters()}
        bad_g = [names[id(p_)] for p_, g in ex._static_grads if not torch.isfinite(g).all()]
        if i == 0:
            have = {id(p_) for p_, _ in ex._static_grads}
            print("   params without a captured grad tensor:", [n for n, p_ in m.named_parameters() if id(p_) not in have][:8], len(have))
        print(f"   device grad-pointer table equals the captured one in {same}/{len(dev_ptrs)} entries; non-finite static grads {len(bad_g)} {bad_g[:3]}")
        print(f"step {i}: {'FALLBACK' if ex.eager_fallbacks > fb else 'replay  '} loss {loss:.6f} non-finite params {len(bad_p)} {bad_p[:12]} moments {bad_m} opt._step {opt._step}", flush=True)


if __name__ == "__main__":
    main()
