#!/usr/bin/env python3
"""cProfile of the eager Swin-T step: where the host time goes (the step cannot be graph-captured: the window plan
needs the mask on the host)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eventpretrain_amd import ops
from eventpretrain_amd.model.pretrain import pr_hub_model as hub
from eventpretrain_amd.optim import FusedAdamW
from eventpretrain_amd.testing import make_args
from eventpretrain_amd.utils import lr_decay as lrd
ops.set_compute_dtype(torch.bfloat16)
B = 64
a = make_args(model_size="tiny", pr_phase="rec", backbone_type="swin", device="cuda")
m = hub.pretrain_hub_model_swin_tiny_patch16(a, emb_frames_dim=512, queue_length=1024, T=0.07).cuda().train()
opt = FusedAdamW(lrd.param_groups_lrd(a, m, 0.05, layer_decay=1), lr=1e-4, betas=(0.9, 0.95))
x = torch.randn(B, 5, 224, 224, device="cuda") * 0.5
y = torch.randn(B, 1, 224, 224, device="cuda")
def step():
    out = m(x, y, is_rec=True)
    out[0].backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
print("host ms/step (launch side)", t_host / 5 * 1e3, "wall ms/step", (time.perf_counter() - t0) / 5 * 1e3)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30); print(s.getvalue()[:6500])
