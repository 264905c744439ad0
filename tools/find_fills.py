#!/usr/bin/env python3
"""Where do the ATen fill / copy / add launches of one step come from? One eager ViT-Base step under torch.profiler with Python
stacks; prints, per (op, innermost repo frame), how many device launches it makes per step."""
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

from eventpretrain_amd import ops  # noqa: E402
from eventpretrain_amd.model.pretrain import pr_hub_model as hub  # noqa: E402
from eventpretrain_amd.optim import FusedAdamW  # noqa: E402
from eventpretrain_amd.testing import make_args  # noqa: E402
from eventpretrain_amd.utils import lr_decay as lrd  # noqa: E402

ops.set_compute_dtype(torch.bfloat16)
B = 64
a = make_args(model_size="base", pr_phase="rec", device="cuda", batch_size=B)
torch.manual_seed(1)
m = hub.pretrain_hub_model_base_patch16(a, emb_frames_dim=512, queue_length=1024, T=0.07).cuda().train()
opt = FusedAdamW(lrd.param_groups_lrd(a, m, 0.05, layer_decay=1), lr=1e-4, betas=(0.9, 0.95))
x = torch.randn(B, 5, 224, 224, device="cuda") * 0.5
y = torch.randn(B, 1, 224, 224, device="cuda")
noise = torch.rand(B, 196, device="cuda")


def step():
    out = m(x, y, is_rec=True, noise=noise)
    out[0].backward()
    opt.step()
    opt.zero_grad(set_to_none=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
want = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::add", "aten::add_", "aten::zeros", "aten::uniform_", "aten::mul", "aten::div", "aten::sum",
        "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::cat", "aten::index", "aten::select", "aten::empty_like")
count = collections.Counter()
for ev in prof.events():
    if ev.name not in want or ev.device_time_total <= 0:
        continue
    fr = [s for s in ev.stack if "/eventpretrain_amd/" in s or "/tools/" in s]
    where = fr[0] if fr else (ev.stack[0] if ev.stack else "?")
    count[(ev.name, where.split("/root/repo/")[-1])] += 1
for (name, where), n in sorted(count.items(), key=lambda kv: -kv[1]):
    print(f"{n:4d}  {name:16s} {where}")
