#!/usr/bin/env python3
"""Probe (not part of the product): which Python lines launch ATen kernels (fill / add / copy ...) inside one eager bf16
ViT-Base step -- everything arithmetic on the path should be an evp_* kernel."""
import os
import sys
import traceback
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

from eventpretrain_amd import ops  # noqa: E402
from eventpretrain_amd.model.pretrain import pr_hub_model as hub  # noqa: E402
from eventpretrain_amd.optim import FusedAdamW  # noqa: E402
from eventpretrain_amd.testing import make_args  # noqa: E402
from eventpretrain_amd.utils import lr_decay as lrd  # noqa: E402

SKIP = ("aten.empty", "aten.view", "aten._unsafe_view", "aten.detach", "aten.as_strided", "aten.slice", "aten.select", "aten.alias",
        "aten.empty_like", "aten.empty_strided", "aten.reshape", "aten.t.", "aten.transpose", "aten.expand", "aten.unsqueeze",
        "aten.squeeze", "aten.permute", "aten.lift_fresh", "aten.is_pinned", "aten._local_scalar_dense", "aten.new_empty")


class Spy(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.hits = Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not name.startswith(SKIP):
            dev = [a.device.type for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor)]
            if "cuda" in dev or "device" in (kwargs or {}):
                frames = [f for f in traceback.extract_stack() if "eventpretrain_amd" in f.filename or f.filename.endswith("aten_probe.py")]
                where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in frames[-3:][::-1])
                self.hits[(name, where)] += 1
        return func(*args, **(kwargs or {}))


def main():
    ops.set_compute_dtype(torch.bfloat16)
    B = 8
    a = make_args(model_size="base", pr_phase="rec", device="cuda", batch_size=B)
    m = hub.pretrain_hub_model_base_patch16(a, emb_frames_dim=512, queue_length=1024, T=0.07).cuda().train()
    opt = FusedAdamW(lrd.param_groups_lrd(a, m, 0.05, layer_decay=1), lr=1e-4, betas=(0.9, 0.95))
    x = torch.randn(B, 5, 224, 224, device="cuda") * 0.5
    y = torch.randn(B, 1, 224, 224, device="cuda")
    noise = torch.rand(B, 196, device="cuda")
    for _ in range(2):
        m(x, y, is_rec=True, noise=noise)[0].backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
    with Spy() as spy:
        m(x, y, is_rec=True, noise=noise)[0].backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    for (name, where), n in sorted(spy.hits.items(), key=lambda kv: -kv[1]):
        print(f"{n:4d}  {name:40s} {where}")


if __name__ == "__main__":
    main()
