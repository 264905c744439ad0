#!/usr/bin/env python3
"""Probe (not part of the product): does a captured step graph keep working while OTHER models are built, warmed up and
captured in the same process?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import ab_step  # noqa: E402
from eventpretrain_amd import ops  # noqa: E402

ops.set_compute_dtype(torch.bfloat16)
exA = ab_step.build(64, {})
print("A:", [round(exA.step().item(), 4) for _ in range(6)], flush=True)
exB = ab_step.build(64, {"set_grad_side": False})
print("A after building B:", [round(exA.step().item(), 4) for _ in range(6)], flush=True)
print("B:", [round(exB.step().item(), 4) for _ in range(6)], flush=True)
print("A:", [round(exA.step().item(), 4) for _ in range(6)], flush=True)
exC = ab_step.build(64, {"set_wgrad_xcd_order": False})
print("A after building C:", [round(exA.step().item(), 4) for _ in range(6)], flush=True)
print("B after building C:", [round(exB.step().item(), 4) for _ in range(6)], flush=True)
print("C:", [round(exC.step().item(), 4) for _ in range(6)], flush=True)
for p in exA.model.parameters():
    if not torch.isfinite(p).all():
        print("non-finite parameter in A:", [n for n, q in exA.model.named_parameters() if q is p])
        break
