#!/usr/bin/env python3
"""K1 time against the number of clips (100 k events each, trusted sorted input): 8 clips = 64 workgroups on 64 CUs ... 128 clips = 1024
workgroups = four rounds. Per-CU-bound work would take the same time up to 32 clips (one round); work bound by a shared resource scales."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eventpretrain_amd.dataset.dataset_utils.events_to_voxel_grid import voxel_grid_batch
from eventpretrain_amd.testing import synthetic_events
n = 100_000
clips = [synthetic_events(i, n) for i in range(16)]
for B in (8, 16, 24, 32, 48, 64, 96, 128):
    ev = torch.from_numpy(np.concatenate([clips[i % 16] for i in range(B)])).cuda()
    off = torch.arange(0, (B + 1) * n, n, dtype=torch.int64).cuda()
    out = torch.empty(B, 5, 224, 224, device="cuda")
    for _ in range(3):
        voxel_grid_batch(ev, off, 5, (224, 224), out=out, assume_sorted="trust")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        voxel_grid_batch(ev, off, 5, (224, 224), out=out, assume_sorted="trust")
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e3
    print(f"{B:4d} clips = {B * 8:5d} workgroups: {t:7.1f} us  {B * (n * 32 + 5 * 224 * 224 * 4) / t * 1e-3:7.1f} GB/s algorithmic", flush=True)
