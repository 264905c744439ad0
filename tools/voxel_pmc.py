#!/usr/bin/env python3
"""Runs K1 (default algorithm) 5 times on bench.py's synthetic batch -- the target of the rocprofv3 --pmc passes that
measure its HBM-side traffic (FETCH_SIZE / WRITE_SIZE; MI355X_MICROARCH.md "HBM")."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eventpretrain_amd.dataset.dataset_utils.events_to_voxel_grid import voxel_grid_batch
from eventpretrain_amd.testing import synthetic_events
B, n = 64, 100_000
ev = torch.from_numpy(np.concatenate([synthetic_events(i, n) for i in range(B)])).cuda()
off = torch.arange(0, (B + 1) * n, n, dtype=torch.int64).cuda()
out = torch.empty(B, 5, 224, 224, device="cuda")
for _ in range(5):
    voxel_grid_batch(ev, off, 5, (224, 224), out=out)
torch.cuda.synchronize()
print("done")
