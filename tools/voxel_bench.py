#!/usr/bin/env python3
"""K1 micro-benchmark: 64 clips x 100k events -> 5x224x224, algorithmic bytes 4.2 MB/clip."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eventpretrain_amd.dataset.dataset_utils.events_to_voxel_grid import voxel_grid_batch
from eventpretrain_amd.testing import synthetic_events
B, n = 64, 100_000
base = synthetic_events(0, n)
rng = np.random.default_rng(1)
evs = []
for i in range(B):
    e = base.copy(); e[:, 0] = (e[:, 0] + rng.integers(0, 224)) % 224; e[:, 1] = (e[:, 1] + rng.integers(0, 224)) % 224; evs.append(e)
ev = torch.from_numpy(np.concatenate(evs)).cuda(); off = torch.arange(0, (B + 1) * n, n, dtype=torch.int64).cuda()
out = torch.empty(B, 5, 224, 224, device="cuda")
ref = voxel_grid_batch(ev, off, 5, (224, 224), algo=1).clone()
bytes_ = B * (n * 32 + 5 * 224 * 224 * 4)
for label, kw in [("single-pass auto", {}), ("single-pass 112", dict(tile_rows=112)), ("single-pass 75", dict(tile_rows=75)), ("single-pass 56", dict(tile_rows=56)),
                  ("single-pass trust", dict(assume_sorted="trust")), ("f64 cells auto", dict(algo=3)), ("f64 cells 56", dict(algo=3, tile_rows=56)),
                  ("packed auto", dict(algo=2)), ("packed tile 56", dict(algo=2, tile_rows=56)),
                  ("global atomics", dict(algo=1)), ("single-pass unsorted", dict(assume_sorted=False))]:
    for _ in range(3): voxel_grid_batch(ev, off, 5, (224, 224), out=out, **kw)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): voxel_grid_batch(ev, off, 5, (224, 224), out=out, **kw)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    print(f"{label:20s} {t*1e6:8.1f} us  {bytes_/t/1e9:8.1f} GB/s  max|diff vs atomics| {(out-ref).abs().max().item():.2e}")
