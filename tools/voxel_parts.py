#!/usr/bin/env python3
"""K1 taken apart (evp_voxel_set_debug): the whole kernel, without the LDS atomics, without the float64 time normalisation, and with the
rows only loaded -- for the f32-cell (algo 0) and f64-cell (algo 3) tiles, trusted sorted input (one bin launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eventpretrain_amd._lib import call
from eventpretrain_amd.dataset.dataset_utils.events_to_voxel_grid import voxel_grid_batch
from eventpretrain_amd.testing import synthetic_events
B, n = 64, 100_000
ev = torch.from_numpy(np.concatenate([synthetic_events(i, n) for i in range(B)])).cuda()
off = torch.arange(0, (B + 1) * n, n, dtype=torch.int64).cuda()
out = torch.empty(B, 5, 224, 224, device="cuda")
for algo, rows in [(0, 0), (0, 75), (3, 0)]:
    for dbg, what in [(0, "everything"), (1, "no LDS atomics"), (2, "no time normalisation"), (3, "rows only loaded")]:
        call("evp_voxel_set_debug", dbg)
        kw = dict(algo=algo, tile_rows=rows, assume_sorted="trust")
        for _ in range(3):
            voxel_grid_batch(ev, off, 5, (224, 224), out=out, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            voxel_grid_batch(ev, off, 5, (224, 224), out=out, **kw)
        e1.record(); torch.cuda.synchronize()
        print(f"algo {algo} tile_rows {rows or 'auto':>4}: {what:24s} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us", flush=True)
call("evp_voxel_set_debug", 0)
