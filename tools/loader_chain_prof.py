#!/usr/bin/env python3
"""The loader chain alone (bench.py's `loader_chain` workload: 64 clips of 150 k events on a 640 x 480 sensor, 100 k-event windows),
10 replays of the self-driven captured chain (+ its warm-up launches) -- run under `rocprofv3 --kernel-trace --stats` for the per-kernel split."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eventpretrain_amd.dataset.pretrain.gpu_input_pipeline import GpuInputPipeline
from eventpretrain_amd.testing import make_args, synthetic_events
B = 64
pa = make_args(crop_min=0.8, input_size=224, fix_events_num=100_000, img_sensor_w=640, img_sensor_h=480, device="cuda")
clip = synthetic_events(4242, 150_000, width=640, height=480)
ev = torch.from_numpy(np.concatenate([clip] * B, 0)).cuda()
off = np.arange(0, (B + 1) * 150_000, 150_000, dtype=np.int64)
frames = torch.randn(B, 1, 480, 640, device="cuda")
pipe = GpuInputPipeline(pa, seed=1)          # device decision stream; the self-driven captured chain is what bench.py times
chain = pipe.capture(ev, B, frames=frames, clip_offsets=off)
torch.cuda.synchronize()
for i in range(10):
    chain.run_next()
torch.cuda.synchronize()
print("done")
