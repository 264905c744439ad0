#!/usr/bin/env python3
"""Loader chain: GPU time per batch of the captured device half alone (same prepared tables replayed) against the pipelined loop with the
host half on the worker thread -- tells whether the loop is bound by the device or by the host."""
import os, sys, time
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
pipe = GpuInputPipeline(pa, seed=1)
chain = pipe.capture(ev, B, frames=frames)
pbs = [pipe.prepare(off, step=i, frame_size=(480, 640)) for i in range(4)]
for k in range(3):
    chain.run(pbs[k % 4])
torch.cuda.synchronize()
for label, n in (("replay only", 48),):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    for k in range(n):
        chain.run(pbs[k % 4])
    b.record()
    torch.cuda.synchronize()
    print(f"{label}: {a.elapsed_time(b) / n * 1e3:.1f} us per batch on the GPU, {(time.perf_counter() - t0) / n * 1e6:.1f} us wall")
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for k in range(48):
    chain.graph.replay()
b.record()
torch.cuda.synchronize()
print(f"graph.replay() alone: {a.elapsed_time(b) / 48 * 1e3:.1f} us")
t0 = time.perf_counter()
for i in range(16):
    pipe.prepare(off, step=50 + i, frame_size=(480, 640))
print(f"host prepare: {(time.perf_counter() - t0) / 16 * 1e3:.3f} ms per batch")
