#!/usr/bin/env python3
"""cProfile of the eager (non-graph) training step: where the host time goes."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.argv = ["bench.py"]
import bench
class A: pass
a = A(); a.model = "base"; a.batch = 64; a.dtype = "bf16"
dev = torch.device("cuda", 0)
args, model, opt = bench.build(a, dev)
ev, off, vox, tgt, S, n_ev = bench.make_batch(a, dev, 0)
def step():
    noise = torch.rand(64, 196, device=dev)
    out = model(vox, tgt, is_rec=True, noise=noise)
    out[0].backward()
    opt.step()
    opt.zero_grad(set_to_none=True)
for _ in range(3): step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
print("host ms/step (launch side)", t_host / 5 * 1e3, "wall ms/step", (time.perf_counter() - t0) / 5 * 1e3)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:5000])
