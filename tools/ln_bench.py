#!/usr/bin/env python3
"""LayerNorm forward / backward at the step's shapes: device time (HIP graph of 20 calls, so the Python/ctypes launch
cost does not hide the kernels) and algorithmic HBM rate."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eventpretrain_amd import ops  # noqa: E402


def graph_time(fn, n=20, reps=10):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / (n * reps)


if __name__ == "__main__":
    for M, D in ((6272, 768), (12544, 512)):
        x = torch.randn(M, D, device="cuda")
        g = torch.randn(D, device="cuda")
        b = torch.randn(D, device="cuda")
        gres = torch.randn(M, D, device="cuda")
        y, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6, torch.bfloat16)
        dy = torch.randn(M, D, device="cuda").bfloat16()
        tf = graph_time(lambda: ops.layernorm_fwd(x, g, b, 1e-6, torch.bfloat16))
        tb = graph_time(lambda: ops.layernorm_bwd(dy, x, g, mean, rstd, gres=gres, want_lp=True))
        bf = M * D * (4 + 2)
        bb = M * D * (2 + 4 + 4 + 4 + 2)
        print(f"LN {M}x{D}: fwd {tf * 1e6:6.1f} us ({bf / tf / 1e12:.2f} TB/s)   bwd+finalize {tb * 1e6:6.1f} us ({bb / tb / 1e12:.2f} TB/s)", flush=True)
