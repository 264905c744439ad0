#!/usr/bin/env python3
"""G4 weight-gradient body (evp_gemm tile 9): time against K for one round of 256 tiles (4096 x 4096 output) -- the intercept is what a
tile pays outside its K loop (ring fill + the f32 C store)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eventpretrain_amd import ops
ops.set_compute_dtype(torch.bfloat16)
M = N = 4096
pts = []
for K in (512, 1024, 2048, 4096, 8192, 12544):
    a = torch.randn(K, M, device="cuda").bfloat16()
    b = torch.randn(K, N, device="cuda").bfloat16()
    c = torch.empty(M, N, device="cuda")
    f = lambda: ops.gemm(a, b, c, M=M, N=N, K=K, trans_a=True, trans_b=True, lda=M, ldb=N, tile=9)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(20):
            f()
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 100 * 1e6
    pts.append((K, us))
    print(f"K={K:6d}: {us:8.1f} us  {2.0 * M * N * K / us * 1e-6:6.0f} TF", flush=True)
(k1, t1), (k2, t2) = pts[2], pts[-1]
slope = (t2 - t1) / (k2 - k1)
print(f"slope {slope * 1000:.2f} us per 1000 k  intercept {t1 - slope * k1:.1f} us (from K={k1} and K={k2})")
