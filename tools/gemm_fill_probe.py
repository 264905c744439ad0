#!/usr/bin/env python3
"""How the 128x128 kernel's time depends on how many tiles each CU holds (256 CUs, two resident workgroups per CU):
same K, tile counts 256 (one lone workgroup per CU), 294, 384, 512 (two per CU), 768."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eventpretrain_amd import ops

def bench(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

T = torch.bfloat16
for K in (768, 3072):
    for tm, tn in ((32, 8), (49, 6), (48, 8), (64, 8), (96, 8), (128, 8)):
        M, N = tm * 128, tn * 128
        x = torch.randn(M, K, device="cuda").to(T); w = (torch.randn(N, K, device="cuda") * 0.05).to(T)
        o = torch.empty(M, N, dtype=T, device="cuda")
        tiles = [int(t) for t in os.environ.get("TILES", "1").split(",")]
        line = "K=%4d tiles=%4d (%dx%d) " % (K, tm * tn, tm, tn)
        for t in tiles:
            us = bench(lambda: ops.gemm(x, w, o, M=M, N=N, K=K, tile=t))
            line += " | tile %2d: %6.1f us %6.1f TF" % (t, us, 2.0 * M * N * K / us / 1e6)
        print(line, flush=True)
