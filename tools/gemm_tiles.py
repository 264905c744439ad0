#!/usr/bin/env python3
"""Tile-variant sweep of evp_gemm on the step's hottest shapes (bf16, random data)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eventpretrain_amd import ops
from eventpretrain_amd._lib import ACT_GELU

def bench(fn, reps=20, warm=4):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

shapes = [("enc.qkv", 6272, 2304, 768), ("enc.proj", 6272, 768, 768), ("enc.fc1", 6272, 3072, 768), ("enc.fc2", 6272, 768, 3072),
          ("cvit.proj", 12544, 768, 768), ("cvit.fc2", 12544, 768, 3072), ("cvit.qkv", 12544, 2304, 768), ("cvit.fc1", 12544, 3072, 768), ("dec.qkv", 12544, 1536, 512), ("dec.fc1", 12544, 2048, 512), ("dec.fc2", 12544, 512, 2048), ("4096^3", 4096, 4096, 4096)]
for name, M, N, K in shapes:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    dy = torch.randn(M, N, device="cuda").bfloat16(); dx = torch.empty(M, K, device="cuda", dtype=torch.bfloat16)
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16); bias = torch.randn(N, device="cuda")
    row = [f"{name:9s}"]
    outs = {}
    for tile in (1, 4):
        t1 = bench(lambda: ops.gemm(x, w, y, M=M, N=N, K=K, bias=bias, tile=tile))
        t2 = bench(lambda: ops.gemm(dy, w, dx, M=M, N=K, K=N, trans_b=True, ldb=K, tile=tile))
        outs[tile] = (y.float().clone(), dx.float().clone())
        row.append(f"tile{tile}: fwd {t1*1e6:6.1f}us {2.0*M*N*K/t1/1e12:5.0f}TF dgrad {t2*1e6:6.1f}us {2.0*M*N*K/t2/1e12:5.0f}TF")
    same = all(torch.equal(a, b) for a, b in zip(outs[1], outs[4]))
    print(" | ".join(row), "| tile 4 == tile 1 bit for bit:", same, flush=True)
