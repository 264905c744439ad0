#!/usr/bin/env python3
"""Steady-state check of the GEMM kernels on large square problems (random bf16 data)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eventpretrain_amd import ops
from eventpretrain_amd._lib import call

def bench(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

for (M, N, K) in [(4096, 4096, 4096), (8192, 8192, 8192), (6272, 3072, 768), (6272, 3072, 6144), (25088, 3072, 768)]:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for variant, tile in [(1, 1), (1, 4), (2, 1)]:
        call("evp_gemm_set_variant", variant)
        t = bench(lambda: ops.gemm(x, w, y, M=M, N=N, K=K, tile=tile))
        print(f"{M}x{N}x{K} variant={variant} tile={tile}: {t*1e6:9.1f} us {2.0*M*N*K/t/1e12:7.1f} TF", flush=True)

print("--- layouts at 4096^3 (tile 128x128, glds) and wgrad shapes")
call("evp_gemm_set_variant", 1)
M = N = K = 4096
a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
yf = torch.empty(M, N, device="cuda")
for ta, tb in [(False, False), (False, True), (True, True)]:
    t = bench(lambda: ops.gemm(a, b, yf, M=M, N=N, K=K, trans_a=ta, trans_b=tb, lda=K, ldb=K, tile=1, splitk=1))
    print(f"4096^3 transA={ta} transB={tb} f32 out: {t*1e6:9.1f} us {2.0*M*N*K/t/1e12:7.1f} TF", flush=True)
Mt, Nn, Kk = 6272, 3072, 768
dy = torch.randn(Mt, Nn, device="cuda").bfloat16(); x = torch.randn(Mt, Kk, device="cuda").bfloat16()
dw = torch.empty(Nn, Kk, device="cuda")
for sk in (1, 2, 4, 8):
    for tile in (1, 2):
        t = bench(lambda: ops.gemm(dy, x, dw, M=Nn, N=Kk, K=Mt, trans_a=True, trans_b=True, lda=Nn, ldb=Kk, tile=tile, splitk=sk))
        print(f"wgrad fc1 splitk={sk} tile={tile}: {t*1e6:9.1f} us {2.0*Mt*Nn*Kk/t/1e12:7.1f} TF", flush=True)
