#!/usr/bin/env python3
"""Reference point, not part of the product: what the vendor GEMM library (hipBLASLt / rocBLAS behind torch.mm) reaches
on the step's GEMM shapes, bf16 operands, f32 accumulate, no epilogue. Run under `rocprofv3 --kernel-trace --stats` to
see which macro-tile it picks per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

def bench(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3

Me, Md = 64 * 98, 64 * 196
shapes = [("enc.qkv", Me, 2304, 768), ("enc.proj", Me, 768, 768), ("enc.fc1", Me, 3072, 768), ("enc.fc2", Me, 768, 3072),
          ("dec.qkv", Md, 1536, 512), ("dec.proj", Md, 512, 512), ("dec.fc1", Md, 2048, 512), ("dec.fc2", Md, 512, 2048)]
T = torch.bfloat16
for name, M, N, K in shapes:
    x = torch.randn(M, K, device="cuda").to(T)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(T)
    dy = torch.randn(M, N, device="cuda").to(T)
    t_f = bench(lambda: torch.mm(x, w.t()))          # forward  NT
    t_d = bench(lambda: torch.mm(dy, w))             # dgrad    NN
    t_w = bench(lambda: torch.mm(dy.t(), x))         # wgrad    TN
    fl = 2.0 * M * N * K
    print("%-9s %6dx%5dx%5d  fwd %6.1fus %6.1fTF  dgrad %6.1fus %6.1fTF  wgrad %6.1fus %6.1fTF" %
          (name, M, N, K, t_f * 1e6, fl / t_f / 1e12, t_d * 1e6, fl / t_d / 1e12, t_w * 1e6, fl / t_w / 1e12), flush=True)
