#!/usr/bin/env python3
"""Attention-core micro-benchmark (fused per-head kernels vs batched-GEMM formulation) on the step's two shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eventpretrain_amd import ops  # noqa: E402


def bench(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


for name, B, N, h, dh in [("enc", 64, 98, 12, 64), ("dec", 64, 196, 16, 32)]:
    C = h * dh
    qkv = (torch.randn(B * N, 3 * C, device="cuda") * 0.8).bfloat16()
    dout = torch.randn(B * N, C, device="cuda").bfloat16()
    out, lse, _ = ops.attention_fused_fwd(qkv, B, N, h, dh)
    fl_f = 4.0 * B * h * N * N * dh
    t = bench(lambda: ops.attention_fused_fwd(qkv, B, N, h, dh))
    print(f"{name} fused fwd   {t*1e6:8.1f} us  {fl_f/t/1e12:6.1f} TF (algorithmic 2 products)")
    t = bench(lambda: ops.attention_fused_bwd(qkv, out, dout, lse, B, N, h, dh))
    print(f"{name} fused bwd   {t*1e6:8.1f} us  {2.5*fl_f/t/1e12:6.1f} TF (algorithmic 5 products)")
    probs, o2 = ops.attention_fwd(qkv, B, N, h, dh)
    t = bench(lambda: ops.attention_fwd(qkv, B, N, h, dh))
    print(f"{name} unfused fwd {t*1e6:8.1f} us")
    t = bench(lambda: ops.attention_bwd(qkv, probs, dout, B, N, h, dh))
    print(f"{name} unfused bwd {t*1e6:8.1f} us")

# per-wave cycle split of the fused forward (staging vs strips)
from eventpretrain_amd._lib import call  # noqa: E402
for name, B, N, h, dh in [("enc", 64, 98, 12, 64), ("dec", 64, 196, 16, 32)]:
    C = h * dh
    qkv = (torch.randn(B * N, 3 * C, device="cuda") * 0.8).bfloat16()
    dbg = torch.zeros(B * h * 4 * 2, dtype=torch.int64, device="cuda")
    call("evp_attention_set_debug_buffer", dbg.data_ptr())
    for _ in range(3):
        ops.attention_fused_fwd(qkv, B, N, h, dh)
    torch.cuda.synchronize()
    call("evp_attention_set_debug_buffer", None)
    d = dbg.cpu().numpy().reshape(-1, 2).astype(float)
    print(f"{name} fwd: staging cycles mean {d[:, 0].mean():.0f} max {d[:, 0].max():.0f}; strips cycles per wave mean {d[:, 1].mean():.0f} max {d[:, 1].max():.0f}")

# per-wave cycle split of the fused backward (staging / query-on-lane pass / key-on-lane pass)
for name, B, N, h, dh in [("enc", 64, 98, 12, 64), ("dec", 64, 196, 16, 32)]:
    C = h * dh
    qkv = (torch.randn(B * N, 3 * C, device="cuda") * 0.8).bfloat16()
    dout = torch.randn(B * N, C, device="cuda").bfloat16()
    out, lse, _ = ops.attention_fused_fwd(qkv, B, N, h, dh)
    dbg = torch.zeros(B * h * 8 * 3, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    call("evp_attention_set_debug_buffer", dbg.data_ptr())
    for _ in range(3):
        ops.attention_fused_bwd(qkv, out, dout, lse, B, N, h, dh)
    torch.cuda.synchronize()
    call("evp_attention_set_debug_buffer", None)
    d = dbg.cpu().numpy().reshape(-1, 3).astype(float)
    print(f"{name} bwd cycles per wave: staging mean {d[:, 0].mean():.0f} max {d[:, 0].max():.0f}; pass 1 (dQ) mean {d[:, 1].mean():.0f} max {d[:, 1].max():.0f}; "
          f"pass 2 (dK, dV) mean {d[:, 2].mean():.0f} max {d[:, 2].max():.0f}")
