#!/usr/bin/env python3
"""Micro-benchmark of evp_gemm on the GEMM shapes of the ViT-Base masked-modeling step (B=64): HIP-event timed,
random bf16 operands (never zeros: MI355X clocks higher on zero data). Prints TFLOP/s per shape and layout."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eventpretrain_amd import ops  # noqa: E402
from eventpretrain_amd._lib import ACT_DGELU, ACT_GELU  # noqa: E402


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--splitk", type=int, default=0)
    ap.add_argument("--only", default="")
    ap.add_argument("--variant", type=int, default=0)
    args = ap.parse_args()
    dev = "cuda"
    if args.variant:
        from eventpretrain_amd._lib import call
        call("evp_gemm_set_variant", args.variant)
    T = torch.bfloat16
    Me, Md = 64 * 98, 64 * 196
    shapes = [  # (name, M, N, K) of the forward Linear; dgrad = (M, K, N), wgrad = (N, K, M)
        ("enc.qkv", Me, 2304, 768), ("enc.proj", Me, 768, 768), ("enc.fc1", Me, 3072, 768), ("enc.fc2", Me, 768, 3072),
        ("dec.qkv", Md, 1536, 512), ("dec.proj", Md, 512, 512), ("dec.fc1", Md, 2048, 512), ("dec.fc2", Md, 512, 2048),
        ("patch", Me, 768, 1280), ("dec.embed", Me, 512, 768), ("dec.pred", Md, 256, 512),
    ]
    tot = {"fwd": [0, 0], "dgrad": [0, 0], "wgrad": [0, 0]}
    for name, M, N, K in shapes:
        if args.only and args.only not in name:
            continue
        x = torch.randn(M, K, device=dev).to(T)
        w = (torch.randn(N, K, device=dev) * 0.05).to(T)
        dy = torch.randn(M, N, device=dev).to(T)
        bias = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev)
        y = torch.empty(M, N, device=dev, dtype=T)
        yf = torch.empty(M, N, device=dev)
        aux = torch.empty(M, N, device=dev, dtype=T)
        dx = torch.empty(M, K, device=dev, dtype=T)
        dw = torch.empty(N, K, device=dev)
        fl = 2.0 * M * N * K
        row = [name, f"{M}x{N}x{K}"]
        variants = [
            ("fwd", lambda: ops.gemm(x, w, y, M=M, N=N, K=K, bias=bias, tile=args.tile)),
            ("fwd+res", lambda: ops.gemm(x, w, yf, M=M, N=N, K=K, bias=bias, residual=res, tile=args.tile)),
            ("fwd+gelu", lambda: ops.gemm(x, w, y, M=M, N=N, K=K, bias=bias, act=ACT_GELU, aux=aux, tile=args.tile)),
            ("dgrad", lambda: ops.gemm(dy, w, dx, M=M, N=K, K=N, trans_b=True, ldb=K, tile=args.tile)),
            ("wgrad", lambda: ops.gemm(dy, x, dw, M=N, N=K, K=M, trans_a=True, trans_b=True, lda=N, ldb=K, tile=args.tile, splitk=args.splitk)),
        ]
        for vn, fn in variants:
            sec = bench(fn)
            row.append(f"{vn} {sec * 1e6:7.1f}us {fl / sec / 1e12:6.1f}TF")
            key = "fwd" if vn.startswith("fwd") else vn
            if vn in ("fwd", "dgrad", "wgrad"):
                tot[key][0] += fl
                tot[key][1] += sec
        print("  ".join(row), flush=True)
    for k, (fl, sec) in tot.items():
        if sec:
            print(f"TOTAL {k}: {fl / sec / 1e12:.1f} TF/s")


if __name__ == "__main__":
    main()
