#!/usr/bin/env python3
"""Tile sweep of evp_gemm on the wide-output GEMMs of the step (qkv / fc1 forward, fc2 data gradient; encoder, decoder and the
contrastive heads), bf16, random data: the 128x128 body (tile 1) against the G4 bodies 256x256 / 256x128 / 128x256 (tiles
20 / 21 / 22), interleaved rounds in one process on one device (boards differ by several percent). HIP events around
back-to-back re-launches: a warm-cache figure, good for ranking variants; the in-step figure is bench.py's `roofline`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eventpretrain_amd import ops  # noqa: E402
from eventpretrain_amd._lib import ACT_DGELU, ACT_GELU  # noqa: E402

TILES = (1, 20, 21, 22)


def bench(fn, reps=12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    Me, Md = 64 * 98, 64 * 196
    # (name, kind, M, N, K): kind f = forward + bias, g = forward + bias + GELU + pre-activation store, d = data gradient with GELU',
    # n = plain data gradient
    cases = [("enc.qkv", "f", Me, 2304, 768), ("enc.fc1", "g", Me, 3072, 768), ("enc.dfc2", "d", Me, 3072, 768),
             ("dec.qkv", "f", Md, 1536, 512), ("dec.fc1", "g", Md, 2048, 512), ("dec.dfc2", "d", Md, 2048, 512),
             ("con.qkv", "f", Md, 2304, 768), ("con.fc1", "g", Md, 3072, 768), ("con.dfc2", "d", Md, 3072, 768),
             ("head.4096", "f", Md, 4096, 768), ("head.d4096", "n", Md, 4096, 768), ("sq4096", "f", 4096, 4096, 4096)]
    for name, kind, M, N, K in cases:
        if only and only not in name:
            continue
        a = torch.randn(M, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        wt = w.t().contiguous()
        bias = torch.randn(N, device="cuda")
        c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        aux = torch.randn(M, N, device="cuda").bfloat16()
        fns = {}
        for t in TILES:
            if kind == "f":
                fns[t] = lambda t=t: ops.gemm(a, w, c, M=M, N=N, K=K, bias=bias, tile=t)
            elif kind == "g":
                fns[t] = lambda t=t: ops.gemm(a, w, c, M=M, N=N, K=K, bias=bias, act=ACT_GELU, aux=aux, tile=t)
            elif kind == "d":
                fns[t] = lambda t=t: ops.gemm(a, wt, c, M=M, N=N, K=K, trans_b=True, ldb=N, act=ACT_DGELU, aux=aux, tile=t)
            else:
                fns[t] = lambda t=t: ops.gemm(a, wt, c, M=M, N=N, K=K, trans_b=True, ldb=N, tile=t)
        times = {t: [] for t in TILES}
        for t in TILES:
            for _ in range(3):
                fns[t]()
        for _ in range(4):
            for t in TILES:
                times[t].append(bench(fns[t]))
        fl = 2.0 * M * N * K
        row = [f"{name:10s} {kind} {M:5d}x{N:4d}x{K:4d}"]
        for t in TILES:
            us = sorted(times[t])[len(times[t]) // 2]
            row.append(f"t{t}: {us:6.1f}us {fl / us * 1e-6:5.0f}TF")
        print(" | ".join(row), flush=True)


if __name__ == "__main__":
    ops.set_compute_dtype(torch.bfloat16)
    main()
