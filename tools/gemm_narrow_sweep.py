#!/usr/bin/env python3
"""Tile sweep of evp_gemm on the NARROW-output GEMMs of the step (N = 768 / 512: proj / fc2 forward with the f32 residual, and the
qkv / proj / fc1 data gradients), bf16, random data: tiles 128x128 (1) and 96x128 (4), interleaved rounds in one
process on one device. Warm re-launch figures (HIP events); ranking only -- the in-step figure is bench.py's `roofline`.
Round-3 result (MI355X): 96x128 wins every N = 768 shape (294 -> 396 tiles), 128x128 every N = 512 shape; the 64x128 / 128x64 half
tiles (three workgroups per CU) lost by 5-30 % and were removed from the library again -- TILES lists what is still built."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eventpretrain_amd import ops  # noqa: E402

TILES = (1, 4)


def bench(fn, reps=12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    Me, Md = 64 * 98, 64 * 196
    # kind r = forward, f32 out + bias + f32 residual; n = data gradient, bf16 out
    cases = [("enc.proj", "r", Me, 768, 768), ("enc.fc2", "r", Me, 768, 3072), ("enc.dqkv", "n", Me, 768, 2304), ("enc.dproj", "n", Me, 768, 768),
             ("enc.dfc1", "n", Me, 768, 3072), ("dec.proj", "r", Md, 512, 512), ("dec.fc2", "r", Md, 512, 2048), ("dec.dqkv", "n", Md, 512, 1536),
             ("dec.dproj", "n", Md, 512, 512), ("dec.dfc1", "n", Md, 512, 2048)]
    for name, kind, M, N, K in cases:
        a = torch.randn(M, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        wt = w.t().contiguous()
        bias = torch.randn(N, device="cuda")
        res = torch.randn(M, N, device="cuda")
        cf = torch.empty(M, N, device="cuda")
        cb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        fns = {}
        for t in TILES:
            if kind == "r":
                fns[t] = lambda t=t: ops.gemm(a, w, cf, M=M, N=N, K=K, bias=bias, residual=res, tile=t)
            else:
                fns[t] = lambda t=t: ops.gemm(a, wt, cb, M=M, N=N, K=K, trans_b=True, ldb=N, tile=t)
        outs = {}
        for t in TILES:
            for _ in range(3):
                fns[t]()
            outs[t] = (cf if kind == "r" else cb).float().clone()
        same = all(torch.equal(outs[1], outs[t]) for t in TILES)
        times = {t: [] for t in TILES}
        for _ in range(4):
            for t in TILES:
                times[t].append(bench(fns[t]))
        fl = 2.0 * M * N * K
        row = [f"{name:10s} {kind} {M:5d}x{N:4d}x{K:4d}"]
        for t in TILES:
            us = sorted(times[t])[len(times[t]) // 2]
            row.append(f"t{t}: {us:6.1f}us {fl / us * 1e-6:5.0f}TF")
        print(" | ".join(row), "| bit-identical:", same, flush=True)


if __name__ == "__main__":
    ops.set_compute_dtype(torch.bfloat16)
    main()
