#!/usr/bin/env python3
"""Probe (not part of the product): LayerNorm backward (evp_layernorm_bwd_cs: dx f32 + bf16 copy + three partial rows per block)
at the step's two shapes, HIP-event timed. EVP_LN_BWD_BLOCKS=<n> changes the grid cap (default 1024).
usage: ln_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eventpretrain_amd.ops import call, ptr, dt, stream_ptr  # noqa: E402


def main():
    for name, M, D in (("enc", 6272, 768), ("dec", 12544, 512)):
        dy = torch.randn(M, D, device="cuda")
        x = torch.randn(M, D, device="cuda")
        gres = torch.randn(M, D, device="cuda")
        gamma = torch.randn(D, device="cuda")
        mean, rstd = x.mean(1), 1.0 / x.std(1)
        dx = torch.empty_like(x)
        lp = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
        nb = call("evp_layernorm_bwd_nblk", M)
        ws = torch.empty(nb, 3 * D, device="cuda")
        run = lambda: call("evp_layernorm_bwd_cs", ptr(dy), dt(dy), ptr(x), 0, 0, ptr(gamma), ptr(mean), ptr(rstd), ptr(gres), M, D,
                           ptr(dx), ptr(lp), ptr(ws), stream_ptr())
        for _ in range(5):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        byt = M * D * (4 + 4 + 4 + 4 + 2) + nb * 3 * D * 4
        print(f"{name} M={M} D={D} blocks={nb}: {us:6.1f} us  {byt / us * 1e-6:5.2f} TB/s algorithmic", flush=True)


if __name__ == "__main__":
    main()
