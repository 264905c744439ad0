#!/usr/bin/env python3
"""Fixed (M, N), K swept: separates the per-tile fixed cost (prologue + epilogue + launch) from the per-k-iteration cost
of evp_gemm's NT bf16 kernel. Prints time per K and a least-squares a + b*K fit."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from eventpretrain_amd import ops  # noqa: E402
from tools.gemm_bench import bench  # noqa: E402


def main():
    T = torch.bfloat16
    for M, N in ((6272, 3072), (6272, 768), (12544, 2048), (12544, 512), (4096, 4096)):
        ks, ts = [], []
        for K in (64, 128, 256, 512, 768, 1024, 2048, 3072, 4096):
            x = torch.randn(M, K, device="cuda").to(T)
            w = (torch.randn(N, K, device="cuda") * 0.05).to(T)
            y = torch.empty(M, N, device="cuda", dtype=T)
            sec = bench(lambda: ops.gemm(x, w, y, M=M, N=N, K=K), reps=50)
            ks.append(K)
            ts.append(sec * 1e6)
            print(f"M={M} N={N} K={K:5d}  {sec * 1e6:8.1f} us  {2.0 * M * N * K / sec / 1e12:7.1f} TF", flush=True)
        A = np.stack([np.ones(len(ks)), np.array(ks, dtype=float)], 1)
        (a, b), *_ = np.linalg.lstsq(A, np.array(ts), rcond=None)
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        print(f"  fit: {a:.1f} us fixed + {b * 64:.3f} us per 64-k step; {tiles} tiles ({tiles / 512:.2f} rounds of 512 slots); "
              f"asymptotic {2.0 * M * N / (b * 1e-6) / 1e12:.0f} TF", flush=True)


if __name__ == "__main__":
    main()
