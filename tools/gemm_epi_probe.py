#!/usr/bin/env python3
"""Main-loop-only time of evp_gemm (epilogue skipped through the measurement switch) vs the full kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eventpretrain_amd import ops  # noqa: E402
from eventpretrain_amd._lib import call  # noqa: E402
from tools.gemm_bench import bench  # noqa: E402

T = torch.bfloat16
for M, N, K in ((6272, 3072, 768), (6272, 768, 768), (12544, 2048, 512), (6272, 768, 3072)):
    x = torch.randn(M, K, device="cuda").to(T)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(T)
    y = torch.empty(M, N, device="cuda", dtype=T)
    yf = torch.empty(M, N, device="cuda")
    res = torch.randn(M, N, device="cuda")
    for tile in (1, 7, 8):
        row = [f"{M}x{N}x{K} tile{tile}"]
        for dbg in (0, 1):
            call("evp_gemm_set_variant", 100 + dbg)
            a = bench(lambda: ops.gemm(x, w, y, M=M, N=N, K=K, tile=tile))
            b = bench(lambda: ops.gemm(x, w, yf, M=M, N=N, K=K, residual=res, tile=tile))
            row.append(f"{'full' if dbg == 0 else 'no-epilogue'}: bf16 {a * 1e6:6.1f}us  f32+res {b * 1e6:6.1f}us")
        call("evp_gemm_set_variant", 100)
        print("   ".join(row), flush=True)

# per-workgroup cycle counters of the persistent kernel (mode 0: full epilogue, mode 2: a quarter of the stores)
import numpy as np  # noqa: E402
dbg = torch.zeros(4 * 512, dtype=torch.int64, device="cuda")
call("evp_gemm_set_debug_buffer", dbg.data_ptr())
for M, N, K in ((6272, 3072, 768), (6272, 768, 768)):
    x = torch.randn(M, K, device="cuda").to(T)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(T)
    y = torch.empty(M, N, device="cuda", dtype=T)
    for mode in (0, 2):
        call("evp_gemm_set_variant", 100 + mode)
        for _ in range(3):
            ops.gemm(x, w, y, M=M, N=N, K=K, tile=7)
        torch.cuda.synchronize()
        call("evp_gemm_set_variant", 100)
        d = dbg.cpu().numpy().reshape(512, 4)
        d = d[d[:, 2] > 0]
        tot, epi, nt = d[:, 0].astype(float), d[:, 1].astype(float), d[:, 2]
        print(f"{M}x{N}x{K} mode {mode}: {len(d)} WGs, tiles/WG {nt.min()}-{nt.max()}, cycles/WG mean {tot.mean():.0f} max {tot.max():.0f}, "
              f"epilogue cycles per tile mean {(epi / nt).mean():.0f} max {(epi / nt).max():.0f}, main cycles per tile {((tot - epi) / nt).mean():.0f}")
call("evp_gemm_set_debug_buffer", None)
