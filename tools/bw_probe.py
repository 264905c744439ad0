#!/usr/bin/env python3
"""Raw HBM write / copy rates at the GEMM output sizes of the step (calibration for the epilogue cost)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eventpretrain_amd import ops  # noqa: E402
from tools.gemm_bench import bench  # noqa: E402

for M, N in ((6272, 3072), (6272, 768), (12544, 2048)):
    yb = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    yf = torch.empty(M, N, device="cuda")
    xf = torch.randn(M, N, device="cuda")
    t_fill_b = bench(lambda: yb.fill_(1.0))
    t_fill_f = bench(lambda: yf.fill_(1.0))
    t_copy_f = bench(lambda: yf.copy_(xf))
    t_cast = bench(lambda: ops.cast(xf, torch.bfloat16))
    t_add = bench(lambda: ops.add(xf, yf))
    print(f"{M}x{N}: fill bf16 {t_fill_b*1e6:6.1f}us ({M*N*2/t_fill_b/1e12:.2f} TB/s)  fill f32 {t_fill_f*1e6:6.1f}us ({M*N*4/t_fill_f/1e12:.2f} TB/s)  "
          f"copy f32 {t_copy_f*1e6:6.1f}us ({M*N*8/t_copy_f/1e12:.2f} TB/s)  cast f32->bf16 {t_cast*1e6:6.1f}us ({M*N*6/t_cast/1e12:.2f} TB/s)  "
          f"add {t_add*1e6:6.1f}us ({M*N*12/t_add/1e12:.2f} TB/s)", flush=True)
