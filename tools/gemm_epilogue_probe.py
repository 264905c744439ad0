#!/usr/bin/env python3
"""Per-workgroup timeline of one GEMM launch from the in-kernel stamps (10 ns wall clock): when each workgroup starts, when its
K loop ends (evp_gemm_set_variant(103) moves the start stamp there) and when it exits -- how long a tile's epilogue takes and how
the rounds of a multi-round launch line up."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from eventpretrain_amd import ops  # noqa: E402
from eventpretrain_amd._lib import ACT_DGELU, ACT_GELU, call  # noqa: E402

ops.set_compute_dtype(torch.bfloat16)
WGS = 4096
cases = [("enc.proj", "f", 6272, 768, 768), ("enc.qkv", "f", 6272, 2304, 768), ("enc.fc1", "g", 6272, 3072, 768), ("dec.dfc2", "d", 12544, 2048, 512),
         ("dec.fc2", "r", 12544, 512, 2048)]
for name, kind, M, N, K in cases:
    a = torch.randn(M, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    wt = w.t().contiguous()
    bias = torch.randn(N, device="cuda")
    c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    cf = torch.empty(M, N, device="cuda", dtype=torch.float32)
    res = torch.randn(M, N, device="cuda")
    aux = torch.randn(M, N, device="cuda").bfloat16()

    def one():
        if kind == "f":
            ops.gemm(a, w, c, M=M, N=N, K=K, bias=bias, tile=1)
        elif kind == "g":
            ops.gemm(a, w, c, M=M, N=N, K=K, bias=bias, act=ACT_GELU, aux=aux, tile=1)
        elif kind == "r":
            ops.gemm(a, w, cf, M=M, N=N, K=K, bias=bias, residual=res, tile=1)
        else:
            ops.gemm(a, wt, c, M=M, N=N, K=K, trans_b=True, ldb=N, act=ACT_DGELU, aux=aux, tile=1)

    out = {}
    for mode in (100, 103):
        call("evp_gemm_set_variant", mode)
        for _ in range(3):
            one()
        torch.cuda.synchronize()
        buf = torch.zeros(2 * WGS, dtype=torch.int64, device="cuda")
        call("evp_gemm_set_stamp_buffer", buf.data_ptr(), 1)
        one()
        torch.cuda.synchronize()
        call("evp_gemm_set_stamp_buffer", None, 0)
        st = buf.cpu().numpy().view(np.uint64).reshape(-1, 2)
        n = ((M + 127) // 128) * ((N + 127) // 128)
        beg, end = (~st[:n, 0]).astype(np.int64), st[:n, 1].astype(np.int64)
        out[mode] = (beg, end)
    call("evp_gemm_set_variant", 100)
    b0, e0 = out[100]
    bk, e1 = out[103]
    t0 = b0.min()
    dur = (e0 - b0) * 0.01
    epi = (e1 - bk) * 0.01
    order = np.argsort(b0)
    first = order[:min(512, n)]
    print(f"{name:9s} {M}x{N}x{K} {n} tiles: launch {0.01 * (e0.max() - t0):6.1f} us | workgroup life mean {dur.mean():5.1f} (first round {dur[first].mean():5.1f}) us | "
          f"epilogue (K loop end -> exit) mean {epi.mean():5.2f} p10 {np.percentile(epi, 10):5.2f} p90 {np.percentile(epi, 90):5.2f} us | "
          f"first-round starts spread {0.01 * (b0[first].max() - t0):4.1f} us, ends {0.01 * (e0[first].min() - t0):5.1f}..{0.01 * (e0[first].max() - t0):5.1f} us", flush=True)
