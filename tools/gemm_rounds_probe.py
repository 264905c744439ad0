#!/usr/bin/env python3
"""What does the last, partly filled round of 128x128 tiles cost? One shape, M varied around whole multiples of 512 tiles; a captured
graph of 100 dependent launches replayed back to back."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eventpretrain_amd import ops  # noqa: E402
from eventpretrain_amd._lib import ACT_DGELU, ACT_GELU  # noqa: E402

ops.set_compute_dtype(torch.bfloat16)
TILE = 0
for v in list(sys.argv[1:]):
    if v.startswith("tile="):
        TILE = int(v[5:])
        sys.argv.remove(v)
for v in sys.argv[1:]:          # 101: the epilogue is skipped (what the K loops alone cost)
    from eventpretrain_amd._lib import call
    call("evp_gemm_set_variant", int(v))
    print("evp_gemm_set_variant(%s)" % v)
for name, kind, N, K, Ms in [("dec.dfc2", "d", 2048, 512, (8192, 12288, 12544, 13312, 14336, 16384)),
                              ("enc.fc1", "g", 3072, 768, (2688, 5376, 5504, 6272, 7168, 8064)),
                              ("enc.proj", "f", 768, 768, (4096, 5376, 6272, 8192, 10880)),]:
    for M in Ms:
        a = torch.randn(M, K, device="cuda").bfloat16()
        w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        wt = w.t().contiguous()
        bias = torch.randn(N, device="cuda")
        c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        aux = torch.randn(M, N, device="cuda").bfloat16()

        def one():
            if kind == "f":
                ops.gemm(a, w, c, M=M, N=N, K=K, bias=bias, tile=TILE)
            elif kind == "g":
                ops.gemm(a, w, c, M=M, N=N, K=K, bias=bias, act=ACT_GELU, aux=aux, tile=TILE)
            else:
                ops.gemm(a, wt, c, M=M, N=N, K=K, trans_b=True, ldb=N, act=ACT_DGELU, aux=aux, tile=TILE)

        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            one()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(100):
                one()
        ts = []
        for r in range(3):
            g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                g.replay()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 500 * 1e6)
        tiles = ((M + 127) // 128) * ((N + 127) // 128)
        us = sorted(ts)[1]
        print(f"{name:9s} M={M:6d} tiles {tiles:5d} = {tiles / 512:5.2f} rounds: {us:6.1f} us  {2.0 * M * N * K / us * 1e-6:5.0f} TF", flush=True)
