#!/usr/bin/env python3
"""Re-launches one GEMM shape back to back for `seconds` (default 20) and prints its rate every ~2 s: what the card sustains on
that kernel alone (clock_watch.sh samples the clocks meanwhile). usage: gemm_hold.py [seconds] [tile]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eventpretrain_amd import ops  # noqa: E402
from eventpretrain_amd._lib import ACT_GELU  # noqa: E402

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ops.set_compute_dtype(torch.bfloat16)
M, N, K = 6272, 3072, 768
a = torch.randn(M, K, device="cuda").bfloat16()
w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
bias = torch.randn(N, device="cuda")
c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
aux = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    ops.gemm(a, w, c, M=M, N=N, K=K, bias=bias, act=ACT_GELU, aux=aux, tile=tile)
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=s):
    for _ in range(200):
        ops.gemm(a, w, c, M=M, N=N, K=K, bias=bias, act=ACT_GELU, aux=aux, tile=tile)
t_end = time.time() + secs
while time.time() < t_end:
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(150):
        g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n = 150 * 200
    print(f"enc.fc1+GELU tile {tile}: {dt / n * 1e6:6.1f} us per launch, {2.0 * M * N * K * n / dt * 1e-12:6.0f} TFLOP/s", flush=True)
