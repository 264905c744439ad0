#!/usr/bin/env python3
"""Do parallel branches of a captured HIP graph run concurrently on this runtime? Two independent chains of GEMM launches
(294 tiles each: 57 % of the 512 workgroup slots) captured (a) on one stream, (b) forked onto two streams; plus the same two
chains launched eagerly on two streams (launch-bound, for reference)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eventpretrain_amd import ops  # noqa: E402

M, N, K, L = 6272, 768, 768, 24
dev = "cuda"
xs = [torch.randn(M, K, device=dev).bfloat16() for _ in range(2)]
w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
bufs = [[torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(2)] for _ in range(2)]


def chain(c):
    src = xs[c]
    for i in range(L):
        dst = bufs[c][i & 1]
        ops.gemm(src, w, dst, M=M, N=N, K=K, trans_b=True)
        src = dst


def capture(fork):
    g = torch.cuda.CUDAGraph()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    s1.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1):
        chain(0); chain(1)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s1):
        if fork:
            s2.wait_stream(s1)
            chain(0)
            with torch.cuda.stream(s2):
                chain(1)
            s1.wait_stream(s2)
        else:
            chain(0); chain(1)
    return g


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


g_serial, g_fork = capture(False), capture(True)
for r in range(3):
    print(f"round {r}: serial graph {timeit(g_serial.replay):.3f} ms | forked graph {timeit(g_fork.replay):.3f} ms", flush=True)
# two graphs, one per chain, replayed on two streams
ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.graph(ga, stream=sa):
    chain(0)
with torch.cuda.graph(gb, stream=sb):
    chain(1)


def two():
    with torch.cuda.stream(sa):
        ga.replay()
    with torch.cuda.stream(sb):
        gb.replay()


print(f"one chain alone {timeit(ga.replay):.3f} ms | two graphs on two streams {timeit(two):.3f} ms", flush=True)
