#!/usr/bin/env python3
"""Correctness (exact on small-integer data) and speed of the 256x256 ring kernel (evp_gemm tile=6) vs the 128x128 kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eventpretrain_amd import ops  # noqa: E402
from tools.gemm_bench import bench  # noqa: E402

T = torch.bfloat16


def check(M, N, K, trans_b, out_dtype, tile=6, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randint(-3, 4, (M, K), generator=g, device="cuda").to(T)
    if trans_b:
        b = torch.randint(-3, 4, (K, N), generator=g, device="cuda").to(T)
        ref = a.float() @ b.float()
    else:
        b = torch.randint(-3, 4, (N, K), generator=g, device="cuda").to(T)
        ref = a.float() @ b.float().t()
    bias = torch.randint(-2, 3, (N,), generator=g, device="cuda").float()
    out = torch.full((M, N), 7.0, device="cuda", dtype=out_dtype)
    ops.gemm(a, b, out, M=M, N=N, K=K, trans_b=trans_b, ldb=(N if trans_b else K), bias=bias, tile=tile)
    torch.cuda.synchronize()
    ref = ref + bias
    if out_dtype == T:
        ref = ref.to(T).float()
    bad = (out.float() != ref).sum().item()
    if bad:
        print(f"check tile={tile} M={M} N={N} K={K} transB={int(trans_b)} out={str(out_dtype)[6:]}: {bad} MISMATCHES", flush=True)
    return bad == 0


def check_tn(M, N, K, tile=6, seed=0, accumulate=False):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randint(-3, 4, (K, M), generator=g, device="cuda").to(T)
    b = torch.randint(-3, 4, (K, N), generator=g, device="cuda").to(T)
    ref = a.float().t() @ b.float()
    out = torch.full((M, N), 2.0, device="cuda")
    ops.gemm(a, b, out, M=M, N=N, K=K, trans_a=True, trans_b=True, lda=M, ldb=N, tile=tile, accumulate=accumulate, splitk=1)
    torch.cuda.synchronize()
    if accumulate:
        ref = ref + 2.0
    bad = (out != ref).sum().item()
    if bad:
        print(f"check TN tile={tile} M={M} N={N} K={K} acc={accumulate}: {bad} MISMATCHES", flush=True)
    return bad == 0


def main():
    ok = True
    for (M, N, K) in ((256, 256, 64), (256, 512, 192), (768, 768, 6272), (96, 288, 3456), (304, 520, 320), (2304, 768, 6272)):
        for acc in (False, True):
            ok &= check_tn(M, N, K, accumulate=acc)
    for (M, N, K) in ((256, 256, 64), (256, 256, 128), (256, 256, 192), (512, 768, 768), (6272, 768, 768), (300, 520, 320), (4096, 4096, 1024)):
        for tb in (False, True):
            for od in (torch.float32, T):
                for tile in (6,):
                    for rep in range(3 if M >= 4096 else 1):
                        ok &= check(M, N, K, tb, od, tile=tile, seed=rep)
    print("checks", "OK" if ok else "FAILED", flush=True)
    if not ok:
        sys.exit(1)
    if len(sys.argv) > 1 and sys.argv[1] == "--check-only":
        return
    shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (6272, 2304, 768), (6272, 768, 768), (6272, 3072, 768), (6272, 768, 3072),
              (12544, 1536, 512), (12544, 512, 512), (12544, 2048, 512), (12544, 512, 2048)]
    for M, N, K in shapes:
        x = torch.randn(M, K, device="cuda").to(T)
        w = (torch.randn(N, K, device="cuda") * 0.05).to(T)
        wt = w.t().contiguous()
        y = torch.empty(M, N, device="cuda", dtype=T)
        fl = 2.0 * M * N * K
        row = [f"{M}x{N}x{K}"]
        for tile in (1, 6):
            s1 = bench(lambda: ops.gemm(x, w, y, M=M, N=N, K=K, tile=tile))
            s2 = bench(lambda: ops.gemm(x, wt, y, M=M, N=N, K=K, trans_b=True, ldb=N, tile=tile))
            row.append(f"tile{tile}: NT {s1 * 1e6:7.1f}us {fl / s1 / 1e12:6.0f}TF  NN {s2 * 1e6:7.1f}us {fl / s2 / 1e12:6.0f}TF")
        print("   ".join(row), flush=True)
    bench_tn()


def bench_tn():
    for M, N, K in ((2304, 768, 6272), (768, 768, 6272), (3072, 768, 6272), (768, 3072, 6272), (1536, 512, 12544), (512, 512, 12544),
                    (2048, 512, 12544), (512, 2048, 12544), (4096, 4096, 4096)):
        a = torch.randn(K, M, device="cuda").to(T)
        b = torch.randn(K, N, device="cuda").to(T)
        out = torch.empty(M, N, device="cuda")
        fl = 2.0 * M * N * K
        row = [f"TN {M}x{N}x{K}"]
        for tile in (1, 6):
            sec = bench(lambda: ops.gemm(a, b, out, M=M, N=N, K=K, trans_a=True, trans_b=True, lda=M, ldb=N, tile=tile, splitk=1))
            row.append(f"tile{tile} {sec * 1e6:7.1f}us {fl / sec / 1e12:6.0f}TF")
        print("   ".join(row), flush=True)


if __name__ == "__main__":
    main()
