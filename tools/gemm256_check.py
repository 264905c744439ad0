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
            ok &= check_tn(M, N, K, tile=7, accumulate=acc)
    # tile 8: deferred-epilogue persistent kernel, incl. bias + GELU/aux, dGELU, residual
    from eventpretrain_amd._lib import ACT_DGELU, ACT_GELU
    for (M, N, K) in ((128, 128, 256), (256, 384, 320), (6272, 768, 768), (1280, 2304, 256), (12544, 512, 2048)):
        g = torch.Generator(device="cuda").manual_seed(M + N)
        a = torch.randint(-2, 3, (M, K), generator=g, device="cuda").to(T)
        b = torch.randint(-2, 3, (N, K), generator=g, device="cuda").to(T)
        bt = b.t().contiguous()
        bias = torch.randint(-2, 3, (N,), generator=g, device="cuda").float()
        res = torch.randint(-4, 5, (M, N), generator=g, device="cuda").float()
        ref = a.float() @ b.float().t() + bias
        for tb in (False, True):
            for od in (torch.float32, T):
                for use_res in (False, True):
                    if use_res and od == T:
                        continue
                    out = torch.full((M, N), 7.0, device="cuda", dtype=od)
                    ops.gemm(a, bt if tb else b, out, M=M, N=N, K=K, trans_b=tb, ldb=(N if tb else K), bias=bias, residual=res if use_res else None, tile=8)
                    r = ref + (res if use_res else 0)
                    r = r.to(od).float()
                    bad = (out.float() != r).sum().item()
                    if bad:
                        ok = False
                        print(f"tile8 M={M} N={N} K={K} tb={tb} out={od} res={use_res}: {bad} MISMATCHES", flush=True)
        # GELU forward with aux (NT) and dGELU (NN) against the default kernel
        h1, a1 = torch.empty(M, N, device="cuda", dtype=T), torch.empty(M, N, device="cuda", dtype=T)
        h8, a8 = torch.empty_like(h1), torch.empty_like(a1)
        ops.gemm(a, b, h1, M=M, N=N, K=K, bias=bias, act=ACT_GELU, aux=a1, alpha=0.125, tile=1)
        ops.gemm(a, b, h8, M=M, N=N, K=K, bias=bias, act=ACT_GELU, aux=a8, alpha=0.125, tile=8)
        d1, d8 = torch.empty(M, N, device="cuda", dtype=T), torch.empty(M, N, device="cuda", dtype=T)
        ops.gemm(a, bt, d1, M=M, N=N, K=K, trans_b=True, ldb=N, act=ACT_DGELU, aux=a1, alpha=0.125, tile=1)
        ops.gemm(a, bt, d8, M=M, N=N, K=K, trans_b=True, ldb=N, act=ACT_DGELU, aux=a1, alpha=0.125, tile=8)
        torch.cuda.synchronize()
        for nm, x1, x8 in (("gelu", h1, h8), ("aux", a1, a8), ("dgelu", d1, d8)):
            if not torch.equal(x1, x8):
                ok = False
                print(f"tile8 {nm} M={M} N={N} K={K}: differs from tile 1 ({(x1 != x8).sum().item()} elements)", flush=True)
    for (M, N, K) in ((256, 256, 64), (256, 256, 128), (256, 256, 192), (512, 768, 768), (6272, 768, 768), (300, 520, 320), (4096, 4096, 1024)):
        for tb in (False, True):
            for od in (torch.float32, T):
                for tile in (6, 7):
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
        for tile in (1, 6, 7):
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
