#!/usr/bin/env python3
"""A/B of whole-step variants in ONE process on ONE device (interleaved rounds; MI355X boards differ by several percent, so
numbers from different gpurun boxes cannot be compared): each variant = a set of ops switches, its own model + HIP graph.
usage: ab_step.py [--rounds 4] [--steps 15] variant[,variant...]   variants: base, noside, noxcd, ..."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eventpretrain_amd import ops  # noqa: E402
from eventpretrain_amd.engine import GraphedStep  # noqa: E402
from eventpretrain_amd.model.pretrain import pr_hub_model as hub  # noqa: E402
from eventpretrain_amd.optim import FusedAdamW  # noqa: E402
from eventpretrain_amd.testing import make_args  # noqa: E402
from eventpretrain_amd.utils import lr_decay as lrd  # noqa: E402

VARIANTS = {
    "base": {},
    "noside": {"set_grad_side": False},
    "noxcd": {"set_wgrad_xcd_order": False},
    "nodefer": {"set_deferred_grads": False},
    "nog4": {"set_wgrad_g4": False},
    "g4fwd": {"_variant": 11},              # wide forward / data-gradient GEMMs on the G4 bodies instead of 128x128 tiles
    "g4fwd256": {"_variant": 12},           # only the one-round 256x256 forward tiles (encoder qkv)
    "g4dgrad": {"_variant": 13},            # only the 128x256 data-gradient tiles (fc2 data gradient with GELU')
    "nowt16": {"_cwt": 18},                 # bf16 epilogue in 4-column pieces (8-byte write-through stores) instead of 8-column ones
    "libplain": {"_vendor": True},             # yardstick only: the plain bf16 GEMMs (no epilogue operand) through torch.mm = hipBLASLt
    "no96": {"_no96": True},                  # 128x128 tiles where the launcher would pick 96x128 (257..384 tiles)
    "split2": {"_split": 2},                 # two half-batches on two streams inside the captured step (kernels of one half fill the other's tails)
    "split4": {"_split": 4},
    "split2s100": {"_split": 2, "_skew_us": 100},   # the second half starts (forward and backward) that much later: a GEMM of one half meets
    "split2s200": {"_split": 2, "_skew_us": 200},   # a LayerNorm / attention kernel of the other instead of its own twin
    "split2s400": {"_split": 2, "_skew_us": 400},
    "noguard": {"_guard_tables": False},     # host free to run ahead (the scalar-table race the guard closes)
}


def apply(cfg):
    defaults = {"set_grad_side": True, "set_wgrad_xcd_order": True, "set_deferred_grads": True, "set_wgrad_g4": True}
    for k, v in {**defaults, **cfg}.items():
        if hasattr(ops, k):
            getattr(ops, k)(v)


_orig_gemm = ops.gemm


def _gemm_no96(a, b, out, *, M, N, K, tile=0, trans_a=False, **kw):
    if tile == 0 and not trans_a and a.dtype == torch.bfloat16:
        t128 = ((M + 127) // 128) * ((N + 127) // 128)
        t96 = ((M + 95) // 96) * ((N + 127) // 128)
        if M >= 128 and N >= 128 and 256 < t128 and t96 <= 512 and kw.get("batch", (1, 1)) == (1, 1):
            tile = 1
    return _orig_gemm(a, b, out, M=M, N=N, K=K, tile=tile, trans_a=trans_a, **kw)


_lib_calls = [0, 0]


def _gemm_lib(a, b, out, *, M, N, K, trans_a=False, trans_b=False, lda=None, ldb=None, ldc=None, bias=None, act=0, aux=None, residual=None,
              alpha=1.0, accumulate=False, batch=(1, 1), tile=0, **kw):
    """torch.mm for the launches that are plain GEMMs on whole contiguous tensors (no bias / activation / residual / batch / offsets)."""
    plain = (a.dtype == torch.bfloat16 and out.dtype == torch.bfloat16 and not trans_a and bias is None and act == 0 and aux is None and
             residual is None and alpha == 1.0 and not accumulate and batch == (1, 1) and not kw and a.dim() == 2 and b.dim() == 2 and
             out.dim() == 2 and a.is_contiguous() and b.is_contiguous() and out.is_contiguous() and a.shape == (M, K) and out.shape == (M, N) and
             lda in (None, K) and ldc in (None, N) and M >= 2048)
    if plain and not trans_b and b.shape == (N, K) and ldb in (None, K):
        _lib_calls[0] += 1
        return torch.mm(a, b.t(), out=out)
    if plain and trans_b and b.shape == (K, N) and ldb in (None, N):
        _lib_calls[0] += 1
        return torch.mm(a, b, out=out)
    _lib_calls[1] += 1
    return _orig_gemm(a, b, out, M=M, N=N, K=K, trans_a=trans_a, trans_b=trans_b, lda=lda, ldb=ldb, ldc=ldc, bias=bias, act=act, aux=aux,
                      residual=residual, alpha=alpha, accumulate=accumulate, batch=batch, tile=tile, **kw)


def build(B, cfg):
    from eventpretrain_amd._lib import call
    ops.gemm = _gemm_no96 if cfg.get("_no96") else (_gemm_lib if cfg.get("_vendor") else _orig_gemm)
    apply(cfg)
    call("evp_gemm_set_variant", cfg.get("_variant", 10))      # the routing is decided at launch time, i.e. baked in at capture
    call("evp_gemm_set_variant", cfg.get("_cwt", 19))
    a = make_args(model_size="base", pr_phase="rec", device="cuda", batch_size=B)
    torch.manual_seed(1)
    m = hub.pretrain_hub_model_base_patch16(a, emb_frames_dim=512, queue_length=1024, T=0.07).cuda().train()
    opt = FusedAdamW(lrd.param_groups_lrd(a, m, 0.05, layer_decay=1), lr=1e-4, betas=(0.9, 0.95))
    x = torch.randn(B, 5, 224, 224, device="cuda") * 0.5
    y = torch.randn(B, 1, 224, 224, device="cuda")
    fwd = lambda mm, xx, yy, noise: mm(xx, yy, is_rec=True, noise=noise)
    n_split = cfg.get("_split", 1)
    if n_split > 1:
        sides = [torch.cuda.Stream() for _ in range(n_split - 1)]
        ops._deferred.join_streams = sides
        skew_cycles = int(cfg.get("_skew_us", 0) * 2100)          # torch.cuda._sleep counts shader cycles (~2.1 GHz)

        class Delay(torch.autograd.Function):                     # identity; its backward runs on the stream of its forward and spins first
            @staticmethod
            def forward(ctx, t):
                return t.view_as(t)

            @staticmethod
            def backward(ctx, g):
                if skew_cycles:
                    torch.cuda._sleep(skew_cycles)
                return g

        def fwd(mm, xx, yy, noise):
            cur = torch.cuda.current_stream()
            h = xx.shape[0] // n_split
            losses = []
            for s_ in sides:
                s_.wait_stream(cur)           # fork BEFORE anything of this step is queued on the main stream
            for i in range(n_split):
                sl = slice(i * h, (i + 1) * h)
                if i == 0:
                    losses.append(mm(xx[sl], yy[sl], is_rec=True, noise=noise[sl])[0])
                else:
                    with torch.cuda.stream(sides[i - 1]):
                        if skew_cycles:
                            torch.cuda._sleep(skew_cycles * i)
                        losses.append(Delay.apply(mm(xx[sl], yy[sl], is_rec=True, noise=noise[sl])[0]))
            for s_ in sides:
                cur.wait_stream(s_)
            tot = losses[0]
            for l_ in losses[1:]:
                tot = tot + l_
            return (tot / n_split,)
    ex = GraphedStep(m, opt, fwd, [x, y], noise_shape=(B, 196), generator=torch.Generator(device="cuda").manual_seed(1), warmup=3)
    assert ex.note.startswith("hip-graph"), ex.note
    ex.guard_tables = cfg.get("_guard_tables", True)
    ops.gemm = _orig_gemm
    ops._deferred.join_streams = []
    apply({})
    call("evp_gemm_set_variant", 10)
    call("evp_gemm_set_variant", 19)
    return ex


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--steps", type=int, default=15)
    ns = ap.parse_args()
    ops.set_compute_dtype(torch.bfloat16)
    names = ns.variants.split(",")
    exs = {n: build(ns.batch, VARIANTS[n]) for n in names}
    times = {n: [] for n in names}
    for r in range(ns.rounds):
        for n in names:
            ex = exs[n]
            for _ in range(3):
                ex.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(ns.steps):
                loss = ex.step()
            torch.cuda.synchronize()
            times[n].append((time.perf_counter() - t0) / ns.steps * 1e3)
            if not torch.isfinite(ex.loss).all():
                print(f"NON-FINITE loss in variant {n}, round {r}", flush=True)
    if _lib_calls[0]:
        print(f"libplain: {_lib_calls[0]} launches through torch.mm, {_lib_calls[1]} through evp_gemm (all eager + capture passes)")
    for n in names:
        t = sorted(times[n])
        print(f"{n:10s} min {t[0]:.3f} ms  median {t[len(t) // 2]:.3f} ms  all {[round(v, 3) for v in times[n]]}  loss {exs[n].loss.item():.4f}", flush=True)


if __name__ == "__main__":
    main()
