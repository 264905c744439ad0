#!/usr/bin/env python3
"""What does a plain streaming kernel reach at the sizes of the step's row kernels? (the practical floor for LayerNorm forward /
backward: launch ramp + tail on 30-120 MB of traffic). ATen add / copy on f32 tensors of the residual stream's shapes, warm and with
a 600 MB buffer rewritten between launches."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eventpretrain_amd import ops  # noqa: E402


def bench(fn, reps=30, flush=None):
    for _ in range(3):
        fn()
    if flush is None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    tot = 0.0
    for _ in range(reps):
        flush.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps * 1e3


flush = torch.empty(150_000_000, device="cuda")
for name, M, D in [("enc", 6272, 768), ("dec", 12544, 512)]:
    x, y = torch.randn(M, D, device="cuda"), torch.randn(M, D, device="cuda")
    o = torch.empty_like(x)
    z = torch.randn(M, D, device="cuda")
    nb = M * D * 4
    t = bench(lambda: torch.add(x, y, out=o))
    tc = bench(lambda: torch.add(x, y, out=o), reps=10, flush=flush)
    print(f"{name} add      (2 reads + 1 write, {3 * nb / 1e6:5.1f} MB): {t:6.1f} us = {3 * nb / t / 1e6:5.2f} TB/s   cold {tc:6.1f} us")
    t = bench(lambda: o.copy_(x))
    print(f"{name} copy     (1 read + 1 write,  {2 * nb / 1e6:5.1f} MB): {t:6.1f} us = {2 * nb / t / 1e6:5.2f} TB/s")
    t = bench(lambda: torch.addcmul(z, x, y, out=o))
    print(f"{name} addcmul  (3 reads + 1 write, {4 * nb / 1e6:5.1f} MB): {t:6.1f} us = {4 * nb / t / 1e6:5.2f} TB/s")
    # the LayerNorm kernels of the step on the same rows
    g, b = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
    ops.set_compute_dtype(torch.bfloat16)
    t = bench(lambda: ops.layernorm_fwd(x, g, b, 1e-6, torch.bfloat16))
    print(f"{name} LN fwd   (f32 in, bf16 out,  {nb * 1.5 / 1e6:5.1f} MB): {t:6.1f} us = {nb * 1.5 / t / 1e6:5.2f} TB/s")
    ln, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6, torch.bfloat16)
    dy = torch.randn(M, D, device="cuda").bfloat16()
    gp, bp = torch.nn.Parameter(g.clone()), torch.nn.Parameter(b.clone())

    def bwd():
        ops.layernorm_bwd(dy, x, gp, mean, rstd, gres=y, want_lp=True, params=(gp, bp), side=True)      # the step's form (three partial rows)
        ops._deferred.b.clear()                                                                            # (its queued column sums are not part of this probe)
    t = bench(bwd)
    byt = nb * (0.5 + 1 + 1 + 1 + 0.5)
    print(f"{name} LN bwd   (dy bf16, x, gres f32 in; dx f32 + bf16 out, {byt / 1e6:5.1f} MB): {t:6.1f} us = {byt / t / 1e6:5.2f} TB/s")
