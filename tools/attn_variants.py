#!/usr/bin/env python3
"""Launch forms of the fused attention kernels on the step's two shapes (evp_attention_set_variant): one workgroup per head against
persistent grids with / without start skew and with the next head's loads in flight. Two timings per form: re-launched back to back
(operands warm in L2 / MALL) and with a 600 MB buffer overwritten between launches (operands cold, closer to the step)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from eventpretrain_amd import ops  # noqa: E402
from eventpretrain_amd._lib import call  # noqa: E402


def bench(fn, reps=20, warm=3, flush=None):
    for _ in range(warm):
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


forms = [("base", 0, 512, 0)]
for grid in (256, 384, 512, 768, 1024):
    for skew in (0, 200, 400):
        forms.append((f"persist g{grid} s{skew}", 1, grid, skew))
for grid in (256, 512):
    forms.append((f"prefetch g{grid}", 2, grid, 0))
    forms.append((f"prefetch g{grid} s300", 2, grid, 300))

flush = torch.empty(150_000_000, device="cuda")
for name, B, N, h, dh in [("enc", 64, 98, 12, 64), ("dec", 64, 196, 16, 32)]:
    C = h * dh
    qkv = (torch.randn(B * N, 3 * C, device="cuda") * 0.8).bfloat16()
    dout = torch.randn(B * N, C, device="cuda").bfloat16()
    call("evp_attention_set_variant", 0, 0, 512, 0)
    call("evp_attention_set_variant", 1, 0, 512, 0)
    out0, lse0, _ = ops.attention_fused_fwd(qkv, B, N, h, dh)
    d0 = ops.attention_fused_bwd(qkv, out0, dout, lse0, B, N, h, dh)
    for label, mode, grid, skew in forms:
        if grid > B * h:
            continue
        call("evp_attention_set_variant", 0, mode, grid, skew)
        call("evp_attention_set_variant", 1, mode, grid, skew)
        out, lse, _ = ops.attention_fused_fwd(qkv, B, N, h, dh)
        dq = ops.attention_fused_bwd(qkv, out0, dout, lse0, B, N, h, dh)
        same = torch.equal(out, out0) and torch.equal(lse, lse0) and torch.equal(dq, d0)
        tf = bench(lambda: ops.attention_fused_fwd(qkv, B, N, h, dh))
        tb = bench(lambda: ops.attention_fused_bwd(qkv, out0, dout, lse0, B, N, h, dh))
        tfc = bench(lambda: ops.attention_fused_fwd(qkv, B, N, h, dh), reps=10, flush=flush)
        tbc = bench(lambda: ops.attention_fused_bwd(qkv, out0, dout, lse0, B, N, h, dh), reps=10, flush=flush)
        print(f"{name} {label:24s} fwd {tf:6.1f} us (cold {tfc:6.1f})  bwd {tb:6.1f} us (cold {tbc:6.1f})  identical {same}", flush=True)
call("evp_attention_set_variant", 0, 0, 512, 0)
call("evp_attention_set_variant", 1, 0, 512, 0)
