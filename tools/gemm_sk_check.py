#!/usr/bin/env python3
"""Stream-K (tile 12) against the data-parallel 128x128 kernel (tile 1) on the step's GEMM shapes: max difference,
run-to-run identity, the workspace's flag page after the run, and HIP-event timings of both."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eventpretrain_amd import ops
from eventpretrain_amd._lib import ACT_DGELU, ACT_GELU, ACT_NONE

def bench(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

T = torch.bfloat16
Me, Md = 64 * 98, 64 * 196
shapes = [("enc.qkv", Me, 2304, 768), ("enc.proj", Me, 768, 768), ("enc.fc1", Me, 3072, 768), ("enc.fc2", Me, 768, 3072),
          ("dec.qkv", Md, 1536, 512), ("dec.proj", Md, 512, 512), ("dec.fc1", Md, 2048, 512), ("dec.fc2", Md, 512, 2048),
          ("patch", Me, 768, 1280), ("dec.embed", Me, 512, 768), ("ragged", 2000, 520, 2048)]
only = sys.argv[1] if len(sys.argv) > 1 else ""
tot = {1: 0.0, 12: 0.0}
for name, M, N, K in shapes:
    if only and only not in name:
        continue
    x = torch.randn(M, K, device="cuda").to(T)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(T)
    dy = torch.randn(M, N, device="cuda").to(T)
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda")
    cases = [("fwd", lambda o, t: ops.gemm(x, w, o, M=M, N=N, K=K, bias=bias, tile=t), (M, N), T),
             ("fwd+res", lambda o, t: ops.gemm(x, w, o, M=M, N=N, K=K, bias=bias, residual=res, tile=t), (M, N), torch.float32),
             ("dgrad", lambda o, t: ops.gemm(dy, w, o, M=M, N=K, K=N, trans_b=True, ldb=K, tile=t), (M, K), torch.float32)]
    line = "%-9s %6dx%5dx%5d " % (name, M, N, K)
    for cname, fn, shp, odt in cases:
        o1, o2, o3 = (torch.empty(*shp, dtype=odt, device="cuda") for _ in range(3))
        fn(o1, 1); fn(o2, 12); fn(o3, 12)
        torch.cuda.synchronize()
        d = (o1.float() - o2.float()).abs().max().item() / max(o1.float().abs().max().item(), 1e-9)
        same = torch.equal(o2, o3)
        t1, t12 = bench(lambda: fn(o1, 1)), bench(lambda: fn(o2, 12))
        if name != "ragged":
            tot[1] += t1; tot[12] += t12
        line += " %s: rel %.1e %s dp %5.1fus sk %5.1fus |" % (cname, d, "same" if same else "DIFFERS", t1, t12)
    print(line, flush=True)
ws = next(iter(ops._sk_ws.values()))
flags = ws[:4096].view(torch.int32)
print("flag page after the run: nonzero entries = %d, error word = %d" % (int((flags != 0).sum()), int(flags[1023])))
print("sum of timings: dp %.1f us, sk %.1f us" % (tot[1], tot[12]))
