#!/usr/bin/env python3
"""Run-to-run bitwise determinism of the bf16 kernels and of a full step (activations must be bit-identical; only
f32-atomic reductions -- bias-gradient column sums, split-K -- may differ in the last bits)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eventpretrain_amd import ops
from eventpretrain_amd._lib import ACT_GELU, ACT_DGELU
T = torch.bfloat16
g = torch.Generator().manual_seed(0)
M, N, K = 6272, 768, 768
x = torch.randn(M, K, generator=g).to(T).cuda(); w = (torch.randn(N, K, generator=g) * 0.05).to(T).cuda()
dy = torch.randn(M, N, generator=g).to(T).cuda(); bias = torch.randn(N, generator=g).cuda()
def same(f, name, n=20):
    ref = f().clone()
    bad = sum(int(not torch.equal(f(), ref)) for _ in range(n))
    print(f"{name:28s} nondeterministic runs: {bad}/{n}")
y = torch.empty(M, N, dtype=T).cuda(); dx = torch.empty(M, K, dtype=T).cuda(); dw = torch.empty(N, K).cuda(); aux = torch.empty(M, N, dtype=T).cuda()
same(lambda: ops.gemm(x, w, y, M=M, N=N, K=K, bias=bias), "gemm NT")
same(lambda: ops.gemm(x, w, y, M=M, N=N, K=K, bias=bias, act=ACT_GELU, aux=aux), "gemm NT gelu")
same(lambda: ops.gemm(dy, w, dx, M=M, N=K, K=N, trans_b=True, ldb=K), "gemm NN")
same(lambda: ops.gemm(dy, x, dw, M=N, N=K, K=M, trans_a=True, trans_b=True, lda=N, ldb=K, splitk=1), "gemm TN splitk=1")
same(lambda: ops.gemm(dy, x, dw, M=N, N=K, K=M, trans_a=True, trans_b=True, lda=N, ldb=K, splitk=4), "gemm TN splitk=4 (atomics)")
B, Nt, h, dh = 64, 98, 12, 64
qkv = (torch.randn(B * Nt, 3 * h * dh, generator=g) * 0.8).to(T).cuda(); do = torch.randn(B * Nt, h * dh, generator=g).to(T).cuda()
out, lse, _ = ops.attention_fused_fwd(qkv, B, Nt, h, dh)
same(lambda: ops.attention_fused_fwd(qkv, B, Nt, h, dh)[0], "attention fwd")
same(lambda: ops.attention_fused_bwd(qkv, out, do, lse, B, Nt, h, dh), "attention bwd")
xf = torch.randn(M, K, generator=g).cuda(); gam = torch.ones(K).cuda(); bet = torch.zeros(K).cuda()
yl, mean, rstd = ops.layernorm_fwd(xf, gam, bet, 1e-6, T)
same(lambda: ops.layernorm_fwd(xf, gam, bet, 1e-6, T)[0], "layernorm fwd")
same(lambda: ops.layernorm_bwd(dy, xf, gam, mean, rstd, gres=xf)[0], "layernorm bwd dx")
same(lambda: ops.layernorm_bwd(dy, xf, gam, mean, rstd, gres=xf)[2], "layernorm bwd dgamma")
same(lambda: ops.colsum(dy), "colsum (atomics)")
# full step
from eventpretrain_amd.model.pretrain import pr_hub_model as hub
from eventpretrain_amd.testing import det_fill_module_, det_normalish, det_uniform, make_args
a = make_args(model_size="small", pr_phase="rec", device="cuda")
m = hub.pretrain_hub_model_small_patch16(a, emb_frames_dim=512, queue_length=8, T=0.07); det_fill_module_(m); m = m.cuda().train()
xx = (det_normalish("d.v", (4, 5, 224, 224)) * 0.5).cuda(); yy = det_normalish("d.s", (4, 1, 224, 224)).cuda(); nz = det_uniform("d.n", (4, 196), 0, 1).cuda()
ops.set_compute_dtype(T)
res = []
for it in range(3):
    for p in m.parameters(): p.grad = None
    o = m(xx, yy, is_rec=True, noise=nz); o[0].backward(); torch.cuda.synchronize()
    res.append((o[0].item(), o[4].clone(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
print("loss", [r[0] for r in res], "pred bitwise equal", torch.equal(res[0][1], res[1][1]), torch.equal(res[0][1], res[2][1]))
diff = [(n, (res[0][2][n] - res[1][2][n]).abs().max().item() / (res[0][2][n].abs().max().item() + 1e-12)) for n in res[0][2]]
diff.sort(key=lambda t: -t[1])
print("largest run-to-run relative grad differences:", diff[:6])
