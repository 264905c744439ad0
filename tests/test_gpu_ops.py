"""Row / column / token kernels and the attention core on the GPU against torch-CPU float64 or the oracle.
f32 kernels: 1e-5-class tolerances; bf16 I/O variants: 1e-2-class."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _g(seed):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("D", [192, 512, 768, 4096, 128])
def test_layernorm_fwd_bwd(D):
    from eventpretrain_amd import ops
    g = _g(D)
    M = 77
    xs = [torch.randn(M, D, generator=g) for _ in range(3)]
    gamma, beta = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g) * 0.1
    dy, gres = torch.randn(M, D, generator=g), torch.randn(M, D, generator=g)
    for n_in in (1, 3):
        xd = [t.double().requires_grad_(True) for t in xs[:n_in]]
        gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
        ref = F.layer_norm(sum(xd), (D,), gd, bd, 1e-6)
        ref.backward(dy.double())
        cu = [t.cuda() for t in xs[:n_in]] + [None] * (3 - n_in)
        y, mean, rstd = ops.layernorm_fwd(cu[0], gamma.cuda(), beta.cuda(), 1e-6, torch.float32, cu[1], cu[2])
        assert torch.allclose(y.cpu().double(), ref.detach(), atol=2e-5, rtol=1e-5)
        dx, dx_lp, dg, db = ops.layernorm_bwd(dy.cuda(), cu[0], gamma.cuda(), mean, rstd, gres=gres.cuda(), x2=cu[1], x3=cu[2], want_lp=True)
        assert torch.allclose(dx.cpu().double(), xd[0].grad + gres.double(), atol=5e-5, rtol=1e-4)
        assert torch.allclose(dx_lp.float().cpu(), dx.cpu(), atol=2e-2, rtol=1e-2)
        assert torch.allclose(dg.cpu().double(), gd.grad, atol=2e-4, rtol=1e-4)
        assert torch.allclose(db.cpu().double(), bd.grad, atol=2e-4, rtol=1e-4)
    # bf16 output / bf16 dy
    yb, _, _ = ops.layernorm_fwd(xs[0].cuda(), gamma.cuda(), beta.cuda(), 1e-6, torch.bfloat16)
    assert torch.allclose(yb.float().cpu(), F.layer_norm(xs[0], (D,), gamma, beta, 1e-6), atol=3e-2, rtol=2e-2)


def test_layernorm_many_rows_partials():
    """More rows than LN blocks (grid-stride) and the two-stage dgamma/dbeta reduction."""
    from eventpretrain_amd import ops
    g = _g(1)
    M, D = 6000, 256
    x, dy = torch.randn(M, D, generator=g), torch.randn(M, D, generator=g)
    gamma, beta = torch.ones(D), torch.zeros(D)
    y, mean, rstd = ops.layernorm_fwd(x.cuda(), gamma.cuda(), beta.cuda(), 1e-5, torch.float32)
    _, _, dg, db = ops.layernorm_bwd(dy.cuda(), x.cuda(), gamma.cuda(), mean, rstd)
    xh = F.layer_norm(x.double(), (D,))
    assert torch.allclose(dg.cpu().double(), (dy.double() * xh).sum(0), atol=2e-3, rtol=1e-4)
    assert torch.allclose(db.cpu().double(), dy.double().sum(0), atol=2e-3, rtol=1e-4)


def test_colsum_and_add_cast_transpose():
    from eventpretrain_amd import ops
    from eventpretrain_amd._lib import call, ptr, stream_ptr, dt
    g = _g(2)
    x = torch.randn(1000, 136, generator=g)
    assert torch.allclose(ops.colsum(x.cuda()).cpu().double(), x.double().sum(0), atol=1e-3)
    assert torch.allclose(ops.colsum(x.bfloat16().cuda()).cpu().double(), x.bfloat16().double().sum(0), atol=1e-2)
    a, b, c = [torch.randn(1003, generator=g) for _ in range(3)]
    assert torch.allclose(ops.add(a.cuda(), b.cuda(), c.cuda()).cpu(), a + b + c, atol=1e-6)
    assert torch.equal(ops.cast(a.cuda(), torch.bfloat16).cpu(), a.bfloat16())
    assert torch.equal(ops.cast(a.bfloat16().cuda(), torch.float32).cpu(), a.bfloat16().float())
    for dtp in (torch.float32, torch.bfloat16):
        m = torch.randn(130, 70, generator=g).to(dtp)
        out = torch.empty(70, 130, dtype=dtp).cuda()
        call("evp_transpose", ptr(m.cuda()), ptr(out), dt(m), 130, 70, stream_ptr())
        assert torch.equal(out.cpu(), m.t().contiguous())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N,h,dh", [(2, 16, 3, 64), (3, 98, 4, 64), (2, 196, 4, 32), (1, 49, 2, 32)])
def test_attention_core_fwd_bwd(dtype, B, N, h, dh):
    from eventpretrain_amd import ops
    g = _g(N)
    Cc = h * dh
    qkv = (torch.randn(B * N, 3 * Cc, generator=g) * 0.7).to(dtype)
    dout = torch.randn(B * N, Cc, generator=g).to(dtype)
    qd = qkv.float().double().requires_grad_(True)
    q, k, v = qd.view(B, N, 3, h, dh).permute(2, 0, 3, 1, 4)
    p = torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5, -1)
    o = (p @ v).transpose(1, 2).reshape(B * N, Cc)
    o.backward(dout.float().double())
    probs, out = ops.attention_fwd(qkv.cuda(), B, N, h, dh)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert torch.allclose(probs[..., :N].float().cpu().double(), p.detach(), atol=tol, rtol=tol)
    assert (probs[..., N:] == 0).all()
    assert torch.allclose(out.float().cpu().double(), o.detach(), atol=tol * 4, rtol=tol)
    dqkv = ops.attention_bwd(qkv.cuda(), probs, dout.cuda(), B, N, h, dh)
    err = (dqkv.float().cpu().double() - qd.grad).abs().max().item() / qd.grad.abs().max().item()
    assert err <= (1e-4 if dtype == torch.float32 else 3e-2), err


@pytest.mark.parametrize("B,N,h,dh", [(2, 16, 3, 64), (3, 98, 4, 64), (2, 196, 4, 32), (1, 49, 2, 32), (2, 224, 2, 64), (2, 1, 1, 32),
                                      (64, 98, 12, 64)])
def test_fused_attention_fwd_bwd(B, N, h, dh):
    """Fused per-head kernels (bf16) against float64 attention on the bf16-rounded inputs, and against the unfused
    HIP formulation."""
    from eventpretrain_amd import ops
    g = _g(1000 + N)
    Cc = h * dh
    qkv = (torch.randn(B * N, 3 * Cc, generator=g) * 0.8).bfloat16()
    dout = torch.randn(B * N, Cc, generator=g).bfloat16()
    qd = qkv.float().double().requires_grad_(True)
    q, k, v = qd.view(B, N, 3, h, dh).permute(2, 0, 3, 1, 4)
    sc = q @ k.transpose(-1, -2) * dh ** -0.5
    p = torch.softmax(sc, -1)
    o = (p @ v).transpose(1, 2).reshape(B * N, Cc)
    o.backward(dout.float().double())
    assert ops.fused_attention_ok(torch.bfloat16, N, dh)
    out, lse, probs = ops.attention_fused_fwd(qkv.cuda(), B, N, h, dh, want_probs=True)
    assert torch.allclose(lse.cpu().double(), torch.logsumexp(sc, -1).detach(), atol=1e-3, rtol=1e-4)
    assert torch.allclose(probs[..., :N].float().cpu().double(), p.detach(), atol=1e-2, rtol=2e-2)
    assert (probs[..., N:] == 0).all()
    assert torch.allclose(out.float().cpu().double(), o.detach(), atol=3e-2, rtol=3e-2)
    # the backward takes the forward's own (bf16) output, as the training step does
    dqkv = ops.attention_fused_bwd(qkv.cuda(), out, dout.cuda(), lse, B, N, h, dh)
    ref = qd.grad
    err = (dqkv.float().cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    assert err <= 3e-2, err
    # the three slices separately (a swapped dK/dV or a transposed tile would pass a max-norm check of the whole)
    got = dqkv.float().cpu().double().view(B, N, 3, h, dh)
    want = ref.view(B, N, 3, h, dh)
    for i in range(3):
        e = (got[:, :, i] - want[:, :, i]).abs().max().item() / max(want[:, :, i].abs().max().item(), 1e-6)
        assert e <= 4e-2 or N == 1, (i, e)


def test_patchify_embed_post_unshuffle_loss():
    from eventpretrain_amd import ops
    from oracle import model_oracle as mo
    g = _g(7)
    B, Cc, H, W, p, D = 3, 5, 64, 96, 16, 192
    L = (H // p) * (W // p)
    x = torch.randn(B, Cc, H, W, generator=g)
    noise = torch.rand(B, L, generator=g)
    ids_keep, mask, ids_restore = mo.masking_from_noise(noise, 0.5)
    n_keep = ids_keep.shape[1]
    w = (torch.randn(D, Cc, p, p, generator=g) * 0.05).requires_grad_(True)
    b = (torch.randn(D, generator=g) * 0.1).requires_grad_(True)
    gam = (torch.rand(D, generator=g) + 0.5).requires_grad_(True)
    bet = (torch.randn(D, generator=g) * 0.1).requires_grad_(True)
    pos = torch.randn(1, L, D, generator=g)
    sd = {"proj.weight": w, "proj.bias": b, "norm.weight": gam, "norm.bias": bet}
    ref = mo.patch_embed(sd, "", x, p) + pos
    ref = torch.gather(ref, 1, ids_keep.unsqueeze(-1).expand(-1, -1, D))
    gout = torch.randn(B, n_keep, D, generator=g)
    ref.backward(gout)
    cw, cb, cg, cbt = [t.detach().clone().cuda().requires_grad_(True) for t in (w, b, gam, bet)]
    out = ops.PatchEmbedFn.apply(x.cuda(), ids_keep.cuda(), cw, cb, cg, cbt, pos.cuda(), p)
    assert torch.allclose(out.cpu(), ref.detach(), atol=2e-5, rtol=1e-5)
    out.backward(gout.cuda())
    for a_, r_ in ((cw, w), (cb, b), (cg, gam), (cbt, bet)):
        assert torch.allclose(a_.grad.cpu(), r_.grad, atol=1e-4 * r_.grad.abs().max().item() + 1e-6, rtol=1e-4)

    # unshuffle
    Dd = 64
    emb = torch.randn(B, n_keep, Dd, generator=g).requires_grad_(True)
    mt = torch.randn(1, 1, Dd, generator=g).requires_grad_(True)
    posd = torch.randn(1, L, Dd, generator=g)
    cat = torch.cat([emb, mt.expand(B, L - n_keep, Dd)], 1)
    r = torch.gather(cat, 1, ids_restore.unsqueeze(-1).expand(-1, -1, Dd)) + posd
    gr = torch.randn(B, L, Dd, generator=g)
    r.backward(gr)
    ce, cm = emb.detach().clone().cuda().requires_grad_(True), mt.detach().clone().cuda().requires_grad_(True)
    o = ops.UnshuffleFn.apply(ce, cm, posd.cuda(), ids_restore.cuda())
    assert torch.equal(o.cpu(), r.detach())
    o.backward(gr.cuda())
    assert torch.allclose(ce.grad.cpu(), emb.grad, atol=1e-6) and torch.allclose(cm.grad.cpu(), mt.grad, atol=1e-4)

    # loss (norm_pix on/off, masked / unmasked mean), gradient w.r.t. pred, upstream scale
    P = p * p
    pred = torch.randn(B, L, P, generator=g).requires_grad_(True)
    tgt = torch.randn(B, 1, H, W, generator=g)
    for norm_pix, ratio in ((True, 0.5), (False, 0.5), (True, 0.0)):
        pred.grad = None
        lr_ = mo.rec_loss(pred, tgt, mask, p, norm_pix, ratio)
        (lr_ * 0.25).backward()
        cp = pred.detach().clone().cuda().requires_grad_(True)
        lc = ops.RecLossFn.apply(cp, tgt.cuda(), mask.cuda() if ratio else None, p, norm_pix)
        assert lc.dim() == 0 and abs(lc.item() - lr_.item()) <= 2e-6 * abs(lr_.item())
        (lc * 0.25).backward()
        assert torch.allclose(cp.grad.cpu(), pred.grad, atol=1e-8, rtol=1e-4)


def test_fused_adamw_matches_torch_adamw():
    from eventpretrain_amd.optim import FusedAdamW
    from eventpretrain_amd import ops
    g = _g(9)
    shapes = [(300, 70), (17,), (1, 1, 64), (40000,), (3, 5, 16, 16)]
    ps = [torch.randn(*s, generator=g) for s in shapes]
    ref = [torch.nn.Parameter(p.clone().double()) for p in ps]
    mine = [torch.nn.Parameter(p.clone().cuda()) for p in ps]
    ro = torch.optim.AdamW([{"params": ref[:2], "weight_decay": 0.05}, {"params": ref[2:], "weight_decay": 0.0}], lr=1e-3, betas=(0.9, 0.95))
    mo_ = FusedAdamW([{"params": mine[:2], "weight_decay": 0.05}, {"params": mine[2:], "weight_decay": 0.0}], lr=1e-3, betas=(0.9, 0.95))
    ops.set_compute_dtype(torch.bfloat16)
    try:
        sh = ops.lp_weight(mine[0])
    finally:
        ops.set_compute_dtype(torch.float32)
    for step in range(4):
        lr = 1e-3 * (step + 1)
        for o in (ro, mo_):
            for grp in o.param_groups:
                grp["lr"] = lr
        for r, m in zip(ref, mine):
            gr = torch.randn(r.shape, generator=g)
            r.grad, m.grad = gr.double(), gr.cuda()
        n = mo_.grad_norm().item()
        assert abs(n - math.sqrt(sum((r.grad ** 2).sum().item() for r in ref))) <= 1e-4 * n
        ro.step()
        mo_.step()
    for r, m in zip(ref, mine):
        assert torch.allclose(m.detach().cpu().double(), r.detach(), atol=2e-6, rtol=1e-5)
    assert torch.equal(sh.cpu(), mine[0].detach().cpu().bfloat16()), "bf16 shadow not refreshed by the optimizer step"


# ------------------------------------------------------------------------------------------------ depthwise 5x5 (ConvViT)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W,C,scale", [(3, 56, 56, 128, 4), (2, 28, 28, 192, 2), (2, 12, 20, 64, 4), (1, 8, 8, 64, 0)])
def test_dwconv5x5_band_and_row_kernels_vs_conv2d(dtype, B, H, W, C, scale):
    """ConvBlock's depthwise conv with the keep mask fused (conv_block.py:41-51): forward, data gradient, weight and bias gradient
    of the LDS-band kernels and of the register row walkers against torch's conv2d in float64; the two kernel forms agree with
    each other to summation order."""
    import torch.nn.functional as F
    from eventpretrain_amd._lib import call, dt, ptr, stream_ptr
    g = torch.Generator().manual_seed(B * 100 + W)
    x = torch.randn(B, H, W, C, generator=g)
    dy = torch.randn(B, H, W, C, generator=g)
    if dtype == torch.bfloat16:
        x, dy = x.bfloat16().float(), dy.bfloat16().float()
    w = torch.randn(C, 1, 5, 5, generator=g) * 0.2
    bias = torch.randn(C, generator=g)
    mask = None
    keep = torch.ones(B, 1, H, W, dtype=torch.float64)
    if scale:
        mask = (torch.rand(B, (H // scale) * (W // scale), generator=g) < 0.5).float()
        keep = (1 - mask.double()).view(B, 1, H // scale, W // scale).repeat_interleave(scale, 2).repeat_interleave(scale, 3)
    xr = x.double().permute(0, 3, 1, 2).clone().requires_grad_(True)
    wr = w.double().clone().requires_grad_(True)
    br = bias.double().clone().requires_grad_(True)
    yr = F.conv2d(xr * keep, wr, br, padding=2, groups=C)
    yr.backward(dy.double().permute(0, 3, 1, 2))
    want = (yr.detach().permute(0, 2, 3, 1), xr.grad.permute(0, 2, 3, 1), wr.grad.view(C, 25), br.grad)

    xd, dyd, wd, bd = x.to(dtype).cuda(), dy.to(dtype).cuda(), w.view(C, 25).cuda().contiguous(), bias.cuda()
    md = mask.cuda() if mask is not None else None
    got = {}
    try:
        for band in (1, 0):
            call("evp_dwconv_set_band", band)
            y = torch.full((B, H, W, C), float("nan"), dtype=dtype, device="cuda")
            call("evp_dwconv5x5_fwd", ptr(xd), dt(xd), ptr(md), int(scale), ptr(wd), ptr(bd), B, H, W, C, ptr(y), stream_ptr())
            ns = call("evp_dwconv5x5_bwd_nslab", B, H, W)
            ws = torch.full((ns * 26 * C,), float("nan"), device="cuda")
            dx = torch.full((B, H, W, C), float("nan"), dtype=dtype, device="cuda")
            dw = torch.full((C, 25), float("nan"), device="cuda")
            db = torch.full((C,), float("nan"), device="cuda")
            call("evp_dwconv5x5_bwd", ptr(dyd), ptr(xd), dt(xd), ptr(md), int(scale), ptr(wd), B, H, W, C, ptr(dx), ptr(dw), ptr(db), ptr(ws),
                 stream_ptr())
            torch.cuda.synchronize()
            got[band] = (y.float().cpu(), dx.float().cpu(), dw.cpu(), db.cpu())
    finally:
        call("evp_dwconv_set_band", 1)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    for band in (1, 0):
        for t, r, name in zip(got[band], want, ("y", "dx", "dw", "db")):
            scale_ = max(1.0, r.abs().max().item())
            assert torch.allclose(t.double(), r, atol=tol * scale_, rtol=tol), (band, name, (t.double() - r).abs().max().item())
    for a, b_, name in zip(got[1], got[0], ("y", "dx", "dw", "db")):
        assert torch.allclose(a, b_, atol=1e-5 * max(1.0, b_.abs().max().item()), rtol=1e-5 if dtype == torch.float32 else 1e-2), name
