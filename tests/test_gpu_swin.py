"""Swin path on the GPU: the window-attention / gather kernels through the C-ABI against the
oracle's formulation, and the drop-in Swin-T hub against the fixture the reference itself produced."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import assert_checksums, jl, rec_inputs

pytestmark = pytest.mark.gpu

F32_LOSS_RTOL = 1e-4
BF16_LOSS_RTOL = 2e-2


def _rand_rel(rng, nG, N, R, frac_blocked):
    rel = rng.integers(0, R, size=(nG, N, N)).astype(np.int32)
    blocked = rng.random((nG, N, N)) < frac_blocked
    blocked[:, np.arange(N), np.arange(N)] = False
    rel[blocked] = -1
    return rel


def _ref_window_attention(qkv, table, rel, nG, heads):
    """oracle formulation (model_oracle.window_attention core) in float64 on the CPU."""
    Bg, N, _, H, dh = qkv.shape
    q, k, v = qkv.double().permute(2, 0, 3, 1, 4)
    s = (q * dh ** -0.5) @ k.transpose(-2, -1)
    relt = torch.from_numpy(rel).long()
    bias = torch.where(relt >= 0, table.double()[relt.clamp_min(0)].permute(3, 0, 1, 2), torch.full((), -100.0, dtype=torch.float64))
    s = s.view(Bg // nG, nG, H, N, N) + bias.permute(1, 0, 2, 3).unsqueeze(0)
    p = torch.softmax(s.view(Bg, H, N, N), -1)
    return (p @ v).transpose(1, 2).reshape(Bg, N, H * dh), p


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Bg,nG,N,H", [(6, 3, 49, 3), (2, 1, 96, 12), (4, 2, 24, 24), (3, 3, 1, 6), (2, 1, 128, 3)])
def test_window_attention_fwd_bwd(dtype, Bg, nG, N, H):
    from eventpretrain_amd._lib import call, dt, ptr, stream_ptr
    rng = np.random.default_rng(Bg * 1000 + N)
    R, dh = 169, 32
    qkv = torch.from_numpy(rng.standard_normal((Bg, N, 3, H, dh)).astype(np.float32))
    table = torch.from_numpy((rng.standard_normal((R, H)) * 0.5).astype(np.float32))
    rel = _rand_rel(rng, nG, N, R, 0.4)
    dout = torch.from_numpy(rng.standard_normal((Bg, N, H * dh)).astype(np.float32))
    if dtype == torch.bfloat16:
        qkv, dout = qkv.bfloat16().float(), dout.bfloat16().float()
    qr = qkv.clone().double().requires_grad_(True)
    tr = table.clone().double().requires_grad_(True)
    o_ref, p_ref = _ref_window_attention(qr, tr, rel, nG, H)
    (o_ref * dout.double()).sum().backward()

    qd, td, rd = qkv.to(dtype).cuda(), table.cuda(), torch.from_numpy(rel).cuda()
    out = torch.empty(Bg, N, H * dh, dtype=dtype, device="cuda")
    probs = torch.empty(Bg, H, N, N, dtype=torch.float32, device="cuda")
    call("evp_window_attention_fwd", ptr(qd), ptr(td), ptr(rd), ptr(out), ptr(probs), Bg, nG, N, H, R, dh ** -0.5, dt(qd), None, 1.0, stream_ptr())
    dqkv = torch.empty_like(qd)
    dtab = torch.full((R, H), 7.0, dtype=torch.float32, device="cuda")       # must be overwritten, not accumulated
    dd = dout.to(dtype).cuda()
    call("evp_window_attention_bwd", ptr(qd), ptr(td), ptr(rd), ptr(out), ptr(dd), ptr(dqkv), ptr(dtab), Bg, nG, N, H, R,
         dh ** -0.5, dt(qd), None, 1.0, stream_ptr())
    torch.cuda.synchronize()
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert torch.allclose(probs.cpu().double(), p_ref.detach(), atol=tol * 0.1 if dtype == torch.float32 else 1e-5, rtol=1e-4)
    assert torch.allclose(out.float().cpu().double(), o_ref.detach(), atol=tol, rtol=tol)
    gq = qr.grad
    assert torch.allclose(dqkv.float().cpu().double(), gq, atol=tol * max(1.0, gq.abs().max().item()), rtol=tol)
    gt = tr.grad
    assert torch.allclose(dtab.cpu().double(), gt, atol=tol * max(1.0, gt.abs().max().item()), rtol=tol)


@pytest.mark.parametrize("Bg,nG,N,H", [(6, 3, 49, 3), (8, 2, 49, 6), (2, 1, 96, 12), (4, 2, 24, 24), (3, 3, 1, 6), (2, 1, 128, 3), (82, 41, 49, 3)])
def test_window_attention_mfma_fwd_bwd(Bg, nG, N, H):
    """The bf16 MFMA form (evp_window_bias_build -> evp_window_attention_fused_fwd/bwd -> evp_window_bias_reduce) against the
    same float64 formulation as the f32 LDS kernels above, incl. fully masked (padding) groups and NaN-poisoned outputs."""
    from eventpretrain_amd._lib import call, ptr, stream_ptr
    rng = np.random.default_rng(Bg * 1000 + N + 7)
    R, dh = 169, 32
    qkv = torch.from_numpy(rng.standard_normal((Bg, N, 3, H, dh)).astype(np.float32)).bfloat16().float()
    table = torch.from_numpy((rng.standard_normal((R, H)) * 0.5).astype(np.float32))
    rel = _rand_rel(rng, nG, N, R, 0.4)
    if nG > 2:
        rel[-1] = -1                          # an empty (padding) group: every pair masked
    dout = torch.from_numpy(rng.standard_normal((Bg, N, H * dh)).astype(np.float32)).bfloat16().float()
    qr = qkv.clone().double().requires_grad_(True)
    tr = table.clone().double().requires_grad_(True)
    o_ref, _ = _ref_window_attention(qr, tr, rel, nG, H)
    (o_ref * dout.double()).sum().backward()

    qd, td, rd = qkv.bfloat16().cuda(), table.cuda(), torch.from_numpy(rel).cuda()
    NP = call("evp_window_attention_fused_np", N)
    nan = float("nan")
    addm = torch.full((nG * H * NP * NP,), nan, device="cuda")
    addmT = torch.full_like(addm, nan)
    call("evp_window_bias_build", ptr(td), ptr(rd), nG, N, H, R, ptr(addm), ptr(addmT), stream_ptr())
    a4 = addm.view(nG, H, NP, NP)
    assert torch.equal(a4.transpose(2, 3), addmT.view(nG, H, NP, NP)) and torch.isfinite(addm).all()
    out = torch.full((Bg, N, H * dh), nan, dtype=torch.bfloat16, device="cuda")
    lse = torch.full((Bg * H * N,), nan, device="cuda")
    call("evp_window_attention_fused_fwd", ptr(qd), ptr(addm), Bg, nG, N, H, dh ** -0.5, ptr(out), ptr(lse), stream_ptr())
    dqkv = torch.full_like(qd, nan)
    nchunk = call("evp_window_attention_fused_nchunk", Bg, nG, H)
    dA = torch.full((nchunk * addm.numel(),), nan, device="cuda")             # the N x N part of every plane is written
    dtab = torch.full((R, H), 7.0, dtype=torch.float32, device="cuda")       # overwritten, not accumulated
    dd = dout.bfloat16().cuda()
    call("evp_window_attention_fused_bwd", ptr(qd), ptr(out), ptr(dd), ptr(lse), ptr(addm), ptr(addmT), Bg, nG, N, H, dh ** -0.5,
         ptr(dqkv), ptr(dA), stream_ptr())
    call("evp_window_bias_reduce", ptr(dA), ptr(rd), Bg, nG, N, H, R, ptr(dtab), stream_ptr())
    torch.cuda.synchronize()
    tol = 2e-2
    assert torch.isfinite(out.float()).all() and torch.isfinite(dqkv.float()).all() and torch.isfinite(dtab).all()
    assert torch.allclose(out.float().cpu().double(), o_ref.detach(), atol=tol, rtol=tol)
    gq = qr.grad
    assert torch.allclose(dqkv.float().cpu().double(), gq, atol=tol * max(1.0, gq.abs().max().item()), rtol=tol)
    gt = tr.grad
    assert torch.allclose(dtab.cpu().double(), gt, atol=tol * max(1.0, gt.abs().max().item()), rtol=tol)


def test_swin_bf16_step_mfma_window_attention_matches_lds_kernels():
    """Whole Swin-T step in bf16: window attention on the MFMA kernels (default) against the f32 LDS kernels (the previous
    default): loss and gradient norms agree to bf16 accuracy."""
    from eventpretrain_amd import ops
    d = load_golden("rec_swin_tiny")
    x, y, noise = rec_inputs("swin", jl(d["cfg"]))
    ops.set_compute_dtype(torch.bfloat16)
    res = {}
    try:
        for mode in (True, False):
            ops.set_window_mfma(mode)
            a, m = _swin_hub()
            out = m(x.cuda(), y.cuda(), is_rec=True, noise=noise.cuda())
            out[0].backward()
            ops.flush_deferred_grads()
            torch.cuda.synchronize()
            res[mode] = (out[0].item(), {n: p.grad.double().norm().item() for n, p in m.named_parameters() if p.grad is not None})
    finally:
        ops.set_window_mfma(True)
        ops.set_compute_dtype(torch.float32)
    assert abs(res[True][0] - res[False][0]) <= 2e-3 * abs(res[False][0]), (res[True][0], res[False][0])
    assert abs(res[True][0] - float(d["loss"])) <= BF16_LOSS_RTOL * abs(float(d["loss"]))
    worst = max(abs(res[True][1][n] - v) / (v + 1e-6) for n, v in res[False][1].items())
    assert worst <= 6e-2, worst
    print(f"[swin bf16] MFMA vs LDS window attention: loss {res[True][0]:.6f} / {res[False][0]:.6f}, worst gradient-norm rel diff {worst:.2e}")


def test_window_attention_rejects_bad_shapes():
    from eventpretrain_amd import EvpError
    from eventpretrain_amd._lib import call, ptr, stream_ptr
    t = torch.zeros(16, device="cuda")
    i = torch.zeros(16, dtype=torch.int32, device="cuda")
    with pytest.raises(EvpError):
        call("evp_window_attention_fwd", ptr(t), ptr(t), ptr(i), ptr(t), None, 5, 2, 49, 3, 169, 1.0, 0, None, 1.0, stream_ptr())   # Bg % nG
    with pytest.raises(EvpError):
        call("evp_window_attention_fwd", ptr(t), ptr(t), ptr(i), ptr(t), None, 4, 2, 129, 3, 169, 1.0, 0, None, 1.0, stream_ptr())  # N > 128
    with pytest.raises(EvpError):
        call("evp_gather_rows_f32", ptr(t), ptr(i), ptr(t), 1, 4, 4, 3, 0, stream_ptr())                                  # C % 4


def test_gather_rows_and_adjoint():
    from eventpretrain_amd import ops
    rng = np.random.default_rng(5)
    B, n, C = 3, 37, 96
    x = torch.from_numpy(rng.standard_normal((B, n, C)).astype(np.float32)).cuda().requires_grad_(True)
    perm = rng.permutation(n)
    slots = np.full(n + 11, -1, dtype=np.int32)
    pos = np.sort(rng.choice(n + 11, n, replace=False))
    slots[pos] = perm                                   # slot -> token, -1 = padding
    fwd = np.where(slots < 0, 0, slots).astype(np.int32)       # the reference pads with token 0
    adj = np.empty(n, dtype=np.int32)
    adj[perm] = pos
    y = ops.GatherRowsFn.apply(x, torch.from_numpy(fwd).cuda(), torch.from_numpy(adj).cuda())
    assert torch.equal(y.detach().cpu(), x.detach().cpu()[:, torch.from_numpy(fwd).long()])
    g = torch.from_numpy(rng.standard_normal(tuple(y.shape)).astype(np.float32))
    g[:, torch.from_numpy(slots < 0)] = 0               # padding rows are dropped downstream: they carry no gradient
    y.backward(g.cuda())
    expect = torch.zeros(B, n, C)
    expect[:, torch.from_numpy(perm).long()] = g[:, torch.from_numpy(pos).long()]
    assert torch.equal(x.grad.cpu(), expect)


def test_fuse_conv_matches_dense_formulation():
    """ops.SwinFuseConvFn against the reference's formulation (zero grid, scatter, Conv2d, gather by ids_keep), f32."""
    from eventpretrain_amd import ops
    import torch.nn.functional as F
    rng = np.random.default_rng(9)
    B, g, k, C, Dout, K = 3, 7, 4, 8, 16, 24
    R = g * k
    vis_cells = np.zeros(g * g, dtype=bool)
    vis_cells[rng.choice(g * g, 25, replace=False)] = True
    vis = np.repeat(np.repeat(vis_cells.reshape(g, g), k, 0), k, 1)
    ys, xs = np.nonzero(vis)
    n = ys.shape[0]
    tokmap = np.full(R * R, -1, dtype=np.int32)
    tokmap[ys * R + xs] = np.arange(n)
    coords = np.stack([ys, xs], -1).astype(np.int32)
    noise = torch.from_numpy(rng.random((B, g * g)).astype(np.float32))
    order = torch.argsort(noise, dim=1, stable=True)
    ids_restore = torch.argsort(order, dim=1, stable=True)
    ids_keep = order[:, :K].contiguous()
    x = torch.from_numpy(rng.standard_normal((B, n, C)).astype(np.float32))
    w = torch.from_numpy((rng.standard_normal((Dout, C, k, k)) * 0.2).astype(np.float32))
    b = torch.from_numpy(rng.standard_normal(Dout).astype(np.float32))
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    dense = torch.zeros(B, R * R, C)
    dense[:, torch.from_numpy(ys * R + xs)] = xr
    d = F.conv2d(dense.view(B, R, R, C).permute(0, 3, 1, 2), wr, br, stride=k).flatten(2).transpose(1, 2)
    ref = torch.gather(d, 1, ids_keep.unsqueeze(-1).expand(-1, -1, Dout))
    go = torch.from_numpy(rng.standard_normal((B, K, Dout)).astype(np.float32))
    ref.backward(go)
    ops.set_compute_dtype(torch.float32)
    ops.set_deferred_grads(False)
    try:
        xg, wg, bg = (t.clone().cuda().requires_grad_(True) for t in (x, w, b))
        y = ops.SwinFuseConvFn.apply(xg, wg, bg, torch.from_numpy(tokmap).cuda(), torch.from_numpy(coords).cuda(), ids_keep.cuda(),
                                     ids_restore.cuda(), R, k)
        y.backward(go.cuda())
        torch.cuda.synchronize()
    finally:
        ops.set_deferred_grads(True)
    assert torch.allclose(y.detach().cpu(), ref.detach(), atol=2e-5, rtol=1e-5)
    assert torch.allclose(xg.grad.cpu(), xr.grad, atol=2e-5, rtol=1e-5)
    assert torch.allclose(wg.grad.cpu(), wr.grad, atol=5e-5, rtol=1e-5)
    assert torch.allclose(bg.grad.cpu(), br.grad, atol=5e-5, rtol=1e-5)


def _swin_hub(pr_phase="rec", **kw):
    from eventpretrain_amd.model.pretrain import pr_hub_model as hub
    from eventpretrain_amd.testing import det_fill_module_, make_args
    a = make_args(model_size="tiny", pr_phase=pr_phase, backbone_type="swin", device="cuda", **kw)
    m = hub.pretrain_hub_model_swin_tiny_patch16(a, emb_frames_dim=512, queue_length=4 if pr_phase != "rec" else 1024, T=0.07)
    det_fill_module_(m)
    return a, m.cuda().train()


def test_swin_state_dict_keys_match_reference():
    d = load_golden("rec_swin_tiny")
    _, m = _swin_hub()
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == jl(d["state_keys"])


def test_swin_rec_step_f32_matches_reference():
    """BASELINE config 5: Swin-T (window 7) masked reconstruction step, f32 mode, against the reference's own outputs:
    mask / ids_restore / coords bit-exact, loss within 1e-4 rel, stage embeddings, prediction, last-block attention,
    every parameter's gradient norm and the sampled full gradients (incl. relative-position tables)."""
    from eventpretrain_amd import ops
    d = load_golden("rec_swin_tiny")
    cfg = jl(d["cfg"])
    a, m = _swin_hub()
    x, y, noise = rec_inputs("swin", cfg)
    ops.set_compute_dtype(torch.float32)
    out = m(x.cuda(), y.cuda(), is_rec=True, noise=noise.cuda())
    (loss, l1, l2, l3, l4, lh, c1, c2, c3, c4, pred, mask, restore, attn) = out
    loss.backward()
    torch.cuda.synchronize()
    assert np.array_equal(mask.cpu().numpy(), d["mask"]) and np.array_equal(restore.cpu().numpy(), d["ids_restore"])
    for c, k in ((c1, "coords_l1"), (c2, "coords_l2"), (c3, "coords_l3"), (c4, "coords_l4")):
        assert c.dtype == torch.int64 and np.array_equal(c.cpu().numpy(), d[k]), k
    rel = abs(loss.item() - float(d["loss"])) / abs(float(d["loss"]))
    assert rel <= F32_LOSS_RTOL, rel
    assert list(attn.shape) == list(d["attn_shape"])
    for t, k in ((l1, "emb_l1"), (l2, "emb_l2"), (l3, "emb_l3"), (l4, "emb_l4"), (lh, "emb_lh"), (pred, "pred"), (attn, "attn")):
        assert_checksums(t.contiguous(), d[k + "_checksums"], 1e-4, k)
    assert torch.allclose(lh.cpu(), torch.from_numpy(d["emb_lh"]), atol=2e-4, rtol=1e-4)
    assert torch.allclose(l4.cpu(), torch.from_numpy(d["emb_l4"]), atol=2e-4, rtol=1e-4)
    params = dict(m.named_parameters())
    worst = 0.0
    for n, gn in zip(jl(d["grad_names"]), d["grad_norms"]):
        assert params[n].grad is not None, n
        e = abs(params[n].grad.double().norm().item() - gn) / (gn + 1e-9)
        worst = max(worst, e)
        assert e <= 3e-3, (n, e)
    for k in d.files:
        if k.startswith("grad::"):
            ref = torch.from_numpy(d[k])
            got = params[k[6:]].grad.cpu()
            assert torch.allclose(got, ref, atol=1e-6 + 2e-3 * ref.abs().max().item(), rtol=2e-3), k
    tot = math.sqrt(sum(p.grad.double().pow(2).sum().item() for p in m.parameters() if p.grad is not None))
    assert abs(tot - float(d["total_grad_norm"])) / float(d["total_grad_norm"]) <= 1e-3
    print(f"[swin-tiny] f32 loss rel err {rel:.2e}, worst grad-norm rel err {worst:.2e}")


def test_swin_rec_step_bf16_reported():
    from eventpretrain_amd import ops
    d = load_golden("rec_swin_tiny")
    a, m = _swin_hub()
    x, y, noise = rec_inputs("swin", jl(d["cfg"]))
    ops.set_compute_dtype(torch.bfloat16)
    try:
        out = m(x.cuda(), y.cuda(), is_rec=True, noise=noise.cuda())
        out[0].backward()
        torch.cuda.synchronize()
    finally:
        ops.set_compute_dtype(torch.float32)
    assert np.array_equal(out[11].cpu().numpy(), d["mask"])
    rel = abs(out[0].item() - float(d["loss"])) / abs(float(d["loss"]))
    print(f"[swin-tiny] bf16 loss rel err {rel:.2e}")
    assert rel <= BF16_LOSS_RTOL
    tot = math.sqrt(sum(p.grad.double().pow(2).sum().item() for p in m.parameters() if p.grad is not None))
    assert abs(tot - float(d["total_grad_norm"])) / float(d["total_grad_norm"]) <= 5e-2


def test_swin_plan_cache_and_second_pattern():
    """A second visibility pattern builds a second plan (different groups) and still steps; the first is served from the
    cache. Also exercises density masking (AvgPool 32x32 of |sum over bins|, swin.py:122-131)."""
    from eventpretrain_amd import ops
    from eventpretrain_amd.testing import det_normalish, det_uniform
    a, m = _swin_hub()
    x = (det_normalish("swin.voxels", (2, 5, 224, 224)) * 0.5).cuda()
    y = det_normalish("swin.sub_frame", (2, 1, 224, 224)).cuda()
    ops.set_compute_dtype(torch.float32)
    n1 = det_uniform("swin.noise", (2, 49), 0.0, 1.0).cuda()
    n2 = det_uniform("swin.noise.b", (2, 49), 0.0, 1.0).cuda()
    l1 = m(x, y, is_rec=True, noise=n1)[0].item()
    l2 = m(x, y, is_rec=True, noise=n2)[0].item()
    l1b = m(x, y, is_rec=True, noise=n1)[0].item()
    assert len(m.backbone._plans) == 2 and l1 == l1b and l1 != l2
    a.masking_strategy = "density"
    out = m(x, y, is_rec=True)
    dens = torch.nn.functional.avg_pool2d(x.sum(1).abs().unsqueeze(1), 32, 32).flatten(1).cpu()
    keep = torch.argsort(dens, dim=1, stable=True)[:, :24]
    exp_mask = torch.ones(2, 49)
    exp_mask.scatter_(1, keep, 0.0)
    assert torch.equal(out[11].cpu(), exp_mask) and math.isfinite(out[0].item())


def test_swin_contrastive_stage_f32_matches_reference():
    """PrHubModel.forward(is_rec=False) on the Swin-T hub: dense Swin forward (64/16/4 full windows + one 49-token group),
    Conv2d(512,768,2,2) CLIP-token projection, MoCo heads, queue InfoNCE and enqueue -- against the reference's outputs."""
    from eventpretrain_amd import ops
    from eventpretrain_amd.testing import det_normalish
    d = load_golden("con_swin_tiny_queue")
    a, m = _swin_hub(pr_phase="con", use_queue=True, mask_ratio=0.0)
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == jl(d["state_keys"])
    x = det_normalish("con.voxels", (2, 5, 224, 224)) * 0.5
    clip = det_normalish("con.clip_emb", (2, 197, 512))
    ops.set_compute_dtype(torch.float32)
    loss, h_org, h_proj, c_org, c_proj, attn = m(x.cuda(), clip.cuda())
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(d["loss"])) <= F32_LOSS_RTOL * abs(float(d["loss"]))
    assert list(attn.shape) == list(d["attn_shape"])
    assert_checksums(h_org, d["emb_h_org_checksums"], 1e-4)
    assert_checksums(h_proj, d["emb_h_proj_checksums"], 2e-4)
    assert_checksums(c_org, d["clip_org_checksums"], 1e-4)
    assert_checksums(c_proj, d["clip_proj_checksums"], 1e-4)
    assert_checksums(attn.float(), d["attn_checksums"], 1e-4)
    params = dict(m.named_parameters())
    for n, gn in zip(jl(d["grad_names"]), d["grad_norms"]):
        assert params[n].grad is not None, n
        assert params[n].grad.double().norm().item() == pytest.approx(gn, rel=5e-3, abs=2e-6), n
    assert_checksums(m.queue, d["queue_after_checksums"], 1e-5)
    assert int(m.queue_ptr) == int(d["queue_ptr_after"][0])
    assert len(m.backbone._plans) == 1


def test_swin_trainer_epoch_runs_and_learns():
    """The reference's epoch loop signature on the Swin hub with FusedAdamW: 4 steps on one repeated batch lower the loss."""
    from eventpretrain_amd import ops
    from eventpretrain_amd.optim import FusedAdamW
    from eventpretrain_amd.testing import det_normalish
    from eventpretrain_amd.trainer.pretrain.pr_trainer import pr_rec_one_epoch
    from eventpretrain_amd.utils import lr_decay as lrd
    from eventpretrain_amd.utils.misc import NativeScalerWithGradNormCount
    a, m = _swin_hub()
    a.lr, a.min_lr, a.warmup_epochs, a.epochs, a.batch_size = 2e-3, 1e-6, 0, 4, 2
    x = det_normalish("swin.voxels", (2, 5, 224, 224)) * 0.5
    y = det_normalish("swin.sub_frame", (2, 1, 224, 224))
    loader = [dict(events_voxel_grid=x, sub_frame=y, image_name=["a", "b"])] * 4
    ops.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(0)
        opt = FusedAdamW(lrd.param_groups_lrd(a, m, a.weight_decay, layer_decay=1), lr=a.lr, betas=(0.9, 0.95))
        scaler = NativeScalerWithGradNormCount()
        first = pr_rec_one_epoch(a, m, loader[:1], opt, 0, scaler)["reconstruct_loss"]
        for ep in range(1, 4):
            last = pr_rec_one_epoch(a, m, loader[:1], opt, ep, scaler)["reconstruct_loss"]
    finally:
        ops.set_compute_dtype(torch.float32)
    print(f"[swin-tiny] trainer loss {first:.4f} -> {last:.4f}")
    assert math.isfinite(last) and last < first


# ------------------------------------------------------------------------------------------- fixed-shape plan + HIP graph
def test_swin_static_plan_equals_pattern_plan_against_reference():
    """The fixed-shape window tables (group size 49, padded group list; what a captured graph needs) give the reference's
    step: same fixture, same tolerances as the pattern-sized plan -- and the two plans agree with each other to f32
    summation order."""
    from eventpretrain_amd import ops
    d = load_golden("rec_swin_tiny")
    x, y, noise = rec_inputs("swin", jl(d["cfg"]))
    ops.set_compute_dtype(torch.float32)
    res = {}
    for mode in ("pattern", "static"):
        a, m = _swin_hub()
        if mode == "static":
            prepare = m.backbone.enable_static_plan("cuda")
            assert prepare(noise) is True
        out = m(x.cuda(), y.cuda(), is_rec=True, noise=noise.cuda())
        out[0].backward()
        torch.cuda.synchronize()
        res[mode] = (out[0].item(), out[10].detach().cpu(), {n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None},
                     [c.cpu() for c in out[6:10]])
    for mode in res:
        assert abs(res[mode][0] - float(d["loss"])) / abs(float(d["loss"])) <= F32_LOSS_RTOL, mode
    assert abs(res["static"][0] - res["pattern"][0]) <= 1e-6 * abs(res["pattern"][0])
    assert torch.allclose(res["static"][1], res["pattern"][1], atol=1e-5, rtol=1e-5)
    for c, k in zip(res["static"][3], ("coords_l1", "coords_l2", "coords_l3", "coords_l4")):
        assert np.array_equal(c.numpy(), d[k]), k
    for n, g in res["pattern"][2].items():
        gs = res["static"][2][n]
        assert torch.allclose(gs, g, atol=1e-7 + 1e-4 * g.abs().max().item(), rtol=1e-4), n


@pytest.mark.parametrize("slack,dtype", [(1.25, torch.float32), (1.05, torch.float32), (0.9, torch.float32), (1.05, torch.bfloat16)])
def test_swin_graphed_step_follows_eager_trajectory(slack, dtype):
    """ONE captured graph serves every mask pattern (tables refreshed by an H2D copy per step); patterns that need more
    groups than the fixed shape holds (slack 1.05: about a third of them) run eagerly in between without disturbing the
    graph; when NO pattern fits (0.9) the executor says so and stays eager. Either way the losses follow the plain eager executor's from the same start and noise stream."""
    from eventpretrain_amd import ops
    from eventpretrain_amd.engine import GraphedStep
    from eventpretrain_amd.optim import FusedAdamW
    from eventpretrain_amd.utils import lr_decay as lrd
    d = load_golden("rec_swin_tiny")
    x, y, _ = rec_inputs("swin", jl(d["cfg"]))
    B = x.shape[0]
    ops.set_compute_dtype(dtype)
    fwd = lambda mm, xx, yy, noise: mm(xx, yy, is_rec=True, noise=noise)
    runs, notes = {}, {}
    for mode in ("eager", "graph"):
        a, m = _swin_hub()
        opt = FusedAdamW(lrd.param_groups_lrd(a, m, 0.05, layer_decay=1), lr=1e-3, betas=(0.9, 0.95))
        prepare = m.backbone.enable_static_plan("cuda", slack=slack) if mode == "graph" else (lambda n: False)
        ex = GraphedStep(m, opt, fwd, [x.cuda(), y.cuda()], noise_shape=(B, 49), use_graph=(mode == "graph"), warmup=2,
                         step_prepare=prepare, host_generator=torch.Generator().manual_seed(77))
        if mode == "graph" and slack > 1.0:
            assert ex.note.startswith("hip-graph"), ex.note
        elif mode == "graph":
            assert ex.note.startswith("eager (graph capture failed"), ex.note      # no pattern fits: nothing to capture
        runs[mode] = [ex.step().item() for _ in range(12)]
        notes[mode] = (ex.eager_fallbacks, getattr(m.backbone, "_static_plan", None))
    ops.set_compute_dtype(torch.float32)
    assert runs["graph"] == pytest.approx(runs["eager"], rel=2e-5 if dtype == torch.float32 else 5e-3), (runs, notes)
    fallbacks, sp = notes["graph"]
    if slack >= 1.25:
        assert fallbacks == 0 and sp.overflows == 0
    if 1.0 < slack < 1.25:
        assert 0 < fallbacks < 12, fallbacks
    print(f"[swin graph] slack {slack}: {fallbacks} eager fall-back steps of 12, plan loads {sp.loads}")
