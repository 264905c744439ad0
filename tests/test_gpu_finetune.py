"""Classification fine-tuning path on the GPU: the drop-in FtClsHubModel and epoch loops against
the fixtures the reference itself produced."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import assert_checksums, jl

pytestmark = pytest.mark.gpu


def _hub(tag):
    from eventpretrain_amd.model.finetune_cls import ft_cls_hub_model as ft
    from eventpretrain_amd.testing import det_fill_module_, make_args
    bt = "vit" if tag == "vit_small" else "swin"
    a = make_args(phase="finetune_cls", model_size="small" if bt == "vit" else "tiny", backbone_type=bt, num_classes=10, mask_ratio=0.0,
                  device="cuda", dataset_type="n-caltech101", clip_grad=None, smoothing=0)
    fac = ft.finetune_cls_hub_model_small_patch16 if bt == "vit" else ft.finetune_cls_hub_model_swin_tiny_window7
    m = fac(a)
    det_fill_module_(m)
    return a, m.cuda().train()


@pytest.mark.parametrize("tag", ["vit_small", "swin_tiny"])
def test_ft_cls_step_f32_matches_reference(tag):
    from eventpretrain_amd import ops
    from eventpretrain_amd.testing import det_normalish
    d = load_golden("ft_cls_" + tag)
    a, m = _hub(tag)
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == jl(d["state_keys"])
    x = (det_normalish("ft.voxels", (2, 5, 224, 224)) * 0.5).cuda()
    label = torch.from_numpy(d["label"]).cuda()
    ops.set_compute_dtype(torch.float32)
    out = m(x)
    pred, emb_h, attn = out[-2], out[-3], out[-1]
    loss = ops.CrossEntropyFn.apply(pred, label)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(d["loss"])) <= 1e-4 * abs(float(d["loss"]))
    assert torch.allclose(pred.cpu(), torch.from_numpy(d["pred"]), atol=1e-4, rtol=1e-4)
    assert_checksums(emb_h, d["emb_h_checksums"], 1e-4)
    assert_checksums(attn.float(), d["attn_checksums"], 1e-4)
    params = dict(m.named_parameters())
    for n, gn in zip(jl(d["grad_names"]), d["grad_norms"]):
        assert params[n].grad is not None, n
        assert params[n].grad.double().norm().item() == pytest.approx(gn, rel=5e-3, abs=2e-7), n
    for k in ("classify_head.weight", "classify_head.bias"):
        ref = torch.from_numpy(d["grad::" + k])
        assert torch.allclose(params[k].grad.cpu(), ref, atol=1e-6 + 1e-3 * ref.abs().max().item(), rtol=1e-3), k


def test_ft_cls_epoch_loops_learn_and_evaluate():
    """ft_train_one_epoch / ft_val with FusedAdamW in bf16 mode on one repeated batch: loss falls, the labels rank in the
    top 5, meters have the reference's names."""
    from eventpretrain_amd import ops
    from eventpretrain_amd.optim import FusedAdamW
    from eventpretrain_amd.testing import det_normalish
    from eventpretrain_amd.trainer.finetune_cls.ft_cls_trainer import ft_train_one_epoch, ft_val
    from eventpretrain_amd.utils import lr_decay as lrd
    from eventpretrain_amd.utils.misc import NativeScalerWithGradNormCount
    a, m = _hub("vit_small")
    a.lr, a.min_lr, a.warmup_epochs, a.epochs = 2e-3, 1e-4, 0, 12
    x = det_normalish("ft.voxels", (2, 5, 224, 224)) * 0.5
    loader = [dict(events_voxel_grid=x, label=torch.tensor([3, 7]), image_name=["a", "b"])]
    ops.set_compute_dtype(torch.bfloat16)
    try:
        opt = FusedAdamW(lrd.param_groups_lrd(a, m, a.weight_decay, layer_decay=0.75), lr=a.lr, betas=(0.9, 0.999))
        scaler = NativeScalerWithGradNormCount()
        hist = [ft_train_one_epoch(a, m, loader, opt, ep, scaler)["loss_cls"] for ep in range(12)]
        stats = ft_val(a, m, loader, 0)
    finally:
        ops.set_compute_dtype(torch.float32)
    assert math.isfinite(hist[-1]) and hist[-1] < 0.7 * hist[0], hist
    assert set(stats) == {"loss_cls", "acc1", "acc5"} and stats["acc1"] >= 50.0 and stats["acc5"] == 100.0


def test_label_smoothing_and_grad_clipping():
    """evp_cross_entropy_smooth through the C-ABI against the oracle's restated LabelSmoothingCrossEntropy (loss and
    logits gradient, f32: 1e-5 rel), and NativeScalerWithGradNormCount(clip_grad=c): one FusedAdamW step equals the
    oracle's AdamW update of gradients scaled by min(1, c / (norm + 1e-6)); the returned norm is the pre-clip norm."""
    from eventpretrain_amd import ops
    from eventpretrain_amd.optim import FusedAdamW
    from eventpretrain_amd.utils.misc import NativeScalerWithGradNormCount
    from oracle import model_oracle as mo
    g = torch.Generator().manual_seed(11)
    for n_cls in (10, 101, 1000):
        pred = (torch.randn(6, n_cls, generator=g) * 2).requires_grad_(True)
        label = torch.randint(0, n_cls, (6,), generator=g)
        for s_ in (0.0, 0.1):
            ref = mo.label_smoothing_ce(pred.double(), label, s_)
            (gref,) = torch.autograd.grad(ref, pred)
            p2 = pred.detach().cuda().requires_grad_(True)
            out = ops.CrossEntropyFn.apply(p2, label.cuda(), s_)
            out.backward()
            assert abs(out.item() - ref.item()) <= 1e-5 * abs(ref.item()), (n_cls, s_)
            assert torch.allclose(p2.grad.cpu(), gref.float(), rtol=1e-4, atol=1e-7), (n_cls, s_)

    w0 = torch.randn(32, 16, generator=g)
    b0 = torch.randn(32, generator=g)
    x = torch.randn(8, 16, generator=g)
    for clip in (0.05, 1e6):
        w = torch.nn.Parameter(w0.clone().cuda())
        b = torch.nn.Parameter(b0.clone().cuda())
        opt = FusedAdamW([{"params": [w], "weight_decay": 0.05}, {"params": [b], "weight_decay": 0.0}], lr=1e-2, betas=(0.9, 0.95))
        loss = (torch.nn.functional.linear(x.cuda(), w, b) ** 2).mean()
        norm = NativeScalerWithGradNormCount()(loss, opt, clip_grad=clip, parameters=[w, b])
        gw, gb = w.grad.cpu(), b.grad.cpu()
        total = mo.grad_norm([gw, gb])
        assert abs(float(norm) - float(total)) <= 1e-5 * float(total)
        c = mo.clip_coef(float(total), clip)
        assert (c < 1.0) == (clip < 1.0)
        for p_new, p_old, gr, wd in ((w, w0, gw, 0.05), (b, b0, gb, 0.0)):
            ref, _, _ = mo.adamw_step(p_old, gr * c, torch.zeros_like(p_old), torch.zeros_like(p_old), 1, 1e-2, wd)
            assert torch.allclose(p_new.detach().cpu(), ref, rtol=1e-5, atol=1e-6), clip
        assert opt.grad_scale == 1.0
