#!/usr/bin/env python3
"""Golden-vector generator: runs the REFERENCE (BIT-Vision/EventPretrain, mounted
read-only at /root/reference) on deterministic inputs and writes plain-array
fixtures under tests/golden/.

TEST INFRASTRUCTURE ONLY. Runs only in the build container (the reference does
not exist on the GPU box). Nothing from the reference is copied: the fixtures
hold inputs, outputs and checksums as numpy arrays. Pickled modules are never
written.

The reference's model files import three names from `timm.models.layers`
(model/sub_module/vit_block.py:5). timm is not installed here; an in-memory
module object supplying `to_2tuple`, `trunc_normal_` and a `DropPath` that is
never executed (all drop rates are 0 -> nn.Identity, vit_block.py:241) is put
in sys.modules for the duration of this script.

Usage:  python oracle/gen_golden.py [--only voxel,pos,mask,tiny,small,base,train,con,convsmall,swin,swincon,augment,evaug,ftcls,density,autocast,conbase,swinbase,convbase,chain,chainnim,trainbf16]
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("EVP_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from eventpretrain_amd.testing import (det_fill_module_, det_normalish, det_uniform,  # noqa: E402
                                       make_args, synthetic_events)


def _install_timm_standin():
    if "timm" in sys.modules:
        return
    import collections.abc
    from itertools import repeat

    def to_2tuple(x):
        if isinstance(x, collections.abc.Iterable) and not isinstance(x, str):
            return tuple(x)
        return tuple(repeat(x, 2))

    class DropPath(torch.nn.Module):  # never executed: rate 0 -> nn.Identity in the reference
        def __init__(self, p=0.0):
            super().__init__()
            self.p = p

        def forward(self, x):
            if self.p == 0.0 or not self.training:
                return x
            raise RuntimeError("DropPath stand-in executed with p>0")

    def trunc_normal_(t, mean=0.0, std=1.0, a=-2.0, b=2.0):
        return torch.nn.init.trunc_normal_(t, mean=mean, std=std, a=a, b=b)

    timm = types.ModuleType("timm")
    models = types.ModuleType("timm.models")
    layers = types.ModuleType("timm.models.layers")
    layers.DropPath, layers.to_2tuple, layers.trunc_normal_ = DropPath, to_2tuple, trunc_normal_
    timm.models, models.layers = models, layers
    timm.__version__ = "0.3.2"
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.layers": layers})


def _ref():
    if REF not in sys.path:
        sys.path.insert(0, REF)
    _install_timm_standin()


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


def checksums(t: torch.Tensor):
    """Layout-sensitive checksums of a tensor in float64."""
    d = t.detach().double().flatten()
    w = det_uniform("checksum.weights", (d.numel(),)).double()
    return np.array([d.sum().item(), d.abs().sum().item(), (d * w).sum().item(), (d * d).sum().item()])


# --------------------------------------------------------------------------- voxel
def gen_voxel():
    _ref()
    torch.set_num_threads(1)     # the reference's shipped setting (main_pretrain.py:12): sequential index_add_
    from dataset.dataset_utils.events_to_voxel_grid import events_to_voxel_grid
    out = {}
    args = make_args(num_bins=5)
    # known-answer test from SURVEY.md section 4
    kat = np.array([[0, 0, 0.0, 1], [1, 0, 0.25, 0], [2, 1, 0.5, 1], [3, 3, 0.9, 0], [3, 3, 1.0, 1]], dtype=np.float64)
    out["kat_events"] = kat
    out["kat_grid"] = events_to_voxel_grid(args, kat.copy(), (4, 4))
    cases = []

    def case(tag, ev, size, bins=5, is_txyp=False):
        a = make_args(num_bins=bins)
        g = events_to_voxel_grid(a, ev.copy(), size, is_txyp=is_txyp)
        out[f"{tag}_events"] = ev
        out[f"{tag}_grid"] = g
        cases.append(dict(tag=tag, H=size[0], W=size[1], bins=bins, is_txyp=bool(is_txyp)))

    # small clips, non-square grid so that H/W mix-ups show up
    for i in range(3):
        case(f"rand{i}", synthetic_events(100 + i, 5000, width=48, height=32), (32, 48))
    case("signed", synthetic_events(110, 4000, width=48, height=32, signed_polarity=True), (32, 48))
    ev = synthetic_events(111, 3000, width=40, height=24)
    case("txyp", ev[:, [2, 0, 1, 3]].copy(), (24, 40), is_txyp=True)
    ev = synthetic_events(112, 2000, width=16, height=16)
    ev[:, 2] = 0.125                                            # deltaT == 0 -> 1.0
    case("samet", ev, (16, 16))
    case("single", synthetic_events(113, 1, width=8, height=8), (8, 8))
    ev = synthetic_events(114, 6000, width=640, height=480)       # sensor->input rescale leaves fractions
    ev[:, 0] *= 48 / 640
    ev[:, 1] *= 32 / 480
    case("frac", ev, (32, 48))
    case("bins3", synthetic_events(115, 3000, width=20, height=12), (12, 20), bins=3)
    case("bins9", synthetic_events(116, 3000, width=20, height=12), (12, 20), bins=9)
    ev = synthetic_events(117, 4000, width=48, height=32)         # microsecond-scale stamps, large offset
    ev[:, 2] = ev[:, 2] * 1e6 + 1.7e9
    case("bigt", ev, (32, 48))
    # full-size clip: keep checksums + a strided sample, not the 1 MB grid
    ev = synthetic_events(0, 100_000)
    g = events_to_voxel_grid(args, ev.copy(), (224, 224))
    out["full0_checksums"] = checksums(g)
    out["full0_sample"] = g[:, ::7, ::5].contiguous()
    out["cases"] = np.array(json.dumps(cases))
    save("voxel", **out)
    torch.set_num_threads(8)


# --------------------------------------------------------------------------- pos embed
def gen_pos():
    _ref()
    from utils.pos_embed import get_2d_sincos_pos_embed
    out = {}
    for d, g in [(64, 4), (128, 4), (192, 4), (384, 14), (512, 14), (768, 14), (256, 7)]:
        pe = get_2d_sincos_pos_embed(d, g)
        out[f"d{d}_g{g}_dtype"] = np.array(str(pe.dtype))
        t = torch.from_numpy(pe).float()
        out[f"d{d}_g{g}_checksums"] = checksums(t)
        if d * g * g <= 192 * 16:
            out[f"d{d}_g{g}_table"] = t
        else:
            out[f"d{d}_g{g}_rows"] = t[[0, 1, 17, g * g - 1]]
    save("pos_embed", **out)


# --------------------------------------------------------------------------- masking
def gen_mask():
    _ref()
    from model.backbone.vit import ViT
    out = {}
    cases = []
    for tag, inp, L, ratio, B, seed in [("l16", 64, 16, 0.5, 4, 11), ("l196", 224, 196, 0.5, 8, 12),
                                         ("l196r75", 224, 196, 0.75, 3, 13), ("l196r10", 224, 196, 0.1, 2, 14)]:
        a = make_args(mask_ratio=ratio)
        m = ViT(a, input_size=inp, patch_size=16, embed_dim=32, depth=1, num_heads=1, mask_ratio=ratio)
        x = torch.zeros(B, 5, inp, inp)
        torch.manual_seed(seed)
        ids_keep, mask, ids_restore = m.random_masking(x)
        torch.manual_seed(seed)
        noise = torch.rand(B, L)                 # same CPU stream the reference just consumed (vit.py:78)
        assert torch.equal(torch.argsort(noise, dim=1)[:, :ids_keep.shape[1]], ids_keep)
        srt = torch.sort(noise, dim=1).values
        assert (srt[:, 1:] > srt[:, :-1]).all(), "tie in noise; pick another seed"
        out[f"{tag}_noise"], out[f"{tag}_ids_keep"] = noise, ids_keep
        out[f"{tag}_mask"], out[f"{tag}_ids_restore"] = mask, ids_restore
        cases.append(dict(tag=tag, L=L, ratio=ratio, B=B))
    out["cases"] = np.array(json.dumps(cases))
    save("masking", **out)


# --------------------------------------------------------------------------- model compositions
def _compose(cfg):
    """Hand-compose backbone + decoder from the reference CLASSES (SURVEY.md header: the hub factories cannot
    build ViT-Base or the 64x64 tiny model)."""
    _ref()
    from functools import partial
    from model.backbone.vit import ViT
    from model.pretrain.pr_rec_decoder import PrRecDecoder
    a = make_args(mask_ratio=cfg["mask_ratio"], patch_size=cfg["patch"])
    ln = partial(torch.nn.LayerNorm, eps=1e-6)
    bb = ViT(a, input_size=cfg["input"], patch_size=cfg["patch"], embed_dim=cfg["dim"], depth=cfg["depth"],
             num_heads=cfg["heads"], mlp_ratio=4, norm_layer=ln, num_bins=5, mask_ratio=cfg["mask_ratio"])
    L = (cfg["input"] // cfg["patch"]) ** 2
    dec = PrRecDecoder(patch_size=cfg["patch"], num_patches=L, encoder_embed_dim=[cfg["dim"]],
                       embed_dim=cfg["dec_dim"], depth=cfg["dec_depth"], num_heads=cfg["dec_heads"],
                       mlp_ratio=[4, 4, 4], norm_layer=ln, frame_chans=1)

    class Hub(torch.nn.Module):  # state-dict keys as PrHubModel: backbone.*, pretrain_rec_decoder.*
        def __init__(self):
            super().__init__()
            self.backbone, self.pretrain_rec_decoder = bb, dec

    hub = Hub()
    det_fill_module_(hub)
    return a, hub


def _rec_loss_ref(a, patch, pred, target, mask):
    """PrHubModel.reconstruct_loss called unbound on a stub (pr_hub_model.py:125-141)."""
    from model.pretrain.pr_hub_model import PrHubModel
    stub = types.SimpleNamespace(patch_size=patch, norm_pix_loss=a.norm_pix_loss, mask_ratio=a.mask_ratio)
    return PrHubModel.reconstruct_loss(stub, pred, target, mask)


CFGS = {
    "tiny": dict(input=64, patch=16, dim=192, depth=12, heads=3, dec_dim=128, dec_depth=4, dec_heads=4,
                 mask_ratio=0.5, B=2),
    "base": dict(input=224, patch=16, dim=768, depth=12, heads=12, dec_dim=512, dec_depth=8, dec_heads=16,
                 mask_ratio=0.5, B=2),
}


def _inputs(tag, cfg):
    B, S = cfg["B"], cfg["input"]
    L = (S // cfg["patch"]) ** 2
    x = det_normalish(f"{tag}.voxels", (B, 5, S, S)) * 0.5
    y = det_normalish(f"{tag}.sub_frame", (B, 1, S, S))
    noise = det_uniform(f"{tag}.noise", (B, L), 0.0, 1.0)
    return x, y, noise


def _run_rec(tag, cfg, hub_forward, named_params, extra=None):
    """Shared: forward with explicit noise (torch.rand patched for the one call), backward, collect."""
    x, y, noise = _inputs(tag, cfg)
    real_rand = torch.rand
    calls = []

    def fake_rand(*shape, **kw):
        calls.append(shape)
        assert tuple(shape) == tuple(noise.shape), shape
        return noise.clone()

    torch.rand = fake_rand
    try:
        res = hub_forward(x, y)
    finally:
        torch.rand = real_rand
    assert len(calls) == 1
    loss, emb_l1, emb_l2, emb_lh, pred, mask, ids_restore = res
    loss.backward()
    out = dict(noise=noise, loss=loss.detach().double(), mask=mask, ids_restore=ids_restore)
    out["pred_checksums"] = checksums(pred)
    out["emb_l1_checksums"], out["emb_l2_checksums"] = checksums(emb_l1), checksums(emb_l2)
    out["emb_lh_checksums"] = checksums(emb_lh)
    names, gnorms, gsums = [], [], []
    for n, p in named_params:
        if p.grad is None:
            continue
        names.append(n)
        gnorms.append(p.grad.double().norm().item())
        gsums.append(checksums(p.grad)[2])
    out["grad_names"] = np.array(json.dumps(names))
    out["grad_norms"] = np.array(gnorms)
    out["grad_wsums"] = np.array(gsums)
    out["total_grad_norm"] = np.array(float(np.sqrt(np.sum(np.square(gnorms)))))
    if extra:
        out.update(extra(res))
    return out


def gen_composed(tag):
    cfg = CFGS[tag]
    a, hub = _compose(cfg)

    def fwd(x, y):
        emb_l1, emb_l2, emb_lh, mask, ids_restore = hub.backbone(x, mask=True)
        pred = hub.pretrain_rec_decoder(emb_lh, ids_restore)
        loss = _rec_loss_ref(a, cfg["patch"], pred, y, mask)
        return loss, emb_l1, emb_l2, emb_lh, pred, mask, ids_restore

    def extra(res):
        if tag != "tiny":
            return {}
        loss, emb_l1, emb_l2, emb_lh, pred, mask, ids_restore = res
        sd = dict(hub.named_parameters())
        e = dict(pred=pred, emb_lh=emb_lh, emb_l1=emb_l1)
        for n in ["backbone.patch_embed.proj.weight", "backbone.vit_block.0.attn.qkv.weight",
                  "backbone.vit_block.0.norm1.weight", "backbone.vit_block.11.mlp.fc2.bias",
                  "pretrain_rec_decoder.mask_token", "pretrain_rec_decoder.pred.weight",
                  "backbone.norm_layer.bias", "backbone.patch_embed.norm.weight"]:
            e["grad::" + n] = sd[n].grad
        return e

    out = _run_rec(tag, cfg, fwd, list(hub.named_parameters()), extra)
    out["cfg"] = np.array(json.dumps(cfg))
    save(f"rec_{tag}", **out)


def gen_small():
    """The as-shipped path: pretrain_hub_model_small_patch16 -> PrHubModel.forward(is_rec=True)."""
    _ref()
    from model.pretrain.pr_hub_model import pretrain_hub_model_small_patch16
    cfg = dict(input=224, patch=16, dim=384, depth=12, heads=12, dec_dim=256, dec_depth=8, dec_heads=8,
               mask_ratio=0.5, B=2)
    a = make_args(model_size="small", pr_phase="rec")
    hub = pretrain_hub_model_small_patch16(a, emb_frames_dim=512, queue_length=1024, T=0.07)
    det_fill_module_(hub)
    hub.train(True)
    out = _run_rec("small", cfg, lambda x, y: hub(x, y, is_rec=True), list(hub.named_parameters()))
    out["cfg"] = np.array(json.dumps(cfg))
    out["state_keys"] = np.array(json.dumps({k: list(v.shape) for k, v in hub.state_dict().items()}))
    save("rec_small", **out)


def gen_convsmall():
    """ConvViT-Small through the as-shipped hub factory (BASELINE.json config 4 at the size the reference can build)."""
    _ref()
    from model.pretrain.pr_hub_model import pretrain_hub_model_small_patch16
    cfg = dict(input=224, patch=16, dims=[128, 256, 384], depth=[2, 2, 11], heads=12, dec_dim=256, dec_depth=8, dec_heads=8,
               mask_ratio=0.5, B=2)
    a = make_args(model_size="small", pr_phase="rec", backbone_type="convvit")
    hub = pretrain_hub_model_small_patch16(a, emb_frames_dim=512, queue_length=1024, T=0.07)
    det_fill_module_(hub)
    hub.train(True)
    out = _run_rec("convsmall", cfg, lambda x, y: hub(x, y, is_rec=True), list(hub.named_parameters()))
    out["cfg"] = np.array(json.dumps(cfg))
    out["state_keys"] = np.array(json.dumps({k: list(v.shape) for k, v in hub.state_dict().items()}))
    save("rec_convsmall", **out)


def gen_convbase():
    """ConvViT-Base (BASELINE.json config 4 as bench.py times it): the reference's own `convvit_base_patch16` factory
    (model/backbone/convvit.py:218-224) hand-composed with `pretrain_rec_decoder_base_patch16` (pr_rec_decoder.py:89-95) -- the hub
    factory always builds the 384-wide small decoder (pr_hub_model.py:77), which cannot take a 768-wide backbone. Same contents as
    rec_convsmall."""
    _ref()
    from model.backbone.convvit import convvit_base_patch16
    from model.pretrain.pr_rec_decoder import pretrain_rec_decoder_base_patch16
    cfg = dict(input=224, patch=16, dims=[256, 384, 768], depth=[2, 2, 11], heads=12, dec_dim=512, dec_depth=8, dec_heads=16,
               mask_ratio=0.5, B=2)
    a = make_args(model_size="base", pr_phase="rec", backbone_type="convvit")
    bb = convvit_base_patch16(args=a, num_bins=5, mask_ratio=0.5, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.)
    dec = pretrain_rec_decoder_base_patch16(frame_chans=1)

    class Hub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone, self.pretrain_rec_decoder = bb, dec

    hub = Hub()
    det_fill_module_(hub)
    hub.train(True)

    def fwd(x, y):
        emb_l1, emb_l2, emb_lh, mask, ids_restore = hub.backbone(x, mask=True)
        pred = hub.pretrain_rec_decoder(emb_lh, ids_restore)
        loss = _rec_loss_ref(a, 16, pred, y, mask)
        return loss, emb_l1, emb_l2, emb_lh, pred, mask, ids_restore

    out = _run_rec("convbase", cfg, fwd, list(hub.named_parameters()))
    out["cfg"] = np.array(json.dumps(cfg))
    out["state_keys"] = np.array(json.dumps({k: list(v.shape) for k, v in hub.state_dict().items()}))
    save("rec_convbase", **out)


def gen_swin():
    """Swin-Tiny (window 7) through the as-shipped hub factory, masked reconstruction step (BASELINE.json config 5;
    model/backbone/swin.py:174-246 + model/sub_module/swin_block.py), B=2, 224x224, decoder patch 32."""
    _ref()
    from model.pretrain.pr_hub_model import pretrain_hub_model_swin_tiny_patch16
    cfg = dict(input=224, patch=32, dims=[96, 192, 384, 768], depths=[2, 2, 6, 2], heads=[3, 6, 12, 24], window=7,
               dec_dim=256, dec_depth=8, dec_heads=8, mask_ratio=0.5, B=2)
    a = make_args(model_size="tiny", pr_phase="rec", backbone_type="swin")
    hub = pretrain_hub_model_swin_tiny_patch16(a, emb_frames_dim=512, queue_length=1024, T=0.07)
    det_fill_module_(hub)
    hub.train(True)
    keep = {}

    def fwd(x, y):
        r = hub(x, y, is_rec=True)
        (loss, emb_l1, emb_l2, emb_l3, emb_l4, emb_lh, c1, c2, c3, c4, pred, mask, ids_restore, attn) = r
        keep.update(emb_l3=emb_l3, emb_l4=emb_l4, c1=c1, c2=c2, c3=c3, c4=c4, attn=attn)
        return loss, emb_l1, emb_l2, emb_lh, pred, mask, ids_restore

    def extra(res):
        e = dict(emb_l3_checksums=checksums(keep["emb_l3"]), emb_l4_checksums=checksums(keep["emb_l4"]),
                 attn_checksums=checksums(keep["attn"]), attn_shape=np.array(keep["attn"].shape),
                 coords_l1=keep["c1"], coords_l2=keep["c2"], coords_l3=keep["c3"], coords_l4=keep["c4"],
                 emb_l4=keep["emb_l4"], emb_lh=res[3])
        sd = dict(hub.named_parameters())
        for n in ["backbone.swin_block.0.blocks.0.attn.relative_position_bias_table",
                  "backbone.swin_block.2.blocks.1.attn.relative_position_bias_table",
                  "backbone.swin_block.0.downsample.reduction.weight", "backbone.stage1_output_decode.weight",
                  "backbone.patch_embed.proj.weight", "backbone.swin_block.3.blocks.1.mlp.fc2.bias"]:
            g = sd[n].grad
            if g.numel() <= 20000:
                e["grad::" + n] = g
            else:
                e["gradsum::" + n] = checksums(g)
        return e

    out = _run_rec("swin", cfg, fwd, list(hub.named_parameters()), extra)
    out["cfg"] = np.array(json.dumps(cfg))
    out["state_keys"] = np.array(json.dumps({k: list(v.shape) for k, v in hub.state_dict().items()}))
    save("rec_swin_tiny", **out)


# --------------------------------------------------------------------------- trainer trajectory
def gen_train_bf16():
    """The same 5 steps with the model's forward under torch.autocast("cpu", bfloat16) -- what the reference's loop is on a GPU, where
    its `with torch.cuda.amp.autocast():` (trainer/pretrain/pr_trainer.py:26) is live (on the CPU that context is a no-op, so
    train_tiny.npz is an fp32 trajectory). Yardstick for the build's bf16 mode over several optimiser steps: losses only."""
    gen_train(autocast_bf16=True)


def gen_train(autocast_bf16=False):
    """5 optimiser steps of the reference's own pr_rec_one_epoch (trainer/pretrain/pr_trainer.py:9-89) with
    param_groups_lrd + AdamW(betas=(0.9,0.95)) as main_pretrain.py:323-343 sets them up, on the tiny model."""
    _ref()
    import utils.lr_decay as lrd
    from trainer.pretrain.pr_trainer import pr_rec_one_epoch
    from utils.misc import NativeScalerWithGradNormCount
    from utils.lr_sched import adjust_learning_rate
    cfg = CFGS["tiny"]
    a, hub = _compose(cfg)
    a.batch_size, a.epochs, a.warmup_epochs, a.accum_iter = cfg["B"], 4, 1, 1
    a.lr = a.blr * a.batch_size * a.accum_iter / 256 * 64     # x64 so that 3 steps move the loss visibly
    a.min_lr = 1e-6

    class Model(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone, self.pretrain_rec_decoder = hub.backbone, hub.pretrain_rec_decoder

        def forward(self, x, y, is_rec=True):
            if autocast_bf16:
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    emb_l1, emb_l2, emb_lh, mask, ids_restore = self.backbone(x, mask=True)
                    pred = self.pretrain_rec_decoder(emb_lh, ids_restore)
                    loss = _rec_loss_ref(a, cfg["patch"], pred.float(), y, mask)
                return loss.float(), emb_l1, emb_l2, emb_lh, pred, mask, ids_restore
            emb_l1, emb_l2, emb_lh, mask, ids_restore = self.backbone(x, mask=True)
            pred = self.pretrain_rec_decoder(emb_lh, ids_restore)
            loss = _rec_loss_ref(a, cfg["patch"], pred, y, mask)
            return loss, emb_l1, emb_l2, emb_lh, pred, mask, ids_restore

    model = Model()
    groups = lrd.param_groups_lrd(a, model, a.weight_decay, layer_decay=1)
    opt = torch.optim.AdamW(groups, lr=a.lr, betas=(0.9, 0.95))
    scaler = NativeScalerWithGradNormCount()
    steps, noises, batches = 5, [], []
    for s in range(steps):
        x = det_normalish(f"train.voxels.{s}", (cfg["B"], 5, 64, 64)) * 0.5
        y = det_normalish(f"train.sub_frame.{s}", (cfg["B"], 1, 64, 64))
        noises.append(det_uniform(f"train.noise.{s}", (cfg["B"], 16), 0.0, 1.0))
        batches.append(dict(events_voxel_grid=x, sub_frame=y, image_name=[f"s{s}"] * cfg["B"]))
    it = iter(noises)
    real_rand = torch.rand
    torch.rand = lambda *s, **k: next(it).clone()
    losses, lrs = [], []
    real_step = opt.step

    def spy_step(*aa, **kk):
        lrs.append(opt.param_groups[0]["lr"])
        return real_step(*aa, **kk)

    opt.step = spy_step
    fwd = model.forward

    def spy_fwd(*aa, **kk):
        r = fwd(*aa, **kk)
        losses.append(r[0].item())
        return r

    model.forward = spy_fwd
    try:
        stats = pr_rec_one_epoch(a, model, batches, opt, 0, scaler, log_writer=None)
    finally:
        torch.rand = real_rand
    if autocast_bf16:
        print("  bf16-autocast losses", losses)
        save("train_tiny_bf16", losses=np.array(losses), lrs=np.array(lrs))
        return
    out = dict(losses=np.array(losses), lrs=np.array(lrs), noise=torch.stack(noises),
               stats=np.array(json.dumps(stats)), lr=np.array(a.lr), min_lr=np.array(a.min_lr),
               warmup_epochs=np.array(a.warmup_epochs), epochs=np.array(a.epochs),
               weight_decay=np.array(a.weight_decay))
    names, psums = [], []
    for n, p in model.named_parameters():
        names.append(n)
        psums.append(checksums(p)[2])
    out["param_names"] = np.array(json.dumps(names))
    out["param_wsums"] = np.array(psums)
    cnt = {"decay": 0, "no_decay": 0}
    for g in groups:
        cnt["decay" if g["weight_decay"] > 0 else "no_decay"] += len(g["params"])
    out["group_decay"] = np.array(json.dumps(cnt))
    out["n_groups"] = np.array(len(groups))
    # LR schedule samples (utils/lr_sched.py:3-16)
    sched = []
    fake_opt = types.SimpleNamespace(param_groups=[{"lr": 0.0}, {"lr": 0.0, "lr_scale": 0.5}])
    sa = make_args(lr=2e-3, min_lr=1e-5, warmup_epochs=5, epochs=40)
    for e in [0.0, 0.25, 2.5, 4.999, 5.0, 5.5, 20.0, 39.0, 39.99]:
        lr = adjust_learning_rate(fake_opt, e, sa)
        sched.append([e, lr, fake_opt.param_groups[0]["lr"], fake_opt.param_groups[1]["lr"]])
    out["sched"] = np.array(sched)
    save("train_tiny", **out)


# --------------------------------------------------------------------------- contrastive stage
def gen_con():
    """PrHubModel.forward(is_rec=False) with and without the queue (pr_hub_model.py:208-245) on ViT-Small 224
    (the only size the hub factory builds), B=2, queue_length=4."""
    _ref()
    from model.pretrain.pr_hub_model import pretrain_hub_model_small_patch16
    for use_queue in (True, False):
        a = make_args(model_size="small", pr_phase="con", use_queue=use_queue, mask_ratio=0.0)
        hub = pretrain_hub_model_small_patch16(a, emb_frames_dim=512, queue_length=4, T=0.07)
        if not use_queue:
            pass
        det_fill_module_(hub)
        hub.train(True)
        x = det_normalish("con.voxels", (2, 5, 224, 224)) * 0.5
        clip = det_normalish("con.clip_emb", (2, 197, 512))
        q0 = hub.queue.clone() if use_queue else None
        loss, emb_h_org, emb_h_proj, clip_org, clip_proj, attn = hub(x, clip)
        loss.backward()
        out = dict(loss=loss.detach().double(), emb_h_org_checksums=checksums(emb_h_org),
                   emb_h_proj_checksums=checksums(emb_h_proj), clip_org_checksums=checksums(clip_org),
                   clip_proj_checksums=checksums(clip_proj), attn_checksums=checksums(attn))
        names, gn = [], []
        for n, p in hub.named_parameters():
            if p.grad is not None:
                names.append(n)
                gn.append(p.grad.double().norm().item())
        out["grad_names"], out["grad_norms"] = np.array(json.dumps(names)), np.array(gn)
        if use_queue:
            out["queue_after_checksums"] = checksums(hub.queue)
            out["queue_ptr_after"] = hub.queue_ptr.clone()
            out["queue_changed"] = np.array(float((hub.queue - q0).abs().sum()))
        bn = {k: checksums(v) for k, v in hub.state_dict().items() if "running_" in k}
        out["bn_keys"] = np.array(json.dumps(list(bn.keys())))
        out["bn_checksums"] = np.stack(list(bn.values()))
        out["state_keys"] = np.array(json.dumps({k: list(v.shape) for k, v in hub.state_dict().items()}))
        save("con_small_queue" if use_queue else "con_small_noqueue", **out)


def gen_swincon():
    """PrHubModel.forward(is_rec=False) on the Swin-T hub with the queue (dense Swin forward swin.py:248-290, Conv2d
    CLIP-token projection pr_hub_model.py:219-220), B=2, queue_length=4."""
    _ref()
    from model.pretrain.pr_hub_model import pretrain_hub_model_swin_tiny_patch16
    a = make_args(model_size="tiny", pr_phase="con", backbone_type="swin", use_queue=True, mask_ratio=0.0)
    hub = pretrain_hub_model_swin_tiny_patch16(a, emb_frames_dim=512, queue_length=4, T=0.07)
    det_fill_module_(hub)
    hub.train(True)
    x = det_normalish("con.voxels", (2, 5, 224, 224)) * 0.5
    clip = det_normalish("con.clip_emb", (2, 197, 512))
    loss, emb_h_org, emb_h_proj, clip_org, clip_proj, attn = hub(x, clip)
    loss.backward()
    out = dict(loss=loss.detach().double(), emb_h_org_checksums=checksums(emb_h_org),
               emb_h_proj_checksums=checksums(emb_h_proj), clip_org_checksums=checksums(clip_org),
               clip_proj_checksums=checksums(clip_proj), attn_checksums=checksums(attn), attn_shape=np.array(attn.shape))
    names, gn = [], []
    for n, p in hub.named_parameters():
        if p.grad is not None:
            names.append(n)
            gn.append(p.grad.double().norm().item())
    out["grad_names"], out["grad_norms"] = np.array(json.dumps(names)), np.array(gn)
    out["queue_after_checksums"] = checksums(hub.queue)
    out["queue_ptr_after"] = hub.queue_ptr.clone()
    out["state_keys"] = np.array(json.dumps({k: list(v.shape) for k, v in hub.state_dict().items()}))
    save("con_swin_tiny_queue", **out)


def gen_chain():
    """The loader's whole per-sample chain run by the reference itself (dataset/pretrain/pr_n_imagenet_dataset.py:82-89 on the running
    numpy stream, then the seeded evg_augment / frame_augment pair of pr_ef_imagenet_dataset.py:187-206):
        np.random.seed(s); get_random_index -> events[start:end] -> events_augment -> events_reshape -> events_to_voxel_grid ->
        seed2 = np.random.randint(1000); evg_augment(seed=seed2); frame_augment(seed=seed2, time_flip_flag)
    on synthetic sensor-shaped clips (640 x 480, as N-ImageNet). The fixture keeps the seeds, the decisions' observable results
    (window, clip length after augmentation, time-flip flag) and checksums + every 7th value of the two outputs."""
    _ref()
    from dataset.augmentation.events_augment import events_augment, events_reshape, get_random_index
    from dataset.augmentation.view_augment import evg_augment, frame_augment
    from dataset.dataset_utils.events_to_voxel_grid import events_to_voxel_grid
    from eventpretrain_amd.testing import synthetic_events
    torch.set_num_threads(1)
    out = {}
    cases = [("a", 101, 40_000, 15_000), ("b", 102, 150_000, 100_000), ("c", 103, 9_000, 15_000), ("d", 104, 150_000, 100_000)]
    for tag, seed, n_ev, fix in cases:
        a = make_args(crop_min=0.8, input_size=224, fix_events_num=fix, img_sensor_w=640, img_sensor_h=480)
        a.num_bins = 5
        ev = synthetic_events(7000 + seed, n_ev, width=640, height=480)
        frame = det_normalish(f"chain.frame.{tag}", (1, 480, 640))
        np.random.seed(seed)
        s0, s1 = get_random_index(a, ev, is_train=True)
        e = ev[s0:s1].copy()
        e = events_augment(a, e, size=(480, 640))
        n_aug = e.shape[0]
        e = events_reshape(e, 640, 480, 224, 224)
        evg = events_to_voxel_grid(a, e, size=(224, 224))
        seed2 = np.random.randint(1000)
        evg, tflag = evg_augment(a, evg, size=(224, 224), seed=seed2)
        fr = frame_augment(a, frame.clone(), seed=seed2, time_flip_flag=tflag)
        evg, fr = evg.contiguous(), fr.contiguous()
        out[f"{tag}_seed"], out[f"{tag}_n"], out[f"{tag}_fix"] = np.array(seed), np.array(n_ev), np.array(fix)
        out[f"{tag}_window"], out[f"{tag}_n_aug"], out[f"{tag}_tflip"] = np.array([s0, s1]), np.array(n_aug), np.array(int(tflag))
        out[f"{tag}_evg_checksums"], out[f"{tag}_evg_sample"] = checksums(evg), evg.flatten()[::7].clone()
        out[f"{tag}_frame_checksums"], out[f"{tag}_frame_sample"] = checksums(fr), fr.flatten()[::7].clone()
    out["tags"] = np.array(json.dumps([c[0] for c in cases]))
    save("loader_chain", **out)


def gen_chain_nimagenet():
    """The events half of PretrainNImageNetDataset.__getitem__ in ITS OWN draw order (dataset/pretrain/pr_n_imagenet_dataset.py:82-89;
    ADVICE r3: gen_chain above re-seeds before evg_augment as the EF-ImageNet dataset does, which the N-ImageNet one never does):
        np.random.seed(s) ONCE, then per sample on the running stream
        get_random_index -> events[start:end] -> events_augment -> events_reshape -> events_to_voxel_grid -> evg_augment(seed=None)
    for several samples in a row, as one DataLoader worker draws them. (The image half of that __getitem__ -- CLIP preprocessing of a
    JPEG -- draws nothing from numpy and is outside the path.) Synthetic sensor-shaped clips (640 x 480)."""
    _ref()
    from dataset.augmentation.events_augment import events_augment, events_reshape, get_random_index
    from dataset.augmentation.view_augment import evg_augment
    from dataset.dataset_utils.events_to_voxel_grid import events_to_voxel_grid
    from eventpretrain_amd.testing import synthetic_events
    torch.set_num_threads(1)
    out = {}
    runs = [("r1", 211, 15_000, [40_000, 9_000, 61_000]), ("r2", 212, 100_000, [150_000, 120_000])]
    for tag, seed, fix, sizes in runs:
        a = make_args(crop_min=0.8, input_size=224, fix_events_num=fix, img_sensor_w=640, img_sensor_h=480)
        a.num_bins = 5
        np.random.seed(seed)
        for i, n_ev in enumerate(sizes):
            ev = synthetic_events(8000 + seed * 10 + i, n_ev, width=640, height=480)
            s0, s1 = get_random_index(a, ev, is_train=True)
            e = ev[s0:s1].copy()
            e = events_augment(a, e, size=(480, 640))
            n_aug = e.shape[0]
            e = events_reshape(e, 640, 480, 224, 224)
            evg = events_to_voxel_grid(a, e, size=(224, 224))
            evg, tflag = evg_augment(a, evg, size=(224, 224))
            evg = evg.contiguous()
            k = f"{tag}_{i}"
            out[f"{k}_window"], out[f"{k}_n_aug"], out[f"{k}_tflip"] = np.array([s0, s1]), np.array(n_aug), np.array(int(tflag))
            out[f"{k}_evg_checksums"], out[f"{k}_evg_sample"] = checksums(evg), evg.flatten()[::7].clone()
        out[f"{tag}_seed"], out[f"{tag}_fix"], out[f"{tag}_sizes"] = np.array(seed), np.array(fix), np.array(sizes)
    out["tags"] = np.array(json.dumps([r[0] for r in runs]))
    save("loader_chain_nimagenet", **out)


def gen_augment():
    """evg_augment of the reference itself (dataset/augmentation/view_augment.py:84-95) under np.random.seed(seed), for
    sensor-shaped and input-shaped grids; the fixture keeps inputs' generator seeds, outputs and time-flip flags."""
    _ref()
    from dataset.augmentation.view_augment import evg_augment
    a = make_args(crop_min=0.8)
    out = {}
    cases = [("a", 11, (5, 224, 224), (224, 224)), ("b", 12, (5, 224, 224), (224, 224)), ("c", 13, (5, 120, 160), (224, 224)),
             ("d", 14, (5, 64, 64), (64, 64)), ("e", 15, (5, 224, 224), (224, 224)), ("f", 16, (3, 96, 128), (64, 64))]
    for tag, seed, shp, size in cases:
        a.num_bins = shp[0]
        v = det_normalish(f"aug.view.{tag}", shp)
        res, tflag = evg_augment(a, v.clone(), size=size, seed=seed)
        out[f"{tag}_seed"], out[f"{tag}_shape"], out[f"{tag}_size"] = np.array(seed), np.array(shp), np.array(size)
        out[f"{tag}_tflip"] = np.array(int(tflag))
        res = res.contiguous()
        if res.numel() <= 30000:
            out[f"{tag}_out"] = res                              # small cases in full
        else:
            out[f"{tag}_checksums"] = checksums(res)             # large ones: checksums + every 7th pixel
            out[f"{tag}_sample"] = res.flatten()[::7].clone()
    out["tags"] = np.array(json.dumps([c[0] for c in cases]))
    save("evg_augment", **out)


def gen_frameaug():
    """frame_augment of the reference itself (view_augment.py:79-89) under np.random.seed(seed), with the time-flip flag its
    evg_augment returns for the same seed (as pr_n_imagenet_dataset.py calls the pair): sensor-shaped and input-shaped frames."""
    _ref()
    from dataset.augmentation.view_augment import evg_augment, frame_augment
    out = {}
    cases = [("a", 31, (1, 224, 224), 224), ("b", 32, (1, 120, 160), 224), ("c", 33, (1, 64, 64), 64), ("d", 34, (3, 96, 128), 64),
             ("e", 35, (1, 260, 346), 224)]
    for tag, seed, shp, S in cases:
        a = make_args(crop_min=0.8, input_size=S)
        a.num_bins = 5
        f = det_normalish(f"aug.frame.{tag}", shp)
        _, tflag = evg_augment(a, det_normalish(f"aug.framevox.{tag}", (5,) + shp[1:]), size=(S, S), seed=seed)
        res = frame_augment(a, f.clone(), seed=seed, time_flip_flag=tflag).contiguous()
        out[f"{tag}_seed"], out[f"{tag}_shape"], out[f"{tag}_size"] = np.array(seed), np.array(shp), np.array(S)
        out[f"{tag}_tflip"] = np.array(int(tflag))
        if res.numel() <= 30000:
            out[f"{tag}_out"] = res
        else:
            out[f"{tag}_checksums"] = checksums(res)
            out[f"{tag}_sample"] = res.flatten()[::7].clone()
    out["tags"] = np.array(json.dumps([c[0] for c in cases]))
    save("frame_augment", **out)


def gen_evaug():
    """events_augment -> events_reshape -> events_to_voxel_grid of the reference itself (events_augment.py:80-86, :22-26,
    pr_n_imagenet_dataset.py:84-87) on seeded sensor-shaped clips; the fixture keeps the clip seeds, the augmented event
    arrays (small clips) and the voxel grids of the rescaled result."""
    _ref()
    from dataset.augmentation.events_augment import events_augment, events_reshape
    from dataset.dataset_utils.events_to_voxel_grid import events_to_voxel_grid
    out = {}
    cases = [("a", 21, 6000, (480, 640), 64), ("b", 22, 3000, (260, 346), 32), ("c", 23, 150, (128, 128), 32), ("d", 24, 60, (128, 128), 32)]
    for tag, seed, n, (sh, sw), S in cases:
        rng = np.random.default_rng(5000 + seed)
        ev = np.stack([np.floor(rng.uniform(0, sw, n)), np.floor(rng.uniform(0, sh, n)), np.sort(rng.uniform(0, 0.05, n)),
                       rng.integers(0, 2, n).astype(np.float64)], 1)
        a = make_args(num_bins=5)
        aug = events_augment(a, ev.copy(), size=(sh, sw), seed=seed)
        res = events_reshape(aug.copy(), sw, sh, S, S)
        vox = events_to_voxel_grid(a, res.copy(), size=(S, S))
        out[f"{tag}_meta"] = np.array([seed, n, sh, sw, S])
        out[f"{tag}_events_in"] = ev
        out[f"{tag}_events_out"] = aug
        out[f"{tag}_voxel"] = vox.numpy() if hasattr(vox, "numpy") else np.asarray(vox)
    out["tags"] = np.array(json.dumps([c[0] for c in cases]))
    save("events_augment", **out)


def gen_ftcls():
    """Classification fine-tuning step (model/finetune_cls/ft_cls_hub_model.py:118-139 + nn.CrossEntropyLoss,
    trainer/finetune_cls/ft_cls_trainer.py:66) on the ViT-Small and Swin-T hubs, B=2, 10 classes."""
    _ref()
    from model.finetune_cls.ft_cls_hub_model import finetune_cls_hub_model_small_patch16, finetune_cls_hub_model_swin_tiny_window7
    for tag, fac, bt in (("vit_small", finetune_cls_hub_model_small_patch16, "vit"), ("swin_tiny", finetune_cls_hub_model_swin_tiny_window7, "swin")):
        a = make_args(phase="finetune_cls", model_size="small" if bt == "vit" else "tiny", backbone_type=bt, num_classes=10, mask_ratio=0.0)
        hub = fac(a)
        det_fill_module_(hub)
        hub.train(True)
        x = det_normalish("ft.voxels", (2, 5, 224, 224)) * 0.5
        label = torch.tensor([3, 7])
        res = hub(x)
        pred, emb_h = res[-2], res[-3]
        loss = torch.nn.CrossEntropyLoss()(pred, label)
        loss.backward()
        out = dict(loss=loss.detach().double(), pred=pred.detach(), emb_h_checksums=checksums(emb_h), attn_checksums=checksums(res[-1]),
                   label=label)
        names, gn = [], []
        for n, p in hub.named_parameters():
            if p.grad is not None:
                names.append(n)
                gn.append(p.grad.double().norm().item())
        out["grad_names"], out["grad_norms"] = np.array(json.dumps(names)), np.array(gn)
        out["grad::classify_head.weight"] = hub.classify_head.weight.grad
        out["grad::classify_head.bias"] = hub.classify_head.bias.grad
        out["state_keys"] = np.array(json.dumps({k: list(v.shape) for k, v in hub.state_dict().items()}))
        save("ft_cls_" + tag, **out)


# --------------------------------------------------------------------------- density / anti-density masking
def gen_density():
    """ViT.random_masking of the reference itself with masking_strategy in {density, anti-density} (vit.py:80-103):
    (a) voxel grids the reference makes from synthetic clips (|sum over bins| is near-integer: near-ties are the norm),
    one of them with an event-free band (exact ties at 0); (b) bell-shaped random grids. The fixture keeps the clip
    seeds / generator names (the grids are re-made bit-exactly by the tests through the pinned voxel oracle), the
    reference's density noise and its ids. Also records whether the reference's ids equal a STABLE argsort of its noise
    (they do on this torch build: ties -> lower index first)."""
    _ref()
    torch.set_num_threads(1)
    from dataset.dataset_utils.events_to_voxel_grid import events_to_voxel_grid
    from model.backbone.vit import ViT
    out, cases = {}, []

    def grids_from_clips(seeds, n_ev, band=None):
        gs = []
        for sd in seeds:
            ev = synthetic_events(sd, n_ev)
            if band is not None:                      # drop the events of some patch rows: exact zero densities
                keep = ~((ev[:, 1] >= band[0]) & (ev[:, 1] < band[1]))
                ev = ev[keep]
            gs.append(events_to_voxel_grid(make_args(num_bins=5), ev.copy(), (224, 224)))
        return torch.stack([torch.as_tensor(g) for g in gs]).float()

    inputs = [("clips", dict(kind="clips", seeds=[300, 301], n_ev=100_000, band=None)),
              ("sparse", dict(kind="clips", seeds=[302, 303], n_ev=3_000, band=[64, 128])),
              ("randn", dict(kind="randn", name="density.randn", B=3))]
    for tag, spec in inputs:
        if spec["kind"] == "clips":
            x = grids_from_clips(spec["seeds"], spec["n_ev"], spec["band"])
        else:
            x = det_normalish(spec["name"], (spec["B"], 5, 224, 224)) * 0.5
        out[f"{tag}_x_checksums"] = checksums(x)
        for strat in ("density", "anti-density"):
            for ratio in (0.5, 0.75):
                a = make_args(mask_ratio=ratio, masking_strategy=strat)
                m = ViT(a, input_size=224, patch_size=16, embed_dim=32, depth=1, num_heads=1, mask_ratio=ratio)
                ids_keep, mask, ids_restore = m.random_masking(x)
                with torch.no_grad():
                    dens = torch.nn.AvgPool2d(16, 16)(abs(torch.sum(x, dim=1))).flatten(1)
                noise = dens if strat == "density" else -dens
                key = f"{tag}_{strat}_{int(ratio * 100)}"
                stable = torch.argsort(noise, dim=1, stable=True)
                out[key + "_noise"], out[key + "_ids_keep"] = noise, ids_keep
                out[key + "_mask"], out[key + "_ids_restore"] = mask, ids_restore
                srt = torch.sort(noise, dim=1).values
                n_ties = int((srt[:, 1:] == srt[:, :-1]).sum())
                eq = bool(torch.equal(stable[:, :ids_keep.shape[1]], ids_keep))
                cases.append(dict(key=key, tag=tag, strategy=strat, ratio=ratio, ties=n_ties, ref_equals_stable=eq))
                print(f"  {key}: ties={n_ties} reference ids == stable argsort: {eq}")
    out["inputs"] = np.array(json.dumps(dict(inputs)))
    out["cases"] = np.array(json.dumps(cases))
    save("masking_density", **out)
    torch.set_num_threads(8)


# --------------------------------------------------------------------------- bf16 autocast of the reference (CPU)
def gen_autocast():
    """SURVEY.md 8d parity gates: the reference's own step under torch.autocast("cpu", bfloat16) on the inputs of the
    rec_{tiny,small,base} fixtures -- what "the reference in bf16" gives, to report the HIP bf16 mode against (the fp32
    fixture stays the parity gate)."""
    _ref()
    out = {}
    for tag in ("tiny", "small", "base"):
        if tag == "small":
            from model.pretrain.pr_hub_model import pretrain_hub_model_small_patch16
            cfg = dict(input=224, patch=16, mask_ratio=0.5, B=2)
            a = make_args(model_size="small", pr_phase="rec")
            hub = pretrain_hub_model_small_patch16(a, emb_frames_dim=512, queue_length=1024, T=0.07)
            det_fill_module_(hub)
            hub.train(True)
            fwd = lambda x, y: hub(x, y, is_rec=True)
        else:
            cfg = CFGS[tag]
            a, hub = _compose(cfg)

            def fwd(x, y, hub=hub, a=a, cfg=cfg):
                emb_l1, emb_l2, emb_lh, mask, ids_restore = hub.backbone(x, mask=True)
                pred = hub.pretrain_rec_decoder(emb_lh, ids_restore)
                return (_rec_loss_ref(a, cfg["patch"], pred.float(), y, mask),)
        x, y, noise = _inputs(tag, cfg)
        real_rand = torch.rand
        torch.rand = lambda *s, **k: noise.clone()
        try:
            with torch.autocast("cpu", dtype=torch.bfloat16):
                loss = fwd(x, y)[0]
        finally:
            torch.rand = real_rand
        loss.float().backward()
        gn = float(np.sqrt(sum(p.grad.double().pow(2).sum().item() for p in hub.parameters() if p.grad is not None)))
        out[f"{tag}_loss"] = np.array(float(loss))
        out[f"{tag}_total_grad_norm"] = np.array(gn)
        print(f"  {tag}: autocast-bf16 loss {float(loss):.6f}, total grad norm {gn:.6f}")
    save("rec_autocast_bf16", **out)


# --------------------------------------------------------------------------- ViT-Base contrastive stage (config 3)
def gen_con_base():
    """PrHubModel.forward(is_rec=False) through the reference's pretrain_hub_model_base_patch16 (ViT-Base, 768-wide heads,
    BASELINE.json config 3) with the queue, B=2, queue_length=8 (a multiple of 8: the f32 path then reads the live queue
    buffer layout without padding). Same contents as con_small_queue."""
    _ref()
    from model.pretrain.pr_hub_model import pretrain_hub_model_base_patch16
    a = make_args(model_size="base", pr_phase="con", use_queue=True, mask_ratio=0.0)
    hub = pretrain_hub_model_base_patch16(a, emb_frames_dim=512, queue_length=8, T=0.07)
    det_fill_module_(hub)
    hub.train(True)
    x = det_normalish("conb.voxels", (2, 5, 224, 224)) * 0.5
    clip = det_normalish("conb.clip_emb", (2, 197, 512))
    q0 = hub.queue.clone()
    loss, emb_h_org, emb_h_proj, clip_org, clip_proj, attn = hub(x, clip)
    loss.backward()
    out = dict(loss=loss.detach().double(), emb_h_org_checksums=checksums(emb_h_org),
               emb_h_proj_checksums=checksums(emb_h_proj), clip_org_checksums=checksums(clip_org),
               clip_proj_checksums=checksums(clip_proj), attn_checksums=checksums(attn))
    names, gn = [], []
    for n, p in hub.named_parameters():
        if p.grad is not None:
            names.append(n)
            gn.append(p.grad.double().norm().item())
    out["grad_names"], out["grad_norms"] = np.array(json.dumps(names)), np.array(gn)
    out["queue_after_checksums"] = checksums(hub.queue)
    out["queue_ptr_after"] = hub.queue_ptr.clone()
    out["queue_changed"] = np.array(float((hub.queue - q0).abs().sum()))
    bn = {k: checksums(v) for k, v in hub.state_dict().items() if "running_" in k}
    out["bn_keys"] = np.array(json.dumps(list(bn.keys())))
    out["bn_checksums"] = np.stack(list(bn.values()))
    out["state_keys"] = np.array(json.dumps({k: list(v.shape) for k, v in hub.state_dict().items()}))
    save("con_base_queue", **out)
    # the 'adj' ("Trans") stage on the same inputs: every backbone parameter frozen except norm_layer
    # (main_pretrain.py:281-284); same forward, gradients only for the heads / CLIP branch / backbone.norm_layer
    hub2 = pretrain_hub_model_base_patch16(make_args(model_size="base", pr_phase="adj", use_queue=True, mask_ratio=0.0),
                                           emb_frames_dim=512, queue_length=8, T=0.07)
    det_fill_module_(hub2)
    hub2.train(True)
    for k, v in hub2.backbone.named_parameters():
        if "norm_layer" not in k:
            v.requires_grad = False
    loss2 = hub2(x, clip)[0]
    loss2.backward()
    names, gn = [], []
    for n, p in hub2.named_parameters():
        if p.grad is not None:
            names.append(n)
            gn.append(p.grad.double().norm().item())
    save("adj_base_queue", loss=loss2.detach().double(), grad_names=np.array(json.dumps(names)), grad_norms=np.array(gn),
         frozen=np.array(json.dumps([n for n, p in hub2.named_parameters() if not p.requires_grad])))


# --------------------------------------------------------------------------- Swin-Base (config 5 at its named width)
SWIN_BASE = dict(input=224, patch=32, dims=[128, 256, 512, 1024], depths=[2, 2, 18, 2], heads=[4, 8, 16, 32], window=7,
                 dec_dim=512, dec_depth=8, dec_heads=16, mask_ratio=0.5, B=2)


def gen_swin_base():
    """Swin-Base (depths 2-2-18-2, widths 128..1024, window 7) hand-composed from the reference CLASSES -- the
    reference ships only the Swin-T factory (swin.py:295-302) -- with a PrRecDecoder of the base width, masked
    reconstruction step, B=2, closed-form weights. Same contents as rec_swin_tiny (checksums instead of full tensors)."""
    _ref()
    from functools import partial
    from model.backbone.swin import SwinTransformer
    from model.pretrain.pr_rec_decoder import PrRecDecoder
    cfg = SWIN_BASE
    a = make_args(model_size="base", pr_phase="rec", backbone_type="swin")
    ln = partial(torch.nn.LayerNorm, eps=1e-6)
    bb = SwinTransformer(args=a, pretrain_img_size=224, patch_size=4, decoder_num_patches=49, num_bins=5, mask_ratio=0.5,
                         embed_dim=cfg["dims"], depths=cfg["depths"], num_heads=cfg["heads"], window_size=7, mlp_ratio=4.,
                         drop_rate=0., attn_drop_rate=0., drop_path_rate=0., norm_layer=ln)
    dec = PrRecDecoder(patch_size=32, num_patches=49, encoder_embed_dim=cfg["dims"], embed_dim=cfg["dec_dim"],
                       depth=cfg["dec_depth"], num_heads=cfg["dec_heads"], mlp_ratio=[4, 4, 4], norm_layer=ln, frame_chans=1)

    class Hub(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone, self.pretrain_rec_decoder = bb, dec

    hub = Hub()
    det_fill_module_(hub)
    hub.train(True)
    keep = {}

    def fwd(x, y):
        (emb_l1, emb_l2, emb_l3, emb_l4, emb_lh, c1, c2, c3, c4, mask, ids_restore, attn) = hub.backbone(x, mask=True)
        pred = hub.pretrain_rec_decoder(emb_lh, ids_restore)
        loss = _rec_loss_ref(a, 32, pred, y, mask)
        keep.update(emb_l3=emb_l3, emb_l4=emb_l4, attn=attn)
        return loss, emb_l1, emb_l2, emb_lh, pred, mask, ids_restore

    def extra(res):
        return dict(emb_l3_checksums=checksums(keep["emb_l3"]), emb_l4_checksums=checksums(keep["emb_l4"]),
                    attn_checksums=checksums(keep["attn"]), attn_shape=np.array(keep["attn"].shape))

    out = _run_rec("swinb", cfg, fwd, list(hub.named_parameters()), extra)
    out["cfg"] = np.array(json.dumps(cfg))
    out["state_keys"] = np.array(json.dumps({k: list(v.shape) for k, v in hub.state_dict().items()}))
    save("rec_swin_base", **out)


GENS = dict(voxel=gen_voxel, pos=gen_pos, mask=gen_mask, tiny=lambda: gen_composed("tiny"), small=gen_small,
            base=lambda: gen_composed("base"), train=gen_train, con=gen_con, convsmall=gen_convsmall, swin=gen_swin, swincon=gen_swincon, augment=gen_augment, evaug=gen_evaug, ftcls=gen_ftcls,
            frameaug=gen_frameaug, density=gen_density, autocast=gen_autocast, conbase=gen_con_base, swinbase=gen_swin_base,
            convbase=gen_convbase, chain=gen_chain, chainnim=gen_chain_nimagenet, trainbf16=gen_train_bf16)

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=",".join(GENS))
    ns = ap.parse_args()
    torch.set_num_threads(8)
    for k in ns.only.split(","):
        print(f"== {k}")
        GENS[k]()
