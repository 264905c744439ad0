"""Round-2 GPU parity / robustness tests (all through the C-ABI):
  * density / anti-density masking: noise bit-exact, ids equal to the reference's up to the order among tied values;
  * BASELINE.json config 3 at its named width: ViT-Base `con` step and the frozen-backbone `adj` ("Trans") step;
  * BASELINE.json config 5 at its named width: Swin-Base masked step;
  * ViT-Base bf16 step reported against the fp32 fixture and the reference's own bf16-autocast run;
  * verified sorted mode of the batched voxel entry point;
  * regression tests for the round-1 advisor findings (gradient-buffer leak, queue aliasing in f32 mode, pinned tables of
    captured graphs, capture warm-up side effects);
  * stage hand-off: a reference-layout checkpoint with `norm_l_h` keys loaded through remap_stage_checkpoint."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import assert_checksums, assert_ids_equal_up_to_ties, density_inputs, jl, rec_inputs

pytestmark = pytest.mark.gpu

F32_LOSS_RTOL = 1e-4
BF16_LOSS_RTOL = 2e-2


# ------------------------------------------------------------------------------------------------------- density masking
@pytest.mark.parametrize("backbone", ["vit", "convvit"])
def test_density_masking_ids_vs_reference(backbone):
    """vit.py:80-103 (copies convvit.py:85-124): the kernel's noise must equal the reference's bit for bit (it adds in
    AvgPool2d's order); ids equal up to the order among exactly tied values, bit for bit on tie-free rows."""
    from eventpretrain_amd.model.backbone import convvit, vit
    from eventpretrain_amd.testing import make_args
    d = load_golden("masking_density")
    for c in jl(d["cases"]):
        a = make_args(mask_ratio=c["ratio"], masking_strategy=c["strategy"], device="cuda", model_size="small")
        fac = vit.vit_small_patch16 if backbone == "vit" else convvit.convvit_small_patch16
        m = fac(args=a, num_bins=5, mask_ratio=c["ratio"], drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0).cuda()
        x = density_inputs(d, c["tag"]).cuda()
        key = c["key"]
        noise = m.masking_noise(x)
        assert np.array_equal(noise.cpu().numpy(), d[key + "_noise"]), (key, "density noise is not bit-exact")
        keep, mask, restore = m.random_masking(x)
        assert_ids_equal_up_to_ties(d[key + "_noise"], keep.shape[1],
                                    (d[key + "_ids_keep"], d[key + "_mask"], d[key + "_ids_restore"]),
                                    (keep.cpu().numpy(), mask.cpu().numpy(), restore.cpu().numpy()), key)


def test_density_noise_patch32_matches_oracle():
    """Swin hub geometry (49 cells of 32x32): same kernel, other window."""
    from eventpretrain_amd import ops
    from eventpretrain_amd.testing import det_normalish
    from oracle.model_oracle import density_noise
    x = det_normalish("density.p32", (2, 5, 224, 224)) * 0.5
    for strat, sign in (("density", 1.0), ("anti-density", -1.0)):
        got = ops.density_noise(x.cuda(), 32, sign).cpu()
        assert torch.equal(got, density_noise(x, 32, strat))


# ------------------------------------------------------------------------------------------------------- config 3: con / adj
def _con_base(phase):
    from eventpretrain_amd.model.pretrain import pr_hub_model as hub
    from eventpretrain_amd.testing import det_fill_module_, make_args
    a = make_args(model_size="base", pr_phase=phase, use_queue=True, mask_ratio=0.0, device="cuda")
    m = hub.pretrain_hub_model_base_patch16(a, emb_frames_dim=512, queue_length=8, T=0.07)
    det_fill_module_(m)
    return a, m.cuda().train()


def _con_inputs():
    from eventpretrain_amd.testing import det_normalish
    return (det_normalish("conb.voxels", (2, 5, 224, 224)) * 0.5).cuda(), det_normalish("conb.clip_emb", (2, 197, 512)).cuda()


def test_con_base_f32_matches_reference():
    """ViT-Base hub, `con` phase, queue length 8 (a multiple of 8: in f32 mode the GEMM operand used to ALIAS the live queue,
    which the enqueue overwrites before backward -- the gradient norms below catch that)."""
    from eventpretrain_amd import ops
    d = load_golden("con_base_queue")
    a, m = _con_base("con")
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == jl(d["state_keys"])
    x, clip = _con_inputs()
    ops.set_compute_dtype(torch.float32)
    loss, h_org, h_proj, c_org, c_proj, attn = m(x, clip)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(d["loss"])) <= F32_LOSS_RTOL * abs(float(d["loss"]))
    assert_checksums(h_org, d["emb_h_org_checksums"], 1e-4)
    assert_checksums(h_proj, d["emb_h_proj_checksums"], 2e-4)
    assert_checksums(c_org, d["clip_org_checksums"], 1e-4)
    assert_checksums(c_proj, d["clip_proj_checksums"], 1e-4)
    assert_checksums(attn.float(), d["attn_checksums"], 1e-4)
    params = dict(m.named_parameters())
    for n, gn in zip(jl(d["grad_names"]), d["grad_norms"]):
        assert params[n].grad is not None, n
        assert params[n].grad.double().norm().item() == pytest.approx(gn, rel=5e-3, abs=2e-6), n
    sd = m.state_dict()
    for k, cs in zip(jl(d["bn_keys"]), d["bn_checksums"]):
        assert_checksums(sd[k], cs, 1e-4, k)
    assert_checksums(m.queue, d["queue_after_checksums"], 1e-5)
    assert int(m.queue_ptr) == int(d["queue_ptr_after"][0])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_adj_stage_frozen_backbone(dtype):
    """`adj` ("Trans", main_pretrain.py:281-284): every backbone parameter except norm_layer frozen. Same loss as `con`;
    frozen parameters get no gradient, no deferred weight-gradient work is queued for them, and FusedAdamW leaves them (and
    their bf16 shadows) untouched while the heads move."""
    from eventpretrain_amd import ops
    from eventpretrain_amd.optim import FusedAdamW
    from eventpretrain_amd.utils import lr_decay as lrd
    d = load_golden("adj_base_queue")
    a, m = _con_base("adj")
    for k, v in m.backbone.named_parameters():
        if "norm_layer" not in k:
            v.requires_grad = False
    frozen = [n for n, p in m.named_parameters() if not p.requires_grad]
    assert sorted(frozen) == sorted(jl(d["frozen"]))
    opt = FusedAdamW(lrd.param_groups_lrd(a, m, 0.05, layer_decay=1), lr=1e-3, betas=(0.9, 0.95))
    x, clip = _con_inputs()
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    ops.set_compute_dtype(dtype)
    try:
        queued = []
        orig = ops._deferred.wgrad
        ops._deferred.wgrad = lambda param, *rest: (queued.append(param), orig(param, *rest))[1]
        try:
            loss = m(x, clip)[0]
            loss.backward()
        finally:
            ops._deferred.wgrad = orig
        torch.cuda.synchronize()
        frozen_ids = {id(p) for p in m.parameters() if not p.requires_grad}
        assert not any(id(p) in frozen_ids for p in queued), "a frozen weight was queued for the grouped weight-gradient launch"
        if dtype == torch.bfloat16:
            assert queued, "the trainable heads should use the deferred path in bf16 mode"
        rel = abs(loss.item() - float(d["loss"])) / abs(float(d["loss"]))
        assert rel <= (F32_LOSS_RTOL if dtype == torch.float32 else 5e-2), rel
        params = dict(m.named_parameters())
        assert all(params[n].grad is None for n in frozen)
        names = jl(d["grad_names"])
        assert sorted(n for n, p in params.items() if p.grad is not None) == sorted(names)
        if dtype == torch.float32:
            for n, gn in zip(names, d["grad_norms"]):
                assert params[n].grad.double().norm().item() == pytest.approx(gn, rel=5e-3, abs=2e-6), n
        opt.step()
        torch.cuda.synchronize()
        for n, p in m.named_parameters():
            if n in frozen:
                assert torch.equal(p.detach(), before[n]), n
                sh = getattr(p, "_evp_lp", None)
                if sh is not None:
                    assert torch.equal(sh, before[n].to(torch.bfloat16)), n
        moved = [n for n in names if not torch.equal(params[n].detach(), before[n])]
        assert len(moved) == len(names)
    finally:
        ops.set_compute_dtype(torch.float32)


# ------------------------------------------------------------------------------------------------------- config 5: Swin-Base
def test_swin_base_rec_f32_matches_reference():
    from eventpretrain_amd import ops
    from eventpretrain_amd.model.pretrain import pr_hub_model as hub
    from eventpretrain_amd.testing import det_fill_module_, make_args
    d = load_golden("rec_swin_base")
    cfg = jl(d["cfg"])
    a = make_args(model_size="base", pr_phase="rec", backbone_type="swin", device="cuda")
    m = hub.pretrain_hub_model_swin_base_patch16(a, emb_frames_dim=512, queue_length=1024, T=0.07)
    det_fill_module_(m)
    m = m.cuda().train()
    assert {k: list(v.shape) for k, v in m.state_dict().items()} == jl(d["state_keys"])
    x, y, noise = rec_inputs("swinb", cfg)
    ops.set_compute_dtype(torch.float32)
    out = m(x.cuda(), y.cuda(), is_rec=True, noise=noise.cuda())
    (loss, l1, l2, l3, l4, lh, c1, c2, c3, c4, pred, mask, restore, attn) = out
    loss.backward()
    torch.cuda.synchronize()
    assert np.array_equal(mask.cpu().numpy(), d["mask"]) and np.array_equal(restore.cpu().numpy(), d["ids_restore"])
    assert abs(loss.item() - float(d["loss"])) <= F32_LOSS_RTOL * abs(float(d["loss"]))
    assert list(attn.shape) == list(d["attn_shape"])
    for t, k in ((l1, "emb_l1"), (l2, "emb_l2"), (l3, "emb_l3"), (l4, "emb_l4"), (lh, "emb_lh"), (pred, "pred"), (attn.float(), "attn")):
        assert_checksums(t, d[k + "_checksums"], 1e-4, k)
    params = dict(m.named_parameters())
    for n, gn in zip(jl(d["grad_names"]), d["grad_norms"]):
        assert params[n].grad is not None, n
        assert params[n].grad.double().norm().item() == pytest.approx(gn, rel=5e-3, abs=2e-6), n


# ------------------------------------------------------------------------------------------------------- bf16 reporting
@pytest.mark.parametrize("tag", ["tiny", "small", "base"])
def test_rec_step_bf16_vs_fp32_and_autocast_fixtures(tag, capsys):
    """the bf16 throughput mode is REPORTED against the reference's fp32 loss and against the reference's own
    bf16-autocast loss (a stated bound, not the 1e-4 parity gate, which holds in f32 mode)."""
    from test_gpu_model import _run
    d, m, rel, (loss, *_rest) = _run(tag, torch.bfloat16)
    a = load_golden("rec_autocast_bf16")
    loss_fp32, loss_ac, got = float(d["loss"]), float(a[f"{tag}_loss"]), loss.item()
    rel32, relac = abs(got - loss_fp32) / loss_fp32, abs(got - loss_ac) / loss_ac
    ref_gap = abs(loss_ac - loss_fp32) / loss_fp32
    with capsys.disabled():
        print(f"\n[bf16 {tag}] loss {got:.6f}: rel err vs reference fp32 {rel32:.2e}, vs reference bf16-autocast {relac:.2e} "
              f"(reference autocast vs its own fp32: {ref_gap:.2e})")
    assert rel32 <= BF16_LOSS_RTOL and relac <= BF16_LOSS_RTOL
    tot = math.sqrt(sum(p.grad.double().pow(2).sum().item() for p in m.parameters() if p.grad is not None))
    assert abs(tot - float(d["total_grad_norm"])) / float(d["total_grad_norm"]) <= 5e-2


# ------------------------------------------------------------------------------------------------------- voxel: verified mode
def test_voxel_batch_verified_sorted_mode():
    """A batch whose clips 1 and 3 are NOT time-sorted (one shuffled, one with two rows swapped across a bin boundary) among
    sorted ones, through the default (verified) mode: every clip equals the oracle, i.e. the reference's result for its own
    t0/t1 rule (events_to_voxel_grid.py:15-22); and the unchecked "trust" mode really does go wrong on them (the check is live)."""
    from eventpretrain_amd.dataset.dataset_utils.events_to_voxel_grid import voxel_grid_batch
    from eventpretrain_amd.testing import synthetic_events
    from oracle.voxel_oracle import voxel_grid
    rng = np.random.default_rng(9)
    evs = [synthetic_events(400 + i, n, width=96, height=64) for i, n in enumerate([5000, 4000, 3000, 6000, 1, 0, 2500, 7000])]
    evs[5] = np.zeros((0, 4))
    evs[1] = evs[1][rng.permutation(evs[1].shape[0])]
    evs[3][[100, 5000]] = evs[3][[5000, 100]]
    ev = torch.from_numpy(np.concatenate(evs)).cuda()
    off = torch.tensor(np.concatenate([[0], np.cumsum([e.shape[0] for e in evs])]), dtype=torch.int64).cuda()
    ref = [voxel_grid(e, 5, (64, 96)) if e.shape[0] else np.zeros((5, 64, 96), np.float32) for e in evs]
    for tr in (0, 9):
        g = voxel_grid_batch(ev, off, 5, (64, 96), tile_rows=tr).cpu().numpy()
        for i in range(len(evs)):
            assert np.abs(g[i] - ref[i]).max() <= 1e-5, (tr, i)
    g0 = voxel_grid_batch(ev, off, 5, (64, 96), assume_sorted=False).cpu().numpy()
    gt = voxel_grid_batch(ev, off, 5, (64, 96), assume_sorted="trust").cpu().numpy()
    for i in range(len(evs)):
        assert np.abs(g0[i] - ref[i]).max() <= 1e-5, i
    assert max(np.abs(gt[i] - ref[i]).max() for i in (1, 3)) > 1e-3
    for i in (0, 2, 4, 5, 6, 7):
        assert np.abs(gt[i] - ref[i]).max() <= 1e-5, i


# ------------------------------------------------------------------------------------------------------- advisor regressions
def _tiny_setup(B=2, seed_fill=True):
    from eventpretrain_amd.model.pretrain import pr_hub_model as hub
    from eventpretrain_amd.optim import FusedAdamW
    from eventpretrain_amd.testing import det_fill_module_, det_normalish, make_args
    from eventpretrain_amd.utils import lr_decay as lrd
    a = make_args(model_size="tiny", pr_phase="rec", device="cuda", lr=1e-3)
    m = hub.pretrain_hub_model_tiny_patch16_64(a, emb_frames_dim=512, queue_length=8, T=0.07)
    if seed_fill:
        det_fill_module_(m)
    m = m.cuda().train()
    opt = FusedAdamW(lrd.param_groups_lrd(a, m, 0.05, layer_decay=1), lr=1e-3, betas=(0.9, 0.95))
    x = (det_normalish("r2.voxels", (B, 5, 64, 64)) * 0.5).cuda()
    y = det_normalish("r2.sub_frame", (B, 1, 64, 64)).cuda()
    return a, m, opt, x, y


def test_eager_bf16_steps_do_not_leak_gradient_buffers():
    """Without a data-parallel reducer nothing consumes the flat gradient buffers of the deferred launches: they must not be
    retained (round 1 pinned one full set per step)."""
    from eventpretrain_amd import ops
    a, m, opt, x, y = _tiny_setup()
    ops.set_compute_dtype(torch.bfloat16)
    try:
        g = torch.Generator(device="cuda").manual_seed(3)
        mem = []
        for s in range(24):
            noise = torch.rand(2, 16, device="cuda", generator=g)
            m(x, y, is_rec=True, noise=noise)[0].backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            torch.cuda.synchronize()
            mem.append(torch.cuda.memory_allocated())
        assert len(ops._deferred.flat_buffers) == 0
        assert mem[-1] == mem[4], (mem[4], mem[-1])
    finally:
        ops.set_compute_dtype(torch.float32)


def test_capture_leaves_training_state_untouched_and_matches_eager():
    """GraphedStep's warm-up steps must not move weights, moments, step counter, buffers or the noise stream; afterwards
    graph replays follow the eager trajectory from the same start."""
    from eventpretrain_amd import ops
    from eventpretrain_amd.engine import GraphedStep
    ops.set_compute_dtype(torch.bfloat16)
    try:
        fwd = lambda mm, xx, yy, noise: mm(xx, yy, is_rec=True, noise=noise)
        runs = {}
        for mode in ("graph", "eager"):
            a, m, opt, x, y = _tiny_setup()
            before = {n: p.detach().clone() for n, p in m.named_parameters()}
            gen = torch.Generator(device="cuda").manual_seed(11)
            gstate = gen.get_state().clone()
            ex = GraphedStep(m, opt, fwd, [x, y], noise_shape=(2, 16), generator=gen, use_graph=(mode == "graph"), warmup=3)
            if mode == "graph":
                assert ex.note.startswith("hip-graph"), ex.note
                for n, p in m.named_parameters():
                    assert torch.equal(p.detach(), before[n]), n
                    sh = getattr(p, "_evp_lp", None)
                    assert sh is None or torch.equal(sh, before[n].to(torch.bfloat16)), n
                assert opt._step == 0
                assert all(float(st["exp_avg"].abs().max()) == 0.0 and float(st["exp_avg_sq"].abs().max()) == 0.0
                           for st in opt.state.values() if "exp_avg" in st)
                assert torch.equal(ex.gen.get_state(), gstate)
            runs[mode] = [ex.step().item() for _ in range(4)]
        assert runs["graph"] == pytest.approx(runs["eager"], rel=2e-3)
        assert runs["graph"][0] == pytest.approx(runs["eager"][0], rel=1e-5)     # first step: identical weights
    finally:
        ops.set_compute_dtype(torch.float32)


def test_captured_graph_survives_later_eager_steps_and_second_capture():
    """The grouped launches of a captured step read their problem tables from pinned buffers the graph re-copies at every
    replay; an eager bf16 step (or another capture) used to rewrite those shared buffers. Replays of graph A must keep
    following A's own eager trajectory while other models step in between."""
    from eventpretrain_amd import ops
    from eventpretrain_amd.engine import GraphedStep
    ops.set_compute_dtype(torch.bfloat16)
    try:
        fwd = lambda mm, xx, yy, noise: mm(xx, yy, is_rec=True, noise=noise)
        a, m, opt, x, y = _tiny_setup()
        ref = GraphedStep(m, opt, fwd, [x, y], noise_shape=(2, 16), generator=torch.Generator(device="cuda").manual_seed(5), use_graph=False)
        want = [ref.step().item() for _ in range(4)]
        del ref, m, opt
        a, mA, optA, x, y = _tiny_setup()
        exA = GraphedStep(mA, optA, fwd, [x, y], noise_shape=(2, 16), generator=torch.Generator(device="cuda").manual_seed(5), warmup=2)
        assert exA.note.startswith("hip-graph"), exA.note
        got = [exA.step().item()]
        # another model, other batch size (other table contents), stepping eagerly, then captured as well
        aB, mB, optB, xB, yB = _tiny_setup(B=4)
        exB = GraphedStep(mB, optB, fwd, [xB, yB], noise_shape=(4, 16), generator=torch.Generator(device="cuda").manual_seed(6), use_graph=False)
        exB.step()
        got.append(exA.step().item())
        exB2 = GraphedStep(mB, optB, fwd, [xB, yB], noise_shape=(4, 16), generator=torch.Generator(device="cuda").manual_seed(7), warmup=2)
        exB2.step()
        got.append(exA.step().item())
        exB.step()
        got.append(exA.step().item())
        torch.cuda.synchronize()
        assert got == pytest.approx(want, rel=1e-5), (got, want)
    finally:
        ops.set_compute_dtype(torch.float32)


def test_host_running_ahead_of_the_device_keeps_each_steps_own_lr():
    """The graph's H2D nodes read the optimizer's pinned lr / bias-correction tables when a replay RUNS. With a different
    learning rate every step, 12 steps queued back to back (host far ahead of the device) must end with bit-identical weights
    to the same 12 steps with a device synchronisation after each."""
    from eventpretrain_amd import ops
    from eventpretrain_amd.engine import GraphedStep
    ops.set_compute_dtype(torch.bfloat16)
    try:
        fwd = lambda mm, xx, yy, noise: mm(xx, yy, is_rec=True, noise=noise)
        finals = []
        for lockstep in (True, False):
            a, m, opt, x, y = _tiny_setup()
            ex = GraphedStep(m, opt, fwd, [x, y], noise_shape=(2, 16), generator=torch.Generator(device="cuda").manual_seed(3), warmup=2)
            assert ex.note.startswith("hip-graph"), ex.note
            torch.cuda.synchronize()
            if not lockstep:                   # keep the device busy so that the host really is several replays ahead
                big = torch.empty(64 << 20, device="cuda")
                for _ in range(40):
                    big.normal_()
            for i in range(12):
                for g in opt.param_groups:
                    g["lr"] = 1e-3 * (1 + i) * g.get("lr_scale", 1.0)
                ex.step()
                if lockstep:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            finals.append([p.detach().clone() for p in m.parameters()])
        for p0, p1 in zip(*finals):
            assert torch.equal(p0, p1)
    finally:
        ops.set_compute_dtype(torch.float32)


# ------------------------------------------------------------------------------------------------------- stage hand-off
def test_stage_checkpoint_with_old_norm_keys_reproduces_con_fixture(tmp_path):
    """main_pretrain.py:265-279: a checkpoint of the MM stage names the backbone's final norm `norm_l_h` (and carries the
    decoder); loaded through remap_stage_checkpoint + load_state_dict(strict=False) into the `con` hub it must give the `con`
    fixture's loss (the fixture's weights are the same closed-form fill)."""
    from eventpretrain_amd import ops
    from eventpretrain_amd.testing import det_value_for
    from eventpretrain_amd.utils.misc import remap_stage_checkpoint
    d = load_golden("con_base_queue")
    a, m = _con_base("con")
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    ckpt = {}
    for k, v in sd.items():
        if k.startswith("backbone.norm_layer."):
            ckpt[k.replace("norm_layer", "norm_h")] = v          # `con` hand-off spelling (main_pretrain.py:272-275)
        elif k.startswith("backbone."):
            ckpt[k] = v
    ckpt["pretrain_rec_decoder.mask_token"] = det_value_for("pretrain_rec_decoder.mask_token", (1, 1, 512))   # ignored by the con hub
    path = tmp_path / "checkpoint_mm.pth"
    torch.save({"model": ckpt, "epoch": 3}, path)
    # scramble the backbone, then load
    with torch.no_grad():
        for p in m.backbone.parameters():
            p.mul_(0.5)
    loaded = torch.load(path, map_location="cpu")["model"]
    msg = m.load_state_dict(remap_stage_checkpoint(loaded, "con"), strict=False)
    assert not [k for k in msg.missing_keys if k.startswith("backbone.")], msg.missing_keys
    assert msg.unexpected_keys == ["pretrain_rec_decoder.mask_token"]
    x, clip = _con_inputs()
    ops.set_compute_dtype(torch.float32)
    loss = m(x, clip)[0]
    assert abs(loss.item() - float(d["loss"])) <= F32_LOSS_RTOL * abs(float(d["loss"]))


# ------------------------------------------------------------------------------------------------------- Swin: host-side noise
def test_swin_host_noise_equals_device_noise():
    """Mask noise given as a CPU tensor: the window plan is made from it on the host (no device->host read-back), the device
    receives the same noise -- loss, mask and ids equal those of the device-noise path."""
    from eventpretrain_amd import ops
    from eventpretrain_amd.model.pretrain import pr_hub_model as hub
    from eventpretrain_amd.testing import det_fill_module_, det_uniform, make_args
    d = load_golden("rec_swin_tiny")
    cfg = jl(d["cfg"])
    a = make_args(model_size="tiny", pr_phase="rec", backbone_type="swin", device="cuda")
    m = hub.pretrain_hub_model_swin_tiny_patch16(a, emb_frames_dim=512, queue_length=1024, T=0.07)
    det_fill_module_(m)
    m = m.cuda().train()
    x, y, noise = rec_inputs("swin", cfg)
    ops.set_compute_dtype(torch.float32)
    with torch.no_grad():
        o_dev = m(x.cuda(), y.cuda(), is_rec=True, noise=noise.cuda())
        o_host = m(x.cuda(), y.cuda(), is_rec=True, noise=noise)
    assert o_dev[0].item() == o_host[0].item()
    assert torch.equal(o_dev[11], o_host[11]) and torch.equal(o_dev[12], o_host[12])
    assert abs(o_host[0].item() - float(d["loss"])) <= F32_LOSS_RTOL * abs(float(d["loss"]))
    # default draw (random strategy): host noise, runs and differs from call to call
    with torch.no_grad():
        l1, l2 = m(x.cuda(), y.cuda(), is_rec=True)[0].item(), m(x.cuda(), y.cuda(), is_rec=True)[0].item()
    assert math.isfinite(l1) and math.isfinite(l2) and l1 != l2


# ------------------------------------------------------------------------------------------------------- frame_augment
def test_frame_augment_kernel_vs_reference():
    """evp_frame_augment_f32 (crop -> bicubic -> h-flip -> negate on time flip) against the reference's own frame_augment
    outputs with the decisions drawn from RandomState(seed) in the reference's order: crop / flip placement exact, the bicubic
    values within 1e-5 (f32 cubic-convolution sums in another order than ATen's), and against the oracle on random boxes."""
    from eventpretrain_amd.dataset.augmentation.view_augment import draw_evg_params, frame_augment_batch
    from eventpretrain_amd.testing import det_normalish
    from oracle import augment_oracle as ao
    d = load_golden("frame_augment")
    for tag in jl(d["tags"]):
        shp, S, seed = tuple(int(v) for v in d[f"{tag}_shape"]), int(d[f"{tag}_size"]), int(d[f"{tag}_seed"])
        f = det_normalish(f"aug.frame.{tag}", shp)
        prm = draw_evg_params(np.random.RandomState(seed), shp[1], shp[2], 0.8)
        assert prm[5] == int(d[f"{tag}_tflip"])
        out = frame_augment_batch(f.unsqueeze(0).cuda(), np.array([prm]), (S, S)).cpu().numpy()[0]
        if f"{tag}_out" in d.files:
            assert np.abs(out - d[f"{tag}_out"]).max() <= 1e-5, tag
        else:
            assert np.abs(out.reshape(-1)[::7] - d[f"{tag}_sample"]).max() <= 1e-5, tag
    rng = np.random.default_rng(4)
    B, C, H, W = 8, 1, 120, 160
    x = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32))
    prm = np.zeros((B, 6), dtype=np.int32)
    for i in range(B):
        w, h = int(rng.integers(2, W + 1)), int(rng.integers(2, H + 1))
        prm[i] = (int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1)), w, h, i & 1, (i >> 1) & 1)
    for size in ((224, 224), (48, 80)):
        out = frame_augment_batch(x.cuda(), prm, size).cpu().numpy()
        for i in range(B):
            assert np.abs(out[i] - ao.frame_transform(x[i].numpy(), tuple(prm[i]), size)).max() <= 2e-5, (size, i)


# ------------------------------------------------------------------------------------------------------- multi-GPU form, con phase
def test_overlapped_data_parallel_con_step_matches_single_rank_graph():
    """The N-rank form of the graphed step (forward+backward graph -> chunked weight gradients interleaved with RCCL all-reduces
    -> AdamW per reduced buffer) for the CONTRASTIVE stage (MoCo heads with BatchNorm, queue InfoNCE + enqueue inside the graph)
    on a one-rank RCCL group: same losses, queue and pointer as the single-graph form."""
    import os
    import torch.distributed as dist
    from eventpretrain_amd import ops
    from eventpretrain_amd.engine import GraphedStep
    from eventpretrain_amd.model.pretrain import pr_hub_model as hub
    from eventpretrain_amd.optim import FusedAdamW
    from eventpretrain_amd.parallel import BucketedGradReducer
    from eventpretrain_amd.testing import det_fill_module_, det_normalish, make_args
    from eventpretrain_amd.utils import lr_decay as lrd
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29542")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    results = []
    ops.set_compute_dtype(torch.bfloat16)
    try:
        x = (det_normalish("dpc.voxels", (32, 5, 224, 224)) * 0.5).cuda()
        clip = det_normalish("dpc.clip", (32, 197, 512)).cuda()
        for multi in (False, True):
            a = make_args(model_size="small", pr_phase="con", use_queue=True, mask_ratio=0.0, device="cuda")
            m = hub.pretrain_hub_model_small_patch16(a, emb_frames_dim=512, queue_length=64, T=0.07)
            det_fill_module_(m)
            m = m.cuda().train()
            opt = FusedAdamW(lrd.param_groups_lrd(a, m, a.weight_decay, layer_decay=1), lr=1e-4, betas=(0.9, 0.95))
            red = BucketedGradReducer.for_module(m) if multi else None
            ex = GraphedStep(m, opt, lambda mm, xx, cc, noise: mm(xx, cc), [x, clip], noise_shape=None, reducer=red, warmup=2, wgrad_chunks=3)
            assert ex.note.startswith("hip-graph"), ex.note
            losses = [ex.step().item() for _ in range(3)]
            torch.cuda.synchronize()
            results.append((losses, m.queue.detach().clone(), int(m.queue_ptr)))
    finally:
        ops.set_compute_dtype(torch.float32)
        if created:
            dist.destroy_process_group()
    (l0, q0, p0), (l1, q1, p1) = results
    assert l0[0] == pytest.approx(l1[0], rel=1e-6) and l0 == pytest.approx(l1, rel=2e-4), (l0, l1)
    assert p0 == p1 == (3 * 32) % 64
    assert (q0 - q1).norm().item() <= 2e-3 * q0.norm().item()


# ------------------------------------------------------------------------------------------------------- ConvBlock drop-in call
def test_conv_block_forward_with_dense_keep_map():
    """ConvBlock.forward(x, mask) with the reference's (B,1,H,W) keep map (conv_block.py:41-51) against the oracle."""
    from eventpretrain_amd import ops
    from eventpretrain_amd.model.sub_module.conv_block import ConvBlock
    from eventpretrain_amd.testing import det_fill_module_, det_normalish
    from oracle import model_oracle as mo
    blk = ConvBlock(input_size=64, kernel_size=5, mlp_ratio=4.)
    det_fill_module_(blk)
    blk = blk.cuda()
    x = det_normalish("cb.x", (2, 64, 28, 28))
    keep = (det_normalish("cb.keep", (2, 1, 7, 7)) > 0).float().repeat_interleave(4, 2).repeat_interleave(4, 3)
    ops.set_compute_dtype(torch.float32)
    sd = {"b." + k: v.detach().cpu() for k, v in blk.state_dict().items()}
    for m in (None, keep):
        got = blk(x.cuda(), None if m is None else m.cuda()).cpu()
        ref = mo.conv_block(sd, "b.", x, m)
        assert torch.allclose(got, ref, atol=2e-5, rtol=1e-5), (m is None, (got - ref).abs().max().item())


# ------------------------------------------------------------------------------------------------------- con epoch loops
def test_con_epoch_loops_run_and_agree():
    """pr_con_one_epoch (stored CLIP tokens) and pr_con_n_one_epoch (tokens from clip_model.encode_image on the fly) on the same
    two batches from the same start: same per-epoch statistics; the queue pointer has advanced by 2 x B."""
    from types import SimpleNamespace
    from eventpretrain_amd import ops
    from eventpretrain_amd.model.pretrain import pr_hub_model as hub
    from eventpretrain_amd.optim import FusedAdamW
    from eventpretrain_amd.testing import det_fill_module_, det_normalish, make_args
    from eventpretrain_amd.trainer.pretrain.pr_trainer import pr_con_n_one_epoch, pr_con_one_epoch
    from eventpretrain_amd.utils import lr_decay as lrd
    from eventpretrain_amd.utils.misc import NativeScalerWithGradNormCount
    ops.set_compute_dtype(torch.float32)
    xs = [det_normalish(f"ce.vox.{i}", (2, 5, 224, 224)) * 0.5 for i in range(2)]
    clips = [det_normalish(f"ce.clip.{i}", (2, 197, 512)) for i in range(2)]
    images = [torch.full((2, 3, 8, 8), float(i)) for i in range(2)]
    clip_model = SimpleNamespace(encode_image=lambda im: clips[int(im[0, 0, 0, 0].item())].to(im.device))
    stats = []
    for variant in ("con", "con-n"):
        a = make_args(model_size="small", pr_phase=variant, use_queue=True, mask_ratio=0.0, device="cuda", lr=1e-4, min_lr=1e-4,
                      warmup_epochs=0, epochs=1, accum_iter=1, print_freq=100, backward=True, visualize=False, test_experiment=False)
        m = hub.pretrain_hub_model_small_patch16(a, emb_frames_dim=512, queue_length=8, T=0.07)
        det_fill_module_(m)
        m = m.cuda().train()
        opt = FusedAdamW(lrd.param_groups_lrd(a, m, 0.05, layer_decay=1), lr=1e-4, betas=(0.9, 0.95))
        if variant == "con":
            loader = [dict(events_voxel_grid=xs[i], clip_emb=clips[i], image_name=["n"] * 2) for i in range(2)]
            st = pr_con_one_epoch(a, m, loader, opt, 0, NativeScalerWithGradNormCount())
        else:
            loader = [dict(events_voxel_grid=xs[i], image=images[i], image_name=["n"] * 2) for i in range(2)]
            st = pr_con_n_one_epoch(a, m, None, clip_model, loader, opt, 0, NativeScalerWithGradNormCount())
        assert int(m.queue_ptr) == 4
        stats.append(st)
    assert stats[0]["contrastive_loss"] == pytest.approx(stats[1]["contrastive_loss"], rel=1e-6)
    assert math.isfinite(stats[0]["contrastive_loss"])


# ------------------------------------------------------------------------------------------- accumulators under replay
def test_graph_replay_clears_its_atomic_accumulators():
    """Outputs that kernels accumulate into with atomics (split-K weight gradient, column sums, the relative-position table
    gradient) must be cleared by something that re-runs on every HIP-graph replay. They used to be cleared by hipMemsetAsync,
    whose graph nodes were not reliable on replay (non-finite Swin gradients on the second replay); now a kernel does it.
    Here the captured outputs are poisoned with NaN between replays."""
    from eventpretrain_amd import ops
    from eventpretrain_amd.ops import call, dt, ptr, stream_ptr
    torch.manual_seed(0)
    M, n_out, k_in = 3528, 96, 384                      # few output tiles, long K: the launcher splits K and adds with atomics
    dy = torch.randn(M, n_out, device="cuda")
    x = torch.randn(M, k_in, device="cuda")
    Bg, nG, N, H, R = 4, 2, 49, 3, 169
    qkv = torch.randn(Bg * N, 3 * H * 32, device="cuda") * 0.3
    table = torch.randn(R, H, device="cuda") * 0.1
    rel = torch.randint(-1, R, (nG, N, N), device="cuda", dtype=torch.int32)
    dout = torch.randn(Bg * N, H * 32, device="cuda")

    def work():
        dw = ops._wgrad(dy, x, n_out, k_in, M)
        cs = ops.colsum(dy)
        att = torch.empty(Bg * N, H * 32, device="cuda")
        call("evp_window_attention_fwd", ptr(qkv), ptr(table), ptr(rel), ptr(att), 0, Bg, nG, N, H, R, 32 ** -0.5, dt(qkv), None, 1.0, stream_ptr())
        dqkv = torch.empty_like(qkv)
        dtable = torch.empty(R, H, device="cuda")
        call("evp_window_attention_bwd", ptr(qkv), ptr(table), ptr(rel), ptr(att), ptr(dout), ptr(dqkv), ptr(dtable), Bg, nG, N, H, R,
             32 ** -0.5, dt(qkv), None, 1.0, stream_ptr())
        return dw, cs, dtable

    want = [t.clone() for t in work()]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        work()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        outs = work()
    for rep in range(3):
        for t in outs:
            t.fill_(float("nan"))
        g.replay()
        torch.cuda.synchronize()
        for t, w, name in zip(outs, want, ("split-K weight gradient", "column sums", "relative-position table gradient")):
            assert torch.isfinite(t).all(), (name, rep)
            assert torch.allclose(t, w, rtol=1e-4, atol=1e-4 * w.abs().max().item()), (name, rep)
