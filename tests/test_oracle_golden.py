"""Pins the oracle (oracle/*.py, oracle/voxel_oracle.c) against fixtures produced by the reference itself
(oracle/gen_golden.py). CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import assert_checksums, checksums, jl, rec_inputs, rec_state_dict
from oracle import model_oracle as mo
from oracle.voxel_oracle import voxel_grid


def test_voxel_kat_exact():
    d = load_golden("voxel")
    g = voxel_grid(d["kat_events"], 5, (4, 4))
    assert np.array_equal(g, d["kat_grid"])
    nz = {(int(b), int(y), int(x)): float(g[b, y, x]) for b, y, x in zip(*np.nonzero(g))}
    # SURVEY.md section 4 known answers
    assert nz == pytest.approx({(0, 0, 0): 1.0, (1, 0, 1): -1.0, (2, 1, 2): 1.0, (3, 3, 3): -0.4, (4, 3, 3): 0.4}, abs=1e-6)


def test_voxel_cases_bit_exact():
    d = load_golden("voxel")
    for c in jl(d["cases"]):
        g = voxel_grid(d[c["tag"] + "_events"], c["bins"], (c["H"], c["W"]), c["is_txyp"])
        assert np.array_equal(g, d[c["tag"] + "_grid"]), c["tag"]


def test_voxel_full_clip():
    from eventpretrain_amd.testing import synthetic_events
    d = load_golden("voxel")
    g = voxel_grid(synthetic_events(0), 5, (224, 224))
    assert np.array_equal(g[:, ::7, ::5], d["full0_sample"])
    assert np.allclose(checksums(torch.from_numpy(g)), d["full0_checksums"], rtol=1e-12)


def test_pos_embed():
    d = load_golden("pos_embed")
    for dim, g in [(64, 4), (128, 4), (192, 4), (384, 14), (512, 14), (768, 14), (256, 7)]:
        t = torch.from_numpy(mo.sincos_2d(dim, g)).float()
        assert mo.sincos_2d(dim, g).dtype == np.float32
        assert np.allclose(checksums(t), d[f"d{dim}_g{g}_checksums"], rtol=0, atol=0), (dim, g)
        if f"d{dim}_g{g}_table" in d.files:
            assert np.array_equal(t.numpy(), d[f"d{dim}_g{g}_table"])
        else:
            assert np.array_equal(t[[0, 1, 17, g * g - 1]].numpy(), d[f"d{dim}_g{g}_rows"])
    # first half encodes w
    t = mo.sincos_2d(384, 14)
    assert np.allclose(t[17, :4], [0.14112, 0.40414152, 0.6173581, 0.7782725], atol=1e-6)
    assert np.allclose(t[17, 192:196], [0.841471, 0.78859305, 0.73482203, 0.68156135], atol=1e-6)


def test_masking_bit_exact():
    d = load_golden("masking")
    for c in jl(d["cases"]):
        t = c["tag"]
        keep, mask, restore = mo.masking_from_noise(torch.from_numpy(d[t + "_noise"]), c["ratio"])
        assert np.array_equal(keep.numpy(), d[t + "_ids_keep"])
        assert np.array_equal(restore.numpy(), d[t + "_ids_restore"])
        assert np.array_equal(mask.numpy(), d[t + "_mask"])


def _check_rec(tag, rtol_loss=2e-6, rtol_cs=2e-5):
    d = load_golden(f"rec_{tag}")
    cfg = jl(d["cfg"])
    sd = {k: v.requires_grad_(v.is_floating_point() and "pos_embed" not in k) for k, v in rec_state_dict(cfg).items()}
    x, y, noise = rec_inputs(tag, cfg)
    assert np.array_equal(noise.numpy(), d["noise"])
    loss, l1, l2, lh, pred, mask, restore = mo.rec_step(sd, x, y, noise, cfg)
    assert np.array_equal(mask.numpy(), d["mask"])
    assert np.array_equal(restore.numpy(), d["ids_restore"])
    assert abs(loss.item() - float(d["loss"])) <= rtol_loss * abs(float(d["loss"]))
    assert_checksums(pred, d["pred_checksums"], rtol_cs, "pred")
    assert_checksums(lh, d["emb_lh_checksums"], rtol_cs, "emb_lh")
    assert_checksums(l1, d["emb_l1_checksums"], rtol_cs, "emb_l1")
    assert_checksums(l2, d["emb_l2_checksums"], rtol_cs, "emb_l2")
    loss.backward()
    names = jl(d["grad_names"])
    for n, gn, ws in zip(names, d["grad_norms"], d["grad_wsums"]):
        g = sd[n].grad
        assert g is not None, n
        assert abs(g.double().norm().item() - gn) <= 1e-4 * gn + 1e-9, n
    return d, sd, (loss, l1, l2, lh, pred)


def test_rec_tiny_matches_reference():
    d, sd, (loss, l1, l2, lh, pred) = _check_rec("tiny")
    assert torch.allclose(pred, torch.from_numpy(d["pred"]), atol=2e-5, rtol=1e-5)
    assert torch.allclose(lh, torch.from_numpy(d["emb_lh"]), atol=2e-5, rtol=1e-5)
    for k in d.files:
        if k.startswith("grad::"):
            ref = torch.from_numpy(d[k])
            assert torch.allclose(sd[k[6:]].grad, ref, atol=1e-6 + 1e-4 * ref.abs().max().item(), rtol=1e-4), k


def test_rec_small_matches_reference():
    _check_rec("small")


def test_rec_small_state_keys():
    d = load_golden("rec_small")
    cfg = jl(d["cfg"])
    ref = jl(d["state_keys"])
    sd = rec_state_dict(cfg)
    assert {k: list(v.shape) for k, v in sd.items()} == ref


def test_rec_convvit_small_matches_reference():
    """ConvViT-S (BASELINE config 4 at the size the reference's hub factory builds) through the oracle."""
    from eventpretrain_amd.testing import det_value_for
    d = load_golden("rec_convsmall")
    cfg = jl(d["cfg"])
    sd = {}
    for k, shp in jl(d["state_keys"]).items():
        if k.endswith("pos_embed"):
            sd[k] = torch.from_numpy(mo.sincos_2d(shp[-1], 14)).float().unsqueeze(0)
        else:
            sd[k] = det_value_for(k, shp).requires_grad_(True)
    x, y, noise = rec_inputs("convsmall", dict(B=2, input=224, patch=16))
    loss, l1, l2, lh, pred, mask, restore = mo.convvit_rec_step(sd, x, y, noise, cfg)
    assert np.array_equal(mask.numpy(), d["mask"]) and np.array_equal(restore.numpy(), d["ids_restore"])
    assert abs(loss.item() - float(d["loss"])) <= 2e-6 * abs(float(d["loss"]))
    assert_checksums(l1, d["emb_l1_checksums"], 2e-5)
    assert_checksums(l2, d["emb_l2_checksums"], 2e-5)
    assert_checksums(lh, d["emb_lh_checksums"], 2e-5)
    assert_checksums(pred, d["pred_checksums"], 2e-5)
    loss.backward()
    for n, gn in zip(jl(d["grad_names"]), d["grad_norms"]):
        assert sd[n].grad is not None, n
        assert abs(sd[n].grad.double().norm().item() - gn) <= 2e-4 * gn + 1e-9, n


def test_rec_convvit_base_matches_reference():
    """ConvViT-Base (the size bench.py --config convvit_base_rec times; reference convvit.py:218-224 + pr_rec_decoder.py:89-95)
    through the oracle: loss, ids, taps, every gradient norm."""
    from eventpretrain_amd.testing import det_value_for
    d = load_golden("rec_convbase")
    cfg = jl(d["cfg"])
    sd = {}
    for k, shp in jl(d["state_keys"]).items():
        if k.endswith("pos_embed"):
            sd[k] = torch.from_numpy(mo.sincos_2d(shp[-1], 14)).float().unsqueeze(0)
        else:
            sd[k] = det_value_for(k, shp).requires_grad_(True)
    x, y, noise = rec_inputs("convbase", dict(B=2, input=224, patch=16))
    loss, l1, l2, lh, pred, mask, restore = mo.convvit_rec_step(sd, x, y, noise, cfg)
    assert np.array_equal(mask.numpy(), d["mask"]) and np.array_equal(restore.numpy(), d["ids_restore"])
    assert abs(loss.item() - float(d["loss"])) <= 2e-6 * abs(float(d["loss"]))
    assert_checksums(l1, d["emb_l1_checksums"], 2e-5)
    assert_checksums(lh, d["emb_lh_checksums"], 2e-5)
    assert_checksums(pred, d["pred_checksums"], 2e-5)
    loss.backward()
    for n, gn in zip(jl(d["grad_names"]), d["grad_norms"]):
        assert sd[n].grad is not None, n
        assert abs(sd[n].grad.double().norm().item() - gn) <= 2e-4 * gn + 1e-9, n


def swin_state_dict(d):
    from eventpretrain_amd.testing import det_value_for
    sd = {}
    for k, shp in jl(d["state_keys"]).items():
        if k.endswith("relative_position_index"):
            continue                                   # unused at pre-training (swin_block.py:100)
        if k.endswith("pos_embed"):
            sd[k] = torch.from_numpy(mo.sincos_2d(shp[-1], 7)).float().unsqueeze(0)
        else:
            sd[k] = det_value_for(k, shp).requires_grad_(True)
    return sd


def test_swin_knapsack_grouping():
    """Host grouping known answers (hand-checked against swin_block.py:277-347)."""
    assert mo.swin_knapsack(10, [5, 4, 6, 3]) == (10, [1, 2])
    assert mo.swin_knapsack(49, [49, 49]) == (49, [0])
    sizes, groups = mo.swin_group_windows(49, [21, 28, 49, 7, 14, 28])
    assert sizes == [49, 49, 49] and sorted(sum(groups, [])) == list(range(6))
    assert all(sum([21, 28, 49, 7, 14, 28][i] for i in g) == s for g, s in zip(groups, sizes))


def test_rec_swin_tiny_matches_reference():
    """Swin-T masked reconstruction step (BASELINE config 5) through the oracle vs the reference's own outputs."""
    d = load_golden("rec_swin_tiny")
    cfg = jl(d["cfg"])
    sd = swin_state_dict(d)
    x, y, noise = rec_inputs("swin", cfg)
    assert np.array_equal(noise.numpy(), d["noise"])
    loss, outs, lh, pred, mask, restore, attn = mo.swin_rec_step(sd, x, y, noise, cfg)
    assert np.array_equal(mask.numpy(), d["mask"]) and np.array_equal(restore.numpy(), d["ids_restore"])
    for i in range(4):
        assert np.array_equal(outs[i][1].numpy(), d[f"coords_l{i + 1}"][0])
    assert abs(loss.item() - float(d["loss"])) <= 2e-6 * abs(float(d["loss"]))
    assert list(attn.shape) == list(d["attn_shape"])
    for t, k in ((outs[0][0], "emb_l1"), (outs[1][0], "emb_l2"), (outs[2][0], "emb_l3"), (outs[3][0], "emb_l4"),
                 (lh, "emb_lh"), (pred, "pred"), (attn, "attn")):
        assert_checksums(t, d[k + "_checksums"], 2e-5, k)
    assert torch.allclose(lh, torch.from_numpy(d["emb_lh"]), atol=3e-5, rtol=1e-5)
    loss.backward()
    for n, gn in zip(jl(d["grad_names"]), d["grad_norms"]):
        assert sd[n].grad is not None, n
        assert abs(sd[n].grad.double().norm().item() - gn) <= 2e-4 * gn + 1e-9, n
    for k in d.files:
        if k.startswith("grad::"):
            ref = torch.from_numpy(d[k])
            assert torch.allclose(sd[k[6:]].grad, ref, atol=1e-6 + 2e-4 * ref.abs().max().item(), rtol=1e-4), k


@pytest.mark.slow
def test_rec_base_matches_reference():
    _check_rec("base")


def test_lr_schedule():
    d = load_golden("train_tiny")
    for e, lr, g0, g1 in d["sched"]:
        got = mo.cosine_lr(e, 2e-3, 1e-5, 5, 40)
        assert got == pytest.approx(lr, rel=1e-12, abs=1e-18)
        assert g0 == pytest.approx(lr) and g1 == pytest.approx(0.5 * lr)


def test_train_trajectory_tiny():
    """5 AdamW steps restated with the oracle's own update rule reproduce the reference trainer's loss sequence,
    LR sequence and final parameters."""
    from eventpretrain_amd.testing import det_normalish
    d = load_golden("train_tiny")
    cfg = dict(input=64, patch=16, dim=192, depth=12, heads=3, dec_dim=128, dec_depth=4, dec_heads=4, mask_ratio=0.5, B=2)
    sd = rec_state_dict(cfg)
    train = [k for k in sd if "pos_embed" not in k]
    decay, no_decay = mo.decay_split([(k, tuple(sd[k].shape)) for k in train])
    assert jl(d["group_decay"]) == {"decay": len(decay), "no_decay": len(no_decay)}
    m = {k: torch.zeros_like(sd[k]) for k in train}
    v = {k: torch.zeros_like(sd[k]) for k in train}
    lr0, min_lr = float(d["lr"]), float(d["min_lr"])
    n_steps = len(d["losses"])
    for s in range(n_steps):
        lr = mo.cosine_lr(s / n_steps + 0, lr0, min_lr, int(d["warmup_epochs"]), int(d["epochs"]))
        assert lr == pytest.approx(d["lrs"][s], rel=1e-12, abs=1e-18)
        for k in train:
            sd[k] = sd[k].detach().requires_grad_(True)
        x = det_normalish(f"train.voxels.{s}", (2, 5, 64, 64)) * 0.5
        y = det_normalish(f"train.sub_frame.{s}", (2, 1, 64, 64))
        loss = mo.rec_step(sd, x, y, torch.from_numpy(d["noise"][s]), cfg)[0]
        assert loss.item() == pytest.approx(d["losses"][s], rel=2e-5), s
        loss.backward()
        with torch.no_grad():
            for k in train:
                wd = float(d["weight_decay"]) if k in decay else 0.0
                p, m[k], v[k] = mo.adamw_step(sd[k], sd[k].grad, m[k], v[k], s + 1, lr, wd)
                sd[k] = p
    names = jl(d["param_names"])
    for n, ws in zip(names, d["param_wsums"]):
        got = checksums(sd[n])[2]
        # the key third of qkv.bias has a mathematically zero gradient (softmax shift invariance); Adam turns its
        # rounding noise into +-lr steps, so that parameter is only checked loosely.
        tol = 5e-4 if n.endswith("attn.qkv.bias") else 1e-5
        assert got == pytest.approx(ws, rel=1e-4, abs=tol), n


@pytest.mark.parametrize("use_queue", [True, False])
def test_con_small_matches_reference(use_queue):
    from eventpretrain_amd.testing import det_normalish, det_value_for, det_uniform
    d = load_golden("con_small_queue" if use_queue else "con_small_noqueue")
    keys = jl(d["state_keys"])
    sd = {}
    for k, shp in keys.items():
        leaf = k.split(".")[-1]
        if leaf == "pos_embed":
            sd[k] = torch.from_numpy(mo.sincos_2d(shp[-1], 14)).float().unsqueeze(0)
        elif leaf == "queue":
            sd[k] = torch.nn.functional.normalize(det_uniform(k, shp, -1.0, 1.0), dim=0)
        elif leaf == "queue_ptr":
            sd[k] = torch.zeros(1, dtype=torch.long)
        else:
            sd[k] = det_value_for(k, shp)
        if sd[k].is_floating_point() and leaf not in ("pos_embed", "queue", "running_mean", "running_var"):
            sd[k].requires_grad_(True)
    x = det_normalish("con.voxels", (2, 5, 224, 224)) * 0.5
    clip = det_normalish("con.clip_emb", (2, 197, 512))
    cfg = dict(patch=16, heads=12, T=0.07, use_queue=use_queue)
    loss, h_org, h_proj, c_org, c_proj, attn, side = mo.con_step(sd, x, clip, cfg)
    assert loss.item() == pytest.approx(float(d["loss"]), rel=5e-6)
    assert_checksums(h_org, d["emb_h_org_checksums"], 2e-5)
    assert_checksums(h_proj, d["emb_h_proj_checksums"], 5e-5)
    assert_checksums(c_org, d["clip_org_checksums"], 2e-5)
    assert_checksums(c_proj, d["clip_proj_checksums"], 2e-5)
    assert_checksums(attn, d["attn_checksums"], 2e-5)
    loss.backward()
    for n, gn in zip(jl(d["grad_names"]), d["grad_norms"]):
        assert sd[n].grad is not None, n
        assert sd[n].grad.double().norm().item() == pytest.approx(gn, rel=3e-3, abs=1e-6), n  # BN backward amplifies fp32 rounding
    for k, cs in zip(jl(d["bn_keys"]), d["bn_checksums"]):
        assert_checksums(side[k], cs, 2e-5, k)
    if use_queue:
        assert_checksums(side["queue"], d["queue_after_checksums"], 1e-6)
        assert int(side["queue_ptr"]) == int(d["queue_ptr_after"][0])


def con_swin_state_dict(d):
    from eventpretrain_amd.testing import det_value_for, det_uniform
    sd = {}
    for k, shp in jl(d["state_keys"]).items():
        leaf = k.split(".")[-1]
        if leaf == "relative_position_index":
            continue
        if leaf == "queue":
            sd[k] = torch.nn.functional.normalize(det_uniform(k, shp, -1.0, 1.0), dim=0)
        elif leaf == "queue_ptr":
            sd[k] = torch.zeros(1, dtype=torch.long)
        else:
            sd[k] = det_value_for(k, shp)
        if sd[k].is_floating_point() and leaf not in ("queue", "running_mean", "running_var"):
            sd[k].requires_grad_(True)
    return sd


def test_con_swin_matches_reference():
    """Contrastive step on the Swin-T hub (dense Swin forward, Conv2d CLIP projection, queue InfoNCE) through the oracle."""
    from eventpretrain_amd.testing import det_normalish
    d = load_golden("con_swin_tiny_queue")
    sd = con_swin_state_dict(d)
    x = det_normalish("con.voxels", (2, 5, 224, 224)) * 0.5
    clip = det_normalish("con.clip_emb", (2, 197, 512))
    cfg = dict(input=224, window=7, depths=[2, 2, 6, 2], heads=[3, 6, 12, 24], T=0.07)
    loss, h_org, h_proj, c_org, c_proj, attn, side = mo.swin_con_step(sd, x, clip, cfg)
    assert loss.item() == pytest.approx(float(d["loss"]), rel=5e-6)
    assert list(attn.shape) == list(d["attn_shape"])
    assert_checksums(h_org, d["emb_h_org_checksums"], 2e-5)
    assert_checksums(h_proj, d["emb_h_proj_checksums"], 5e-5)
    assert_checksums(c_org, d["clip_org_checksums"], 2e-5)
    assert_checksums(c_proj, d["clip_proj_checksums"], 2e-5)
    assert_checksums(attn, d["attn_checksums"], 2e-5)
    loss.backward()
    for n, gn in zip(jl(d["grad_names"]), d["grad_norms"]):
        assert sd[n].grad is not None, n
        assert sd[n].grad.double().norm().item() == pytest.approx(gn, rel=3e-3, abs=1e-6), n
    assert_checksums(side["queue"], d["queue_after_checksums"], 1e-6)
    assert int(side["queue_ptr"]) == int(d["queue_ptr_after"][0])


def test_evg_augment_matches_reference():
    """View augmentation of the voxel grid (crop box, nearest resize, h-flip, time flip + negate): the oracle's decision
    stream and pixel transform against the reference's own evg_augment under np.random.seed (bit-exact)."""
    from oracle import augment_oracle as ao
    from eventpretrain_amd.testing import det_normalish
    d = load_golden("evg_augment")
    for tag in jl(d["tags"]):
        shp, size, seed = tuple(int(v) for v in d[f"{tag}_shape"]), tuple(int(v) for v in d[f"{tag}_size"]), int(d[f"{tag}_seed"])
        v = det_normalish(f"aug.view.{tag}", shp).numpy()
        prm = ao.draw_evg_params(np.random.RandomState(seed), shp[1], shp[2], 0.8)
        assert prm[5] == int(d[f"{tag}_tflip"])
        out = ao.evg_transform(v, prm, size, negate=shp[0] in (5, 6))
        if f"{tag}_out" in d.files:
            assert np.array_equal(out, d[f"{tag}_out"]), tag
        else:
            assert np.array_equal(out.reshape(-1)[::7], d[f"{tag}_sample"]), tag
            assert_checksums(torch.from_numpy(out), d[f"{tag}_checksums"], 1e-12, tag)


FT_CFG = {"vit_small": dict(backbone="vit", patch=16, heads=12),
          "swin_tiny": dict(backbone="swin", input=224, window=7, depths=[2, 2, 6, 2], heads=[3, 6, 12, 24])}


def test_events_augment_matches_reference():
    """Event-level augmentation (erase rows, add correlated rows, re-sort), the sensor -> input rescale and the voxel grid
    of the result: the oracle's decision stream + merge against the reference's events_augment / events_reshape /
    events_to_voxel_grid under np.random.seed (bit-exact; fixture clips have distinct stamps)."""
    from oracle import augment_oracle as ao
    d = load_golden("events_augment")
    for tag in jl(d["tags"]):
        seed, n, sh, sw, S = (int(v) for v in d[tag + "_meta"])
        ev = d[tag + "_events_in"]
        dec = ao.draw_erase_add(np.random.RandomState(seed), n)
        assert (dec is None) == (n < 100), tag
        out = ao.erase_add_apply(ev, dec, (sh, sw))
        assert np.array_equal(out, d[tag + "_events_out"]), tag
        assert np.all(np.diff(out[:, 2]) >= 0)
        g = voxel_grid(ao.events_reshape(out, sw, sh, S, S), 5, (S, S))
        assert np.array_equal(g, d[tag + "_voxel"]), tag


def test_loader_chain_in_the_n_imagenet_draw_order():
    """the events half of PretrainNImageNetDataset.__getitem__ in the dataset's OWN order (one seed, then sample after sample
    on the running stream; evg_augment never re-seeds) -- the oracle's composition against the reference's functions run in that order
    (tests/golden/loader_chain_nimagenet.npz, oracle/gen_golden.py gen_chain_nimagenet). One draw too many or too few anywhere in a
    sample would shift every later sample's window."""
    from oracle import augment_oracle as ao
    from eventpretrain_amd.testing import synthetic_events
    d = load_golden("loader_chain_nimagenet")
    for tag in jl(d["tags"]):
        seed, fix = int(d[f"{tag}_seed"]), int(d[f"{tag}_fix"])
        rs = np.random.RandomState(seed)
        for i, n_ev in enumerate(int(v) for v in d[f"{tag}_sizes"]):
            ev = synthetic_events(8000 + seed * 10 + i, n_ev, width=640, height=480)
            win, n_aug, prm, g = ao.n_imagenet_sample(rs, ev, fix, (480, 640), 224, 5)
            k = f"{tag}_{i}"
            assert list(win) == d[f"{k}_window"].tolist(), k
            assert n_aug == int(d[f"{k}_n_aug"]) and prm[5] == int(d[f"{k}_tflip"]), k
            assert np.array_equal(g.reshape(-1)[::7], d[f"{k}_evg_sample"]), k
            assert_checksums(torch.from_numpy(g), d[f"{k}_evg_checksums"], 1e-12, k)


def test_label_smoothing_formula():
    """The restated timm LabelSmoothingCrossEntropy against torch's built-in label_smoothing (same published formula)."""
    g = torch.Generator().manual_seed(5)
    pred = torch.randn(7, 10, generator=g, dtype=torch.float64) * 3
    label = torch.randint(0, 10, (7,), generator=g)
    for s_ in (0.0, 0.1, 0.3):
        assert torch.allclose(mo.label_smoothing_ce(pred, label, s_), torch.nn.functional.cross_entropy(pred, label, label_smoothing=s_), rtol=1e-12)
    assert mo.clip_coef(10.0, 1.0) == pytest.approx(1.0 / (10.0 + 1e-6)) and mo.clip_coef(0.5, 1.0) == 1.0


def ft_state_dict(d, grid=14):
    from eventpretrain_amd.testing import det_value_for
    sd = {}
    for k, shp in jl(d["state_keys"]).items():
        if k.endswith("relative_position_index"):
            continue
        if k.endswith("pos_embed"):
            sd[k] = torch.from_numpy(mo.sincos_2d(shp[-1], grid)).float().unsqueeze(0)
        else:
            sd[k] = det_value_for(k, shp).requires_grad_(True)
    return sd


@pytest.mark.parametrize("tag", ["vit_small", "swin_tiny"])
def test_ft_cls_matches_reference(tag):
    """Classification fine-tuning step (dense backbone, token mean, Linear head, cross-entropy) through the oracle
    against the reference's own FtClsHubModel outputs."""
    from eventpretrain_amd.testing import det_normalish
    d = load_golden("ft_cls_" + tag)
    sd = ft_state_dict(d)
    x = det_normalish("ft.voxels", (2, 5, 224, 224)) * 0.5
    loss, pred, emb_h, attn = mo.ft_cls_step(sd, x, torch.from_numpy(d["label"]), FT_CFG[tag])
    assert loss.item() == pytest.approx(float(d["loss"]), rel=5e-6)
    assert torch.allclose(pred, torch.from_numpy(d["pred"]), atol=2e-5, rtol=1e-5)
    assert_checksums(emb_h, d["emb_h_checksums"], 2e-5)
    assert_checksums(attn, d["attn_checksums"], 2e-5)
    loss.backward()
    for n, gn in zip(jl(d["grad_names"]), d["grad_norms"]):
        assert sd[n].grad is not None, n
        assert sd[n].grad.double().norm().item() == pytest.approx(gn, rel=1e-3, abs=1e-7), n
    for k in ("classify_head.weight", "classify_head.bias"):
        ref = torch.from_numpy(d["grad::" + k])
        assert torch.allclose(sd[k].grad, ref, atol=1e-6 + 1e-4 * ref.abs().max().item(), rtol=1e-4), k


# ----------------------------------------------------------------------------------------------- round 2 fixtures
def test_density_masking_matches_reference():
    """vit.py:80-103 with masking_strategy density / anti-density: the oracle's noise equals the reference's bit for bit
    (voxel-made grids with their many exact ties, an event-free band, random grids), its ids agree with the reference's up to
    the order among tied values (helpers.assert_ids_equal_up_to_ties says why that order is not a property of the algorithm)."""
    from helpers import assert_ids_equal_up_to_ties, density_inputs
    d = load_golden("masking_density")
    seen_ties = seen_exact = 0
    for c in jl(d["cases"]):
        key = c["key"]
        x = density_inputs(d, c["tag"])
        n = mo.density_noise(x, 16, c["strategy"])
        assert np.array_equal(n.numpy(), d[key + "_noise"]), key
        k, m, r = mo.masking_from_noise(n, c["ratio"])
        assert_ids_equal_up_to_ties(d[key + "_noise"], k.shape[1], (d[key + "_ids_keep"], d[key + "_mask"], d[key + "_ids_restore"]),
                                    (k.numpy(), m.numpy(), r.numpy()), key)
        seen_ties += c["ties"] > 0
        seen_exact += c["ties"] == 0
    assert seen_ties and seen_exact


def _con_state_dict(d, grid=14):
    from eventpretrain_amd.testing import det_uniform, det_value_for
    sd = {}
    for k, shp in jl(d["state_keys"]).items():
        leaf = k.split(".")[-1]
        if leaf == "pos_embed":
            sd[k] = torch.from_numpy(mo.sincos_2d(shp[-1], grid)).float().unsqueeze(0)
        elif leaf == "queue":
            sd[k] = torch.nn.functional.normalize(det_uniform(k, shp, -1.0, 1.0), dim=0)
        elif leaf == "queue_ptr":
            sd[k] = torch.zeros(1, dtype=torch.long)
        else:
            sd[k] = det_value_for(k, shp)
        if sd[k].is_floating_point() and leaf not in ("pos_embed", "queue", "running_mean", "running_var"):
            sd[k].requires_grad_(True)
    return sd


@pytest.mark.slow
def test_con_base_and_adj_match_reference():
    """BASELINE.json config 3 at its named width: the reference's ViT-Base hub in the `con` phase (queue length 8) and the
    same step with the backbone frozen except norm_layer (`adj`, main_pretrain.py:281-284)."""
    from eventpretrain_amd.testing import det_normalish
    d = load_golden("con_base_queue")
    sd = _con_state_dict(d)
    x = det_normalish("conb.voxels", (2, 5, 224, 224)) * 0.5
    clip = det_normalish("conb.clip_emb", (2, 197, 512))
    cfg = dict(patch=16, heads=12, T=0.07, use_queue=True)
    loss, h_org, h_proj, c_org, c_proj, attn, side = mo.con_step(sd, x, clip, cfg)
    assert loss.item() == pytest.approx(float(d["loss"]), rel=5e-6)
    assert_checksums(h_org, d["emb_h_org_checksums"], 2e-5)
    assert_checksums(h_proj, d["emb_h_proj_checksums"], 5e-5)
    assert_checksums(attn, d["attn_checksums"], 2e-5)
    assert_checksums(side["queue"], d["queue_after_checksums"], 1e-6)
    assert int(side["queue_ptr"]) == int(d["queue_ptr_after"][0])
    # adj: same forward; only the unfrozen parameters receive gradients
    a = load_golden("adj_base_queue")
    assert float(a["loss"]) == pytest.approx(float(d["loss"]), rel=1e-6)
    frozen = set(jl(a["frozen"]))
    for k in frozen:
        sd[k].requires_grad_(False)
    loss2 = mo.con_step(sd, x, clip, cfg)[0]
    loss2.backward()
    names = jl(a["grad_names"])
    assert not (set(names) & frozen) and all(k.startswith("backbone.") and "norm_layer" not in k for k in frozen)
    for n, gn in zip(names, a["grad_norms"]):
        assert sd[n].grad is not None, n
        assert sd[n].grad.double().norm().item() == pytest.approx(gn, rel=3e-3, abs=1e-6), n
    assert all(sd[k].grad is None for k in frozen)


@pytest.mark.slow
def test_rec_swin_base_matches_reference():
    """Swin-Base (2-2-18-2, 128..1024) instantiated from the reference's classes: forward through the oracle."""
    d = load_golden("rec_swin_base")
    cfg = jl(d["cfg"])
    sd = swin_state_dict(d)
    x, y, noise = rec_inputs("swinb", cfg)
    assert np.array_equal(noise.numpy(), d["noise"])
    with torch.no_grad():
        loss, outs, lh, pred, mask, restore, attn = mo.swin_rec_step(sd, x, y, noise, cfg)
    assert np.array_equal(mask.numpy(), d["mask"]) and np.array_equal(restore.numpy(), d["ids_restore"])
    assert abs(loss.item() - float(d["loss"])) <= 2e-6 * abs(float(d["loss"]))
    assert list(attn.shape) == list(d["attn_shape"])
    for t, k in ((outs[0][0], "emb_l1"), (outs[1][0], "emb_l2"), (outs[2][0], "emb_l3"), (outs[3][0], "emb_l4"), (lh, "emb_lh"),
                 (pred, "pred"), (attn, "attn")):
        assert_checksums(t, d[k + "_checksums"], 2e-5, k)


def test_autocast_fixture_is_bf16_class():
    """The reference's own bf16-autocast losses sit within 1e-2 of its fp32 losses: the scale the HIP bf16 mode is reported on."""
    a = load_golden("rec_autocast_bf16")
    for tag in ("tiny", "small", "base"):
        f = float(load_golden(f"rec_{tag}")["loss"])
        assert abs(float(a[f"{tag}_loss"]) - f) / f <= 1e-2, tag


def test_frame_augment_matches_reference():
    """Target-side view augmentation (view_augment.py:79-89): the oracle's decisions (same legacy stream as evg_augment) and
    its crop / bicubic / flip / negate transform against the reference's own frame_augment under np.random.seed (bit-exact:
    same ATen bicubic op on the CPU)."""
    from oracle import augment_oracle as ao
    from eventpretrain_amd.testing import det_normalish
    d = load_golden("frame_augment")
    for tag in jl(d["tags"]):
        shp, S, seed = tuple(int(v) for v in d[f"{tag}_shape"]), int(d[f"{tag}_size"]), int(d[f"{tag}_seed"])
        f = det_normalish(f"aug.frame.{tag}", shp).numpy()
        prm = ao.draw_evg_params(np.random.RandomState(seed), shp[1], shp[2], 0.8)
        assert prm[5] == int(d[f"{tag}_tflip"])
        out = ao.frame_transform(f, prm, (S, S))
        if f"{tag}_out" in d.files:
            assert np.array_equal(out, d[f"{tag}_out"]), tag
        else:
            assert np.array_equal(out.reshape(-1)[::7], d[f"{tag}_sample"]), tag
