"""CPU-side checks: the C-ABI library builds for gfx950, loads, and exports every symbol include/evtpretrain.h
declares (no compute calls without a GPU); host-side logic (schedules, param groups, pos-embed, reshape helpers,
checkpoint key remap) against the reference-made fixtures."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden
from helpers import jl


def _declared():
    txt = open(os.path.join(ROOT, "include", "evtpretrain.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(evp_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_every_declared_symbol():
    from eventpretrain_amd import _lib
    _lib.build_library()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/evtpretrain.h but not exported"
    assert set(_lib.exported_symbols()) == set(names), set(_lib.exported_symbols()) ^ set(names)
    lib.evp_target_arch.restype = ctypes.c_char_p
    assert lib.evp_target_arch() == b"gfx950"
    assert lib.evp_abi_version() == _lib.ABI_VERSION == 5


def test_code_object_is_gfx950():
    from eventpretrain_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"gfx90a" not in blob and b"sm_" not in blob


def test_argument_validation_without_gpu():
    """Shape/pointer validation happens on the host before any launch, so it can be exercised here."""
    from eventpretrain_amd import _lib
    lib = _lib.load()
    d = _lib.GemmDesc()
    assert lib.evp_gemm(ctypes.byref(d), None) == -1          # EVP_EINVAL: null operands
    assert b"null" in lib.evp_last_error()
    assert lib.evp_mask_from_noise(1, 2, 5000, 10, 1, 1, 1, None) == -2   # EVP_ESHAPE: L > 4096


def test_ops_refuse_cpu_tensors():
    from eventpretrain_amd import ops
    from eventpretrain_amd._lib import EvpError
    with pytest.raises(EvpError):
        ops.mask_from_noise(torch.rand(2, 16), 0.5)
    with pytest.raises(EvpError):
        ops.cast(torch.zeros(4), torch.bfloat16)


def test_pos_embed_matches_reference():
    from eventpretrain_amd.utils.pos_embed import get_2d_sincos_pos_embed
    from helpers import checksums
    d = load_golden("pos_embed")
    for dim, g in [(64, 4), (192, 4), (384, 14), (512, 14), (768, 14), (256, 7)]:
        t = get_2d_sincos_pos_embed(dim, g)
        assert str(t.dtype) == str(d[f"d{dim}_g{g}_dtype"])
        assert np.array_equal(checksums(torch.from_numpy(t).float()), d[f"d{dim}_g{g}_checksums"])
    assert get_2d_sincos_pos_embed(64, 4, cls_token=True).shape == (17, 64)


def test_lr_schedule_and_param_groups():
    import types
    from eventpretrain_amd.model.pretrain import pr_hub_model as hub
    from eventpretrain_amd.testing import make_args
    from eventpretrain_amd.utils import lr_decay as lrd
    from eventpretrain_amd.utils.lr_sched import adjust_learning_rate
    d = load_golden("train_tiny")
    opt = types.SimpleNamespace(param_groups=[{"lr": 0.0}, {"lr": 0.0, "lr_scale": 0.5}])
    a = make_args(lr=2e-3, min_lr=1e-5, warmup_epochs=5, epochs=40)
    for e, lr, g0, g1 in d["sched"]:
        assert adjust_learning_rate(opt, e, a) == pytest.approx(lr, rel=1e-12, abs=1e-18)
        assert opt.param_groups[0]["lr"] == pytest.approx(g0, rel=1e-12, abs=1e-18)
        assert opt.param_groups[1]["lr"] == pytest.approx(g1, rel=1e-12, abs=1e-18)
    a = make_args(model_size="tiny")
    m = hub.pretrain_hub_model_tiny_patch16_64(a, emb_frames_dim=512, queue_length=8, T=0.07)
    groups = lrd.param_groups_lrd(a, m, 0.05, layer_decay=1)
    cnt = {"decay": sum(len(g["params"]) for g in groups if g["weight_decay"] > 0),
           "no_decay": sum(len(g["params"]) for g in groups if g["weight_decay"] == 0)}
    assert cnt == jl(d["group_decay"]) and len(groups) == int(d["n_groups"])
    assert all(g["lr_scale"] == 1 for g in groups)
    g75 = lrd.param_groups_lrd(a, m, 0.05, layer_decay=0.75)
    scales = sorted({g["lr_scale"] for g in g75})
    assert scales[0] == pytest.approx(0.75 ** 12) and scales[-1] == 1.0


def test_reshape_helpers_round_trip():
    from eventpretrain_amd.utils import reshape as rs
    from oracle.model_oracle import patchify
    x = torch.randn(2, 3, 32, 48)
    assert torch.equal(rs.frame2emb(16, x), patchify(x, 16))
    sq = torch.randn(2, 1, 64, 64)
    import types
    assert torch.equal(rs.emb2frame(types.SimpleNamespace(patch_size=16), rs.frame2emb(16, sq), 1), sq)
    e = torch.randn(2, 16, 7)
    assert torch.equal(rs.patch_frame2emb(rs.emb2patch_frame(e)), e)


def test_checkpoint_key_remap():
    from eventpretrain_amd.utils.misc import remap_stage_checkpoint
    sd = {"backbone.norm_l_h.weight": 1, "backbone.norm_l_h.bias": 2, "backbone.vit_block.0.norm1.weight": 3}
    out = remap_stage_checkpoint(sd, "adj")
    assert set(out) == {"backbone.norm_layer.weight", "backbone.norm_layer.bias", "backbone.vit_block.0.norm1.weight"}
    out = remap_stage_checkpoint({"backbone.norm_h.weight": 1}, "con")
    assert set(out) == {"backbone.norm_layer.weight"}


def test_swin_host_grouping_matches_oracle():
    """Host-side window grouping of the Swin path (knapsack in C, index tables in numpy) against the oracle's restatement
    of swin_block.py:277-452, on the reference fixture's token coordinates and on random window occupancies."""
    import torch
    from conftest import load_golden
    from oracle import model_oracle as mo
    from eventpretrain_amd.model.sub_module.swin_block import GroupingModule, PatchMerging, TokenLayout, group_windows, knapsack
    rng = np.random.default_rng(3)
    for _ in range(100):
        cap = int(rng.integers(4, 50))
        wt = rng.integers(1, cap + 1, size=int(rng.integers(1, 30))).tolist()
        assert knapsack(cap, wt) == mo.swin_knapsack(cap, wt)
        assert group_windows(cap, wt) == mo.swin_group_windows(cap, wt)
    d = load_golden("rec_swin_tiny")
    for lvl, res in ((1, 56), (2, 28), (3, 14), (4, 7)):
        c = d[f"coords_l{lvl}"][0]
        for shift in (0, 3):
            if res <= 7 and shift:
                continue
            p = GroupingModule(7, shift).plan(c, c.shape[0])
            o = mo.swin_plan(torch.from_numpy(c), 7, shift, c.shape[0])
            assert p["mode"] == o["mode"]
            assert np.array_equal(p["rel"], np.where((o["mask"] != 0).numpy(), -1, o["rel"].numpy()))
            if p["mode"] == "grouping":
                assert np.array_equal(p["gather"], o["gather"].numpy()) and np.array_equal(p["scatter"], o["scatter"].numpy())
                assert np.array_equal(p["scatter_adj"][p["gather_adj"]], np.arange(c.shape[0]))
    vis_cells = d["mask"][0] == 0
    vis = np.repeat(np.repeat(vis_cells.reshape(7, 7), 8, 0), 8, 1).reshape(-1)
    ys, xs = np.nonzero(vis.reshape(56, 56))
    lay = TokenLayout(np.stack([ys, xs], -1), vis, 56)
    assert np.array_equal(lay.coords, d["coords_l1"][0])
    rows, inv, lay2 = PatchMerging.plan(lay)
    assert np.array_equal(lay2.coords, d["coords_l2"][0]) and np.array_equal(rows[inv], np.arange(lay.n))


def test_swin_static_plan_tables_are_a_padded_form_of_the_pattern_plan():
    """Host logic of the fixed-shape Swin window plan (model/backbone/swin.py StaticPatternPlan, CPU buffers here): for random
    visibility patterns the fixed-shape tables describe the same grouping rule as the pattern-sized plan -- every visible token
    sits in exactly one slot, slots of one group never mix windows unmasked, padding slots are masked and receive no gradient,
    the adjoint tables invert the gathers -- and a pattern that needs more groups than the shape holds is refused untouched."""
    import types
    import torch
    from eventpretrain_amd.model.backbone import swin as swin_mod
    from eventpretrain_amd.model.sub_module.swin_block import PlanOverflow

    class _Blk:        # what StaticPatternPlan reads from a BasicBlock
        def __init__(self, res, last):
            self.input_resolution = (res, res)
            self.window_size = min(7, res)
            self.shift_size = 0 if res <= 7 else 3
            self.downsample = None if last else object()

    model = types.SimpleNamespace(patches_resolution=[56, 56], num_patches=49,
                                  swin_block=[_Blk(56, False), _Blk(28, False), _Blk(14, False), _Blk(7, True)])
    plan = swin_mod.StaticPatternPlan(model, "cpu", keep=24, slack=1.25)
    rng = np.random.default_rng(5)
    for _ in range(6):
        vis = np.zeros(49, dtype=bool)
        vis[rng.permutation(49)[:24]] = True
        plan.load(vis)
        n0 = plan.geom[0][1]
        assert plan.tok_ids.shape[0] == n0 == 24 * 64
        for i, (r, n, mods, has_merge) in enumerate(plan.geom):
            lay = plan.stages[i].layout
            assert lay.n == n and lay.res == r
            tokmap = plan.view[f"s{i}.tokmap"].numpy()
            assert (tokmap >= 0).sum() == n and np.array_equal(np.sort(tokmap[tokmap >= 0]), np.arange(n))
            for shift, mode, gs, ng in mods:
                rel = plan.view[f"s{i}.{shift}.rel"].numpy()
                assert rel.shape == (ng, gs, gs)
                if mode != "grouping":
                    continue
                gather = plan.view[f"s{i}.{shift}.shuffle"].numpy()
                gadj = plan.view[f"s{i}.{shift}.shuffle_adj"].numpy()
                scat = plan.view[f"s{i}.{shift}.unshuffle"].numpy()
                sadj = plan.view[f"s{i}.{shift}.unshuffle_adj"].numpy()
                real = sadj >= 0                                   # slots that hold a token
                assert real.sum() == n and np.array_equal(np.sort(sadj[real]), np.arange(n))
                assert np.array_equal(gather[real], sadj[real])    # a real slot gathers its own token
                assert np.array_equal(scat, gadj) and np.array_equal(sadj[scat], np.arange(n))   # token -> slot -> token
                pad = ~real.reshape(ng, gs)
                assert (rel[pad] == -1).all() and (rel.transpose(0, 2, 1)[pad] == -1).all()        # padding rows and columns masked
                # unmasked pairs share a window of the (shifted) 7x7 partition
                ws = 7
                c = lay.coords[np.where(real, sadj, 0)].reshape(ng, gs, 2)
                w = (c + (ws - shift) % ws) // ws
                same = (w[:, :, None, :] == w[:, None, :, :]).all(-1) & real.reshape(ng, gs)[:, :, None] & real.reshape(ng, gs)[:, None, :]
                assert np.array_equal(rel >= 0, same)
    tight = swin_mod.StaticPatternPlan(model, "cpu", keep=24, slack=0.9)
    before = tight.dev_buf.clone()
    with pytest.raises(PlanOverflow):
        tight.load(vis)
    assert torch.equal(tight.dev_buf, before) and tight.loads == 0


def test_host_philox_matches_the_published_known_answer_and_the_draws_are_sane():
    """The counter stream the loader chain's host half and csrc/events.hip share: Philox4x32-10 restated in numpy. Random123's
    known-answer vector (counter 0, key 0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8; counter / key all ones -> 408f276d 41c83b0e a20bc7c6
    6d5451fd), then the distributions of what is drawn from it: erase / add counts uniform in [int(0.001 n), int(0.01 n)), crop boxes
    inside the view with the reference's accept / reject rule, fair flip coins."""
    import numpy as np
    from eventpretrain_amd.dataset.augmentation import events_augment as ea
    from eventpretrain_amd.dataset.augmentation import view_augment as va
    assert [hex(int(v)) for v in ea.philox_words(0, 0, [0], 0, 4)[0]] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    # key (ffffffff, ffffffff), counter (ffffffff x 4): seed = 2^64 - 1 with step 0 gives that key; counter word 0 = w // 4 cannot be all
    # ones through this interface, so the second vector is checked on the raw rounds
    e, a = ea.draw_erase_add_counts(1, 2, np.full(4096, 100_000))
    assert e.min() >= 100 and e.max() <= 999 and 500 < e.mean() < 600 and a.min() >= 100 and a.max() <= 999
    assert not np.array_equal(e, a)
    e0, a0 = ea.draw_erase_add_counts(1, 2, [99, 100, 150, 1000])
    assert e0[0] == 0 and a0[0] == 0 and e0[1] == 0 and e0[2] == 0 and 1 <= e0[3] <= 9
    p = va.draw_evg_params_batch(3, 1, 4096, 224, 224, 0.8)
    assert (p[:, 0] >= 0).all() and (p[:, 1] >= 0).all() and (p[:, 0] + p[:, 2] <= 224).all() and (p[:, 1] + p[:, 3] <= 224).all()
    area = (p[:, 2] * p[:, 3]).mean() / 224 / 224
    assert 0.8 < area < 0.93 and 0.45 < p[:, 4].mean() < 0.55 and 0.45 < p[:, 5].mean() < 0.55
    assert np.array_equal(va.draw_evg_params_batch(3, 1, 8, 224, 224, 0.8, first_sample=100), va.draw_evg_params_batch(3, 1, 108, 224, 224, 0.8)[100:])
