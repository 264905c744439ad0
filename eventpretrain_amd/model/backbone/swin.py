"""Swin-T backbone on sparse tokens (reference model/backbone/swin.py:13-302): same constructor, factory, return tuples
and state-dict keys.

Per step the only data-dependent host decision is the 49-cell visibility pattern of sample 0, which the reference
applies to the whole batch (swin.py:151). It is read back once (49 floats), turned into per-stage `StagePlan`s (window
groups, gather tables, relative-position indices; cached per pattern) and everything else stays on the GPU: the 4x4
patch projection is evaluated only at the visible tokens, the stage blocks run on grouped tokens, and the three
stage-fusion convs are evaluated only at each sample's kept decoder cells (ops.SwinFuseConvFn) instead of
re-densifying into zero grids."""
from collections import OrderedDict
from functools import partial

import numpy as np
import torch
import torch.nn as nn

from ... import ops
from ..sub_module.swin_block import (BasicBlock, GroupingModule, PatchEmbed, PatchMerging, PlanOverflow, StagePlan, TokenLayout,
                                     _dev_i32)
from .vit import init_linear_and_norm

_PLAN_CACHE_SIZE = 32


class _PatternPlan:
    """All host-derived tables for one visibility pattern."""

    def __init__(self, model, vis_cells, device):
        res = model.patches_resolution[0]
        g = int(round(vis_cells.shape[0] ** 0.5))
        rep = res // g
        vis = np.repeat(np.repeat(vis_cells.reshape(g, g), rep, 0), rep, 1).reshape(-1)
        ys, xs = np.nonzero(vis.reshape(res, res))
        layout = TokenLayout(np.stack([ys, xs], -1), vis, res)
        self.tok_ids = torch.from_numpy(np.nonzero(vis)[0].astype(np.int64)).to(device)      # visible stage-1 tokens
        self.stages, self.fuse = [], []
        for blk in model.swin_block:
            sp = StagePlan(blk, layout, device)
            self.stages.append(sp)
            r = layout.res
            tokmap = np.full(r * r, -1, dtype=np.int32)
            tokmap[layout.coords[:, 0] * r + layout.coords[:, 1]] = np.arange(layout.n, dtype=np.int32)
            self.fuse.append((_dev_i32(tokmap, device), _dev_i32(layout.coords, device), r, r // g))
            if sp.merge is not None:
                layout = sp.merge[2]


def _cell_layout(vis_cells, res):
    g = int(round(vis_cells.shape[0] ** 0.5))
    rep = res // g
    vis = np.repeat(np.repeat(vis_cells.reshape(g, g), rep, 0), rep, 1).reshape(-1)
    ys, xs = np.nonzero(vis.reshape(res, res))
    return TokenLayout(np.stack([ys, xs], -1), vis, res), vis


class _StaticStage:
    """StagePlan's interface (layout, plain, shifted, merge) over tables with fixed device addresses."""

    def __init__(self):
        self.layout = self.plain = self.shifted = self.merge = None


class StaticPatternPlan:
    """The same tables as _PatternPlan, but of ONE shape for every visibility pattern with `keep` visible cells and at fixed
    device addresses, so that a captured HIP graph can serve any pattern: all tables live in one device byte buffer whose
    typed views the forward uses; `load(vis_cells)` recomputes them on the host (window grouping with a fixed group size and
    group count, GroupingModule.plan(fixed=...)) into a pinned staging buffer and enqueues ONE H2D copy on the current stream.
    `slack`: group-count head-room over tokens / group_size; a pattern that needs more raises PlanOverflow (the caller then
    runs that step eagerly with a pattern-sized plan)."""

    RING = 3

    def __init__(self, model, device, keep, slack=1.25):
        self.device = torch.device(device)
        self.keep = int(keep)
        res0 = model.patches_resolution[0]
        self.g = int(round(model.num_patches ** 0.5))
        probe = np.zeros(model.num_patches, dtype=bool)
        probe[:self.keep] = True
        layout, _ = _cell_layout(probe, res0)
        spec = [("tok_ids", np.int64, (layout.n,))]
        self.geom = []                                   # per stage: (res, n, [(shift, mode, gs, ng)], has_merge)
        for i, blk in enumerate(model.swin_block):
            n, r, ws = layout.n, layout.res, blk.window_size
            mods = []
            for shift in ((0, blk.shift_size) if ws < min(blk.input_resolution) else (0,)):
                if n <= 2 * ws * ws:
                    mode, gs, ng = "masking", n, 1
                else:
                    mode, gs = "grouping", ws * ws
                    ng = int(np.ceil(n / gs * slack)) + 1
                    spec += [(f"s{i}.{shift}.{k}", np.int32, sh) for k, sh in
                             (("shuffle", (ng * gs,)), ("shuffle_adj", (n,)), ("unshuffle", (n,)), ("unshuffle_adj", (ng * gs,)))]
                spec.append((f"s{i}.{shift}.rel", np.int32, (ng, gs, gs)))
                mods.append((shift, mode, gs, ng))
            spec += [(f"s{i}.tokmap", np.int32, (r * r,)), (f"s{i}.coords32", np.int32, (n, 2)), (f"s{i}.coords64", np.int64, (1, n, 2))]
            has_merge = blk.downsample is not None
            if has_merge:
                spec += [(f"s{i}.merge_rows", np.int32, (n,)), (f"s{i}.merge_inv", np.int32, (n,))]
            self.geom.append((r, n, mods, has_merge))
            if has_merge:
                layout = PatchMerging.plan(layout)[2]
        self.offsets, off = {}, 0
        for name, dtp, shape in spec:
            nb = int(np.prod(shape)) * np.dtype(dtp).itemsize
            self.offsets[name] = (off, nb, dtp, shape)
            off += (nb + 15) // 16 * 16
        self.nbytes = off
        self.dev_buf = torch.zeros(self.nbytes, dtype=torch.uint8, device=self.device)
        self._cuda = self.device.type == "cuda"     # a CPU "device" is only used by the host-logic tests
        self.pins = [torch.zeros(self.nbytes, dtype=torch.uint8) for _ in range(self.RING)]
        if self._cuda:
            self.pins = [p_.pin_memory() for p_ in self.pins]
        self.pin_events = [None] * self.RING
        self.turn = 0
        tdt = {np.int32: torch.int32, np.int64: torch.int64}
        self.view = {name: self.dev_buf[o:o + nb].view(tdt[dtp]).view(*shape) for name, (o, nb, dtp, shape) in self.offsets.items()}
        # the objects the forward walks
        self.tok_ids = self.view["tok_ids"]
        self.stages, self.fuse = [], []
        for i, (blk, (r, n, mods, has_merge)) in enumerate(zip(model.swin_block, self.geom)):
            st = _StaticStage()
            gms = []
            for shift, mode, gs, ng in mods:
                gm = GroupingModule(blk.window_size, shift)
                tabs = tuple(self.view[f"s{i}.{shift}.{k}"] for k in ("shuffle", "shuffle_adj", "unshuffle", "unshuffle_adj")) if mode == "grouping" else None
                gms.append(gm.bind(mode, gs, ng, self.view[f"s{i}.{shift}.rel"], tabs))
            st.plain, st.shifted = gms[0], gms[-1]
            self.stages.append(st)
            self.fuse.append((self.view[f"s{i}.tokmap"], self.view[f"s{i}.coords32"], r, r // self.g))
        self.loads = self.overflows = 0

    def _host(self, pin, name):
        o, nb, dtp, shape = self.offsets[name]
        return pin.numpy()[o:o + nb].view(dtp).reshape(shape)

    def load(self, vis_cells):
        """Host tables for this pattern -> pinned staging -> device (one async copy). Raises PlanOverflow before touching
        anything the device could still be reading."""
        vis_cells = np.ascontiguousarray(vis_cells, dtype=bool)
        if int(vis_cells.sum()) != self.keep:
            raise PlanOverflow(f"pattern has {int(vis_cells.sum())} visible cells, the static plan was built for {self.keep}")
        res0 = self.geom[0][0]
        layout, vis = _cell_layout(vis_cells, res0)
        staged, layouts = {"tok_ids": np.nonzero(vis)[0].astype(np.int64)}, []
        for i, (r, n, mods, has_merge) in enumerate(self.geom):
            assert layout.n == n and layout.res == r
            for (shift, mode, gs, ng), gm in zip(mods, (self.stages[i].plain, self.stages[i].shifted) if len(mods) == 2 else (self.stages[i].plain,)):
                pl = gm.plan(layout.coords, n, fixed=(gs, ng) if mode == "grouping" else None)
                assert pl["mode"] == mode
                staged[f"s{i}.{shift}.rel"] = pl["rel"]
                if mode == "grouping":
                    for k, src in (("shuffle", "gather"), ("shuffle_adj", "gather_adj"), ("unshuffle", "scatter"), ("unshuffle_adj", "scatter_adj")):
                        staged[f"s{i}.{shift}.{k}"] = pl[src]
            tokmap = np.full(r * r, -1, dtype=np.int32)
            tokmap[layout.coords[:, 0] * r + layout.coords[:, 1]] = np.arange(n, dtype=np.int32)
            staged[f"s{i}.tokmap"], staged[f"s{i}.coords32"], staged[f"s{i}.coords64"] = tokmap, layout.coords, layout.coords[None]
            layouts.append(layout)
            if has_merge:
                rows, inv, layout = PatchMerging.plan(layout)
                staged[f"s{i}.merge_rows"], staged[f"s{i}.merge_inv"] = rows, inv
        # nothing overflowed: publish
        t = self.turn
        self.turn = (t + 1) % self.RING
        if self.pin_events[t] is not None:
            self.pin_events[t].synchronize()           # the copy that last read this pinned buffer has run
        pin = self.pins[t]
        for name, arr in staged.items():
            self._host(pin, name)[...] = arr
        self.dev_buf.copy_(pin, non_blocking=True)
        if self._cuda:
            ev = self.pin_events[t] or torch.cuda.Event()
            ev.record()
            self.pin_events[t] = ev
        for i, lay in enumerate(layouts):
            lay.coords_dev = self.view[f"s{i}.coords64"]
            self.stages[i].layout = lay
            if self.geom[i][3]:
                nxt = layouts[i + 1]
                self.stages[i].merge = (self.view[f"s{i}.merge_rows"], self.view[f"s{i}.merge_inv"], nxt)
        self.loads += 1


class SwinTransformer(nn.Module):
    def __init__(self, args, img_size=224, patch_size=4, decoder_num_patches=49, num_bins=3, mask_ratio=0.50,
                 embed_dim=[96, 192, 384, 768], depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7,
                 out_indices=(0, 1, 2, 3), mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0.2, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs):
        super().__init__()
        self.drop_rate = float(drop_rate)          # pos_drop (swin.py:63,185,258)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]    # stochastic depth decay rule (swin.py:66)
        self.args = args
        self.img_size = img_size
        self.patch_size = patch_size
        self.num_patches = decoder_num_patches
        self.num_layers = len(depths)
        self.embed_dim = embed_dim[0]
        self.out_indices = out_indices
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=num_bins, embed_dim=embed_dim[0],
                                      norm_layer=norm_layer)
        self.patches_resolution = self.patch_embed.patches_resolution
        res = self.patches_resolution
        self.swin_block = nn.ModuleList([
            BasicBlock(dim=int(embed_dim[0] * 2 ** i), input_resolution=(res[0] // (2 ** i), res[1] // (2 ** i)),
                       depth=depths[i], num_heads=num_heads[i], window_size=window_size, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias,
                       qk_scale=qk_scale, drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer=norm_layer,
                       downsample=PatchMerging if (i < self.num_layers - 1) else None)
            for i in range(self.num_layers)])
        self.norm_layer = norm_layer(embed_dim[-1])
        if args.phase == "pretrain" and args.pr_phase in ("rec", "rec+con", "rec-n"):
            self.mask_ratio = mask_ratio
            self.stage1_output_decode = nn.Conv2d(embed_dim[0], embed_dim[-1], kernel_size=8, stride=8)
            self.stage2_output_decode = nn.Conv2d(embed_dim[1], embed_dim[-1], kernel_size=4, stride=4)
            self.stage3_output_decode = nn.Conv2d(embed_dim[2], embed_dim[-1], kernel_size=2, stride=2)
        if args.phase in ("finetune_semseg", "finetune_flow"):
            raise NotImplementedError("dense-prediction fine-tuning heads are out of scope (SURVEY.md section 2, rows 18-21)")
        self._plans = OrderedDict()
        self._static = None          # StaticPatternPlan when a step executor replays this model as a HIP graph

    # ------------------------------------------------------------------------------------------------ static plan
    def enable_static_plan(self, device, slack=1.25):
        """Fixed-shape, fixed-address window tables for the masked forward (see StaticPatternPlan). Returns the per-step
        hook `prepare(noise_cpu) -> bool`: call it with the step's mask noise (CPU tensor [B, cells]; row 0 decides the
        pattern, swin.py:151) BEFORE the forward / the graph replay; False = this pattern does not fit the fixed shape
        (the static plan is switched off for that one step: run it eagerly)."""
        if self.args.masking_strategy != "random":
            # density / anti-density noise is computed on the device from the voxel grid: the pattern is not known on the host
            # before the launch, which is what the fixed-address plan needs
            raise NotImplementedError("enable_static_plan: only the random masking strategy draws its noise on the host")
        keep = int(self.num_patches * (1 - self.mask_ratio))
        self._static_plan = StaticPatternPlan(self, device, keep, slack)
        self._static = self._static_plan

        def prepare(noise_cpu):
            n0 = noise_cpu[0].detach().float().numpy()
            vis = np.zeros(self.num_patches, dtype=bool)
            vis[np.argsort(n0, kind="stable")[:keep]] = True
            try:
                self._static_plan.load(vis)
            except PlanOverflow:
                self._static_plan.overflows += 1
                self._static = None
                return False
            self._static = self._static_plan
            return True
        return prepare

    _init_weights = staticmethod(init_linear_and_norm)

    # ------------------------------------------------------------------------------------------------ masking
    def masking_noise(self, x):
        strategy = self.args.masking_strategy
        if strategy == "random":
            return torch.rand(x.shape[0], self.num_patches, device=x.device)
        if strategy in ("density", "anti-density"):
            p = self.img_size // int(round(self.num_patches ** .5))          # AvgPool2d(32, 32) at 224 (swin.py:125)
            return ops.density_noise(x.detach(), p, 1.0 if strategy == "density" else -1.0)
        raise ValueError(strategy)

    def random_masking(self, x, noise=None):
        """-> (ids_keep [B,K], mask [B,L] (1 = removed), ids_restore [B,L]) on the coarse decoder grid (swin.py:109-145)."""
        if noise is None:
            noise = self.masking_noise(x)
        return ops.mask_from_noise(noise.contiguous().float(), self.mask_ratio)

    def _pattern_plan(self, vis_cells, device):
        key = (vis_cells.tobytes(), str(device))
        plan = self._plans.get(key)
        if plan is None:
            plan = _PatternPlan(self, vis_cells, device)
            self._plans[key] = plan
            if len(self._plans) > _PLAN_CACHE_SIZE:
                self._plans.popitem(last=False)
        else:
            self._plans.move_to_end(key)
        return plan

    # ------------------------------------------------------------------------------------------------ forward
    def _run_stages(self, x, plan, want_attn):
        B = x.shape[0]
        ids = plan.tok_ids.unsqueeze(0).expand(B, -1).contiguous()
        t = self.patch_embed(x, ids)
        if self.training and self.drop_rate > 0:          # pos_drop (swin.py:185,258)
            t = ops.DropoutFn.apply(t, self.drop_rate, ops.draw_drop_seed(t.device))
        outs, attn = [], None
        last = len(self.swin_block) - 1
        for i, blk in enumerate(self.swin_block):
            sp = plan.stages[i]
            if i < last:
                e, lay, t, _, _ = blk(t, sp.layout, sp)
            else:
                e, lay, attn = blk(t, sp.layout, sp, return_attn=want_attn)
            outs.append((e, lay))
        return outs, attn

    def _masked_tail(self, x, plan, ids_keep, mask_t, ids_restore, dev, eps):
        outs, attn = self._run_stages(x, plan, True)
        emb_stage4 = outs[-1][0]
        if self.args.use_feature_fusion:
            fused = []
            for i, conv in enumerate((self.stage1_output_decode, self.stage2_output_decode, self.stage3_output_decode)):
                tokmap, coords, r, k = plan.fuse[i]
                fused.append(ops.SwinFuseConvFn.apply(outs[i][0], conv.weight, conv.bias, tokmap, coords, ids_keep,
                                                      ids_restore, r, k))
            s12 = ops.AddFn.apply(fused[0], fused[1])
            emb_lh = ops.LayerNormFn.apply(s12, fused[2], emb_stage4, self.norm_layer.weight, self.norm_layer.bias, eps)
        else:
            emb_lh = ops.LayerNormFn.apply(emb_stage4, None, None, self.norm_layer.weight, self.norm_layer.bias, eps)
        coords = [lay.coords_tensor(dev) for _, lay in outs]
        return (outs[0][0], outs[1][0], outs[2][0], outs[3][0], emb_lh, coords[0], coords[1], coords[2], coords[3],
                mask_t, ids_restore, attn)

    def forward(self, x, mask=False, noise=None):
        eps = self.norm_layer.eps
        dev = x.device
        if mask:
            # The window plan is host work and depends on the visibility pattern of sample 0 (swin.py:151). With the random
            # strategy the noise is therefore drawn on the HOST (or taken from a CPU tensor the caller passes): the pattern
            # is known before anything is launched and the step has no device->host read-back; the device gets the same
            # noise by an asynchronous copy and computes the ids with the usual kernel. Device-resident noise (explicit
            # CUDA tensor, density strategies) keeps the one read-back of the 49-float mask row.
            vis_cells = None
            static = self._static if (noise is not None and noise.is_cuda) else None
            if static is not None:
                # the executor loaded this step's tables (prepare(noise_cpu)) and hands over the same noise in device memory
                ids_keep, mask_t, ids_restore = self.random_masking(x, noise)
                return self._masked_tail(x, static, ids_keep, mask_t, ids_restore, dev, eps)
            if noise is None and self.args.masking_strategy == "random":
                noise = torch.rand(x.shape[0], self.num_patches)
            if noise is not None and not noise.is_cuda:
                n0 = noise[0].detach().float().numpy()
                keep = int(self.num_patches * (1 - self.mask_ratio))
                vis_cells = np.zeros(self.num_patches, dtype=bool)
                vis_cells[np.argsort(n0, kind="stable")[:keep]] = True      # same order as the kernel: value, then index
                noise = noise.to(dev, non_blocking=True)
            ids_keep, mask_t, ids_restore = self.random_masking(x, noise)
            if vis_cells is None:
                vis_cells = mask_t[0].detach().cpu().numpy() == 0      # the one host read-back of the step
            plan = self._pattern_plan(vis_cells, dev)
            return self._masked_tail(x, plan, ids_keep, mask_t, ids_restore, dev, eps)

        plan = self._pattern_plan(np.ones(self.num_patches, dtype=bool), dev)
        outs, attn = self._run_stages(x, plan, True)
        emb_h = ops.LayerNormFn.apply(outs[-1][0], None, None, self.norm_layer.weight, self.norm_layer.bias, eps)
        return outs[0][0], outs[1][0], outs[2][0], outs[3][0], emb_h, attn


def swin_tiny_window7(args, **kwargs):
    return SwinTransformer(args=args, pretrain_img_size=224, patch_size=4, decoder_num_patches=49,
                           embed_dim=[96, 192, 384, 768], depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], window_size=7,
                           mlp_ratio=4., norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def swin_base_window7(args, **kwargs):
    """Swin-Base (BASELINE.json config 5 at its named size): the reference's class (swin.py:13-292) with the standard
    Swin-B stage plan -- depths 2-2-18-2, widths 128..1024, 4..32 heads (32 channels per head, as in Swin-T) -- for which
    the reference itself ships no factory (swin.py:295-302 is Swin-T only). Parity: tests/golden/rec_swin_base.npz, made by
    instantiating the reference's SwinTransformer with exactly these arguments."""
    return SwinTransformer(args=args, pretrain_img_size=224, patch_size=4, decoder_num_patches=49,
                           embed_dim=[128, 256, 512, 1024], depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=7,
                           mlp_ratio=4., norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
