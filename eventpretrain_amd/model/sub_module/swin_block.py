"""Swin building blocks on sparse (visible-only) tokens with the reference's class names, constructor arguments and
state-dict keys (reference model/sub_module/swin_block.py). nn.Linear / nn.LayerNorm / nn.Conv2d members only hold
parameters: every forward runs on the HIP kernels of libevtpretrain.so (eventpretrain_amd.ops).

Division of labour. Which tokens are visible is one 49-entry pattern per step (the reference applies sample 0's mask
to the whole batch, swin.py:151), so everything that depends only on the pattern -- window ids, the knapsack
packing of windows into groups, gather/scatter index tables, relative-position indices -- is HOST work
(`GroupingModule.prepare`, numpy + the C knapsack `evp_swin_group_windows`) done once per pattern and uploaded as
small int32 tables. Everything that touches activations is a kernel: row gathers (group / merge / 2x2 merging), the
fused block (`ops.SwinBlockFn`) and the window attention with the gathered bias (csrc/window.hip).

Where the reference passes a float mask (0 / -100) and a separate index tensor to WindowAttention, this port passes
ONE int32 tensor `rel` [n_groups, N, N]: the relative-position index where a pair may attend, -1 where the reference
zeroes the bias and adds -100 (swin_block.py:140-149)."""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from ... import _lib, ops
from .vit_block import _pair


def get_coordinates(h, w, device="cpu"):
    """(2, h, w) integer grid, row index first (swin_block.py:67-71)."""
    ys, xs = torch.meshgrid(torch.arange(h, device=device), torch.arange(w, device=device), indexing="ij")
    return torch.stack([ys, xs])


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        self.drop_rate = float(drop)                # applied by the block's fused function (ops.BlockDrop)
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)


class PatchEmbed(nn.Module):
    """Conv2d(k=s=patch) -> optional LayerNorm, token-major output (swin_block.py:26-64)."""

    def __init__(self, img_size=224, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None):
        super().__init__()
        self.img_size = _pair(img_size)
        self.patch_size = _pair(patch_size)
        self.patches_resolution = [self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1]]
        self.num_patches = self.patches_resolution[0] * self.patches_resolution[1]
        self.in_chans = in_chans
        self.embed_dim = embed_dim
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def forward(self, x, ids_keep=None):
        """(B,C,H,W) f32 -> (B, n, D) f32; with `ids_keep` (int64 [B,n]) only those tokens are formed."""
        B, Cc, H, W = x.shape
        if (H, W) != self.img_size:
            raise AssertionError(f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]}).")
        t = ops.PatchProjFn.apply(x, ids_keep, self.proj.weight, self.proj.bias, self.patch_size[0])
        if self.norm is not None:
            t = ops.LayerNormFn.apply(t, None, None, self.norm.weight, self.norm.bias, self.norm.eps)
        return t


class WindowAttention(nn.Module):
    """Parameter holder (relative_position_bias_table, qkv, proj) + the unused `relative_position_index` buffer the
    reference keeps for checkpoint compatibility (swin_block.py:90-110). The math runs inside ops.SwinBlockFn."""

    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.attn_drop_rate = float(attn_drop)      # swin_block.py:113,152: applied inside ops.SwinBlockFn (keep flags into the LDS kernels)
        self.proj_drop_rate = float(proj_drop)
        if qk_scale is not None:
            raise NotImplementedError("qk_scale override is not used on the pre-training path")
        if not qkv_bias:
            raise NotImplementedError("qkv_bias=False is not used on the pre-training path")
        self.dim = dim
        self.window_size = _pair(window_size)
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        wh, ww = self.window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * wh - 1) * (2 * ww - 1), num_heads))
        flat = torch.flatten(get_coordinates(wh, ww), 1)
        rel = (flat[:, :, None] - flat[:, None, :]).permute(1, 2, 0).contiguous()
        rel[:, :, 0] += wh - 1
        rel[:, :, 1] += ww - 1
        rel[:, :, 0] *= 2 * ww - 1
        self.register_buffer("relative_position_index", rel.sum(-1))
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)

    def extra_repr(self):
        return f"dim={self.dim}, window_size={self.window_size}, num_heads={self.num_heads}"


class TokenLayout:
    """Host description of a sparse token set: `coords` int64 [n,2] (row, col) in row-major order and the boolean
    visibility map `vis` [res*res] it was cut from. What the reference carries as (coords, patch_mask) tensors."""

    def __init__(self, coords, vis, res, coords_dev=None):
        self.coords = np.ascontiguousarray(coords, dtype=np.int64)
        self.vis = np.ascontiguousarray(vis, dtype=bool)
        self.res = int(res)
        self.coords_dev = coords_dev      # int64 [1,n,2] already in device memory (static plan), else made on demand

    @property
    def n(self):
        return self.coords.shape[0]

    def coords_tensor(self, device):
        if self.coords_dev is not None:
            return self.coords_dev
        return torch.from_numpy(self.coords).unsqueeze(0).to(device)

    def mask_tensor(self, device):
        return torch.from_numpy(self.vis).unsqueeze(0).to(device)


def _dev_i32(a, device):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(device, non_blocking=True)


class PatchMerging(nn.Module):
    """2x2 neighbourhood concat in the order (0,0),(1,0),(0,1),(1,1) -> LayerNorm(4C) -> Linear(4C, 2C, bias=False)
    on the visible tokens only (swin_block.py:165-212). The regrouping is one row gather with a host-built index."""

    def __init__(self, input_resolution, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.input_resolution = input_resolution
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)

    @staticmethod
    def plan(layout):
        """-> (idx_fwd int32 [n], idx_bwd int32 [n], TokenLayout of the merged grid)."""
        res = layout.res
        if res % 2:
            raise AssertionError(f"x size ({res}*{res}) are not even.")
        vis = layout.vis.reshape(res, res)
        rank = (np.cumsum(layout.vis) - 1).reshape(res, res)
        blk = lambda a: a.reshape(res // 2, 2, res // 2, 2).transpose(0, 2, 1, 3).reshape(-1, 4)
        vis_b, rank_b = blk(vis), blk(rank)
        any_b, all_b = vis_b.any(1), vis_b.all(1)
        if not np.array_equal(any_b, all_b):
            raise AssertionError("a 2x2 merging neighbourhood is only partly visible; the masking grid must be coarser")
        rows = rank_b[any_b][:, [0, 2, 1, 3]].reshape(-1)
        inv = np.empty_like(rows)
        inv[rows] = np.arange(rows.shape[0])
        ys, xs = np.nonzero(any_b.reshape(res // 2, res // 2))
        return rows, inv, TokenLayout(np.stack([ys, xs], -1), any_b, res // 2)

    def forward(self, x, layout, plan=None):
        B, L, Cc = x.shape
        rows, inv, new_layout = plan if plan is not None else self.plan(layout)
        dev = x.device
        if not torch.is_tensor(rows):
            rows, inv = _dev_i32(rows, dev), _dev_i32(inv, dev)
        t = ops.GatherRowsFn.apply(x, rows, inv).view(B, L // 4, 4 * Cc)
        t = ops.LayerNormFn.apply(t, None, None, self.norm.weight, self.norm.bias, self.norm.eps)
        t = ops.LinearFn.apply(t, self.reduction.weight, None)
        return t, new_layout

    def extra_repr(self):
        return f"input_resolution={self.input_resolution}, dim={self.dim}"


class SwinTransformerBlock(nn.Module):
    """x = x + WindowAttn(LN(x)); x = x + Mlp(LN(x)) on grouped tokens: one fused autograd node (ops.SwinBlockFn)."""

    def __init__(self, dim, input_resolution, num_heads, window_size=7, shift_size=0, mlp_ratio=4., qkv_bias=True,
                 qk_scale=None, drop=0., attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        self.drop_path_rate = float(drop_path)      # swin_block.py:257,270-271: per group instance (x.shape[0]), training mode only
        self.drop_rate = float(drop)
        self.attn_drop_rate = float(attn_drop)      # swin_block.py:113,152 (ops.draw_block_drop reads it)
        self.dim = dim
        self.input_resolution = input_resolution
        self.num_heads = num_heads
        self.window_size = window_size
        self.shift_size = shift_size
        self.mlp_ratio = mlp_ratio
        if min(self.input_resolution) <= self.window_size:
            self.shift_size = 0
            self.window_size = min(self.input_resolution)
        if not 0 <= self.shift_size < self.window_size:
            raise AssertionError("shift_size must in 0-window_size")
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, window_size=_pair(self.window_size), num_heads=num_heads, qkv_bias=qkv_bias,
                                    qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), drop=drop)

    def forward(self, x, rel, return_attn=False, block_drop=None):
        rd = block_drop if block_drop is not None else ops.draw_block_drop(self, x.shape[0], x.device)
        return ops.swin_block(x, self, rel, self.norm1.eps, want_attn=return_attn, rd=rd)

    def extra_repr(self):
        return (f"dim={self.dim}, input_resolution={self.input_resolution}, num_heads={self.num_heads}, "
                f"window_size={self.window_size}, shift_size={self.shift_size}, mlp_ratio={self.mlp_ratio}")


# ------------------------------------------------------------------------------------------------ host grouping
def knapsack(W, wt):
    """0/1 knapsack with value == weight (swin_block.py:277-319) -> (best, increasing index list). Row-vectorised
    table; the back-tracking rule (take item i when the table value differs from the row above) is the reference's."""
    n = len(wt)
    K = np.zeros((n + 1, W + 1), dtype=np.int64)
    for i in range(1, n + 1):
        w = int(wt[i - 1])
        K[i] = K[i - 1]
        if w <= W:
            np.maximum(K[i - 1, w:], K[i - 1, :W + 1 - w] + w, out=K[i, w:])
    res = best = int(K[n, W])
    cap, idx = W, []
    for i in range(n, 0, -1):
        if res <= 0:
            break
        if res == K[i - 1, cap]:
            continue
        idx.append(i - 1)
        res -= int(wt[i - 1])
        cap -= int(wt[i - 1])
    return best, idx[::-1]


def group_windows(group_size, num_ele_win):
    """Greedy repetition of the knapsack over the still-unassigned windows (swin_block.py:322-347), evaluated by the
    C routine evp_swin_group_windows -> (tokens per group, window indices per group)."""
    counts = np.ascontiguousarray(num_ele_win, dtype=np.int32)
    n = counts.shape[0]
    group_of = np.zeros(n, dtype=np.int32)
    sizes = np.zeros(max(n, 1), dtype=np.int32)
    ng = C.c_int32(0)
    _lib.call("evp_swin_group_windows", counts.ctypes.data, n, int(group_size), group_of.ctypes.data, sizes.ctypes.data,
              C.addressof(ng))
    groups = [np.nonzero(group_of == g)[0].tolist() for g in range(ng.value)]
    return sizes[:ng.value].tolist(), groups


class PlanOverflow(Exception):
    """A visibility pattern needs more window groups than a fixed-shape plan was sized for."""


class GroupingModule:
    """Packs the visible tokens of the (shifted) windows into equally sized groups, or -- for at most 2*ws*ws tokens --
    keeps them as one group under a mask (swin_block.py:350-466). `prepare` is host work on the token coordinates and
    returns the device table `rel`; `group` / `merge` are row gathers."""

    def __init__(self, window_size, shift_size, group_size=None):
        if not 0 <= shift_size < window_size:
            raise AssertionError("shift_size must be in [0, window_size)")
        self.window_size = window_size
        self.shift_size = shift_size
        self.group_size = group_size or window_size ** 2
        self._mode = None

    def _window_id(self, coords):
        ws = self.window_size
        w = (coords + (ws - self.shift_size) % ws) // ws
        return w[:, 0] * coords.shape[0] + w[:, 1]

    def _rel_index(self, c):
        """relative-position index of every token pair of a group, int32 [..., n, n] (swin_block.py:375-381); rows and
        columns are handled separately in int32 -- the [..., n, n, 2] int64 form cost 10 ms of host time per step"""
        ws = self.window_size
        cy = np.ascontiguousarray(c[..., 0], dtype=np.int32)
        cx = np.ascontiguousarray(c[..., 1], dtype=np.int32)
        out = cy[..., :, None] - cy[..., None, :]
        out += ws - 1
        out *= 2 * ws - 1
        out += cx[..., :, None]
        out -= cx[..., None, :]
        out += ws - 1
        return out

    def plan(self, coords, num_tokens, fixed=None):
        """coords int64 [n,2] (host). -> dict of host arrays (see prepare).
        `fixed=(group_size, n_groups)`: a plan of that exact shape for ANY pattern (what a captured HIP graph needs): the
        knapsack packs into groups of `group_size` instead of the pattern's largest window population, and the group list is
        padded with empty groups (every slot a masked copy of token 0, exactly how the reference pads a partly filled
        group, swin_block.py:418-452). Masked pairs get -100 before the softmax, so the result differs from the
        pattern-sized plan only by f32 summation order. Raises PlanOverflow when the pattern needs more groups."""
        coords = np.asarray(coords, dtype=np.int64).reshape(-1, 2)
        wid = self._window_id(coords)
        ws = self.window_size
        if num_tokens <= 2 * ws * ws:
            rel = self._rel_index(coords)
            rel[wid[:, None] != wid[None, :]] = -1
            rel = rel[None]
            return dict(mode="masking", rel=rel, group_size=int(num_tokens), n_groups=1)
        order = np.argsort(wid, kind="stable")
        swid = wid[order]
        starts = np.nonzero(np.r_[True, swid[1:] != swid[:-1]])[0]
        counts = np.diff(np.r_[starts, swid.shape[0]])
        gs = int(min(ws * ws, counts.max())) if fixed is None else int(fixed[0])
        if counts.max() > gs:
            raise PlanOverflow(f"a window holds {int(counts.max())} tokens, fixed group size is {gs}")
        sizes, groups = group_windows(gs, counts)
        ng = len(groups)
        if fixed is not None:
            if ng > int(fixed[1]):
                raise PlanOverflow(f"pattern needs {ng} groups, the fixed plan has {int(fixed[1])}")
            ng = int(fixed[1])
        slot_tok = np.full((ng, gs), -1, dtype=np.int64)
        slot_wid = np.full((ng, gs), -1, dtype=np.int64)
        for g, wins in enumerate(groups):
            o = 0
            for w in wins:
                s, c = starts[w], counts[w]
                slot_tok[g, o:o + c] = order[s:s + c]
                slot_wid[g, o:o + c] = swid[s:s + c]
                o += c
        flat = slot_tok.reshape(-1)
        real = flat >= 0
        tok_slot = np.empty(coords.shape[0], dtype=np.int64)
        tok_slot[flat[real]] = np.nonzero(real)[0]
        cs = coords[np.where(real, flat, 0)].reshape(ng, gs, 2)
        sw = slot_wid.astype(np.int32)
        rel = self._rel_index(cs)
        rel[(sw[:, :, None] != sw[:, None, :]) | (sw[:, :, None] < 0)] = -1
        return dict(mode="grouping", rel=rel, group_size=gs, n_groups=ng, gather=np.where(real, flat, 0), gather_adj=tok_slot,
                    scatter=tok_slot, scatter_adj=flat)

    def prepare(self, coords, num_tokens, device="cuda"):
        p = self.plan(coords, num_tokens)
        self._mode = p["mode"]
        self.group_size = p["group_size"]
        self.n_groups = p["n_groups"]
        self.rel = _dev_i32(p["rel"], device)
        if self._mode == "grouping":
            self.idx_shuffle, self.idx_shuffle_adj = _dev_i32(p["gather"], device), _dev_i32(p["gather_adj"], device)
            self.idx_unshuffle, self.idx_unshuffle_adj = _dev_i32(p["scatter"], device), _dev_i32(p["scatter_adj"], device)
        return self.rel

    def bind(self, mode, group_size, n_groups, rel, tables=None):
        """Use tables that already live in device memory (a static plan's views: fixed addresses, refreshed by one H2D copy
        per step) instead of uploading fresh ones."""
        self._mode, self.group_size, self.n_groups, self.rel = mode, int(group_size), int(n_groups), rel
        if mode == "grouping":
            self.idx_shuffle, self.idx_shuffle_adj, self.idx_unshuffle, self.idx_unshuffle_adj = tables
        return self

    def group(self, x):
        if self._mode == "grouping":
            x = ops.GatherRowsFn.apply(x, self.idx_shuffle, self.idx_shuffle_adj)
            x = x.reshape(-1, self.group_size, x.shape[-1])
        return x

    def merge(self, x, batch):
        if self._mode == "grouping":
            x = x.reshape(batch, -1, x.shape[-1])
            x = ops.GatherRowsFn.apply(x, self.idx_unshuffle, self.idx_unshuffle_adj)
        return x


class StagePlan:
    """Everything one BasicBlock needs for one visibility pattern: the two GroupingModules and the merging plan."""

    def __init__(self, block, layout, device):
        n = layout.n
        self.layout = layout
        self.plain = GroupingModule(block.window_size, 0)
        self.plain.prepare(layout.coords, n, device)
        if block.window_size < min(block.input_resolution):
            self.shifted = GroupingModule(block.window_size, block.shift_size)
            self.shifted.prepare(layout.coords, n, device)
        else:
            self.shifted = self.plain
        self.merge = None
        if block.downsample is not None:
            rows, inv, new_layout = PatchMerging.plan(layout)
            self.merge = (_dev_i32(rows, device), _dev_i32(inv, device), new_layout)


class BasicBlock(nn.Module):
    """One Swin stage: `depth` blocks alternating plain / shifted windows, then PatchMerging (swin_block.py:469-557)."""

    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop=0., attn_drop=0., drop_path=0., norm_layer=nn.LayerNorm, downsample=None, use_checkpoint=False):
        super().__init__()
        if use_checkpoint:
            raise NotImplementedError("activation checkpointing is not used on the pre-training path")
        self.dim = dim
        self.input_resolution = input_resolution
        self.depth = depth
        self.window_size = window_size
        if min(self.input_resolution) <= self.window_size:
            self.shift_size = 0
            self.window_size = min(self.input_resolution)
        else:
            self.shift_size = window_size // 2
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim=dim, input_resolution=input_resolution, num_heads=num_heads, window_size=window_size,
                                 shift_size=0 if (i % 2 == 0) else window_size // 2, mlp_ratio=mlp_ratio,
                                 qkv_bias=qkv_bias, qk_scale=qk_scale, drop=drop, attn_drop=attn_drop,
                                 drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path, norm_layer=norm_layer)
            for i in range(depth)])
        self.downsample = downsample(input_resolution, dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def forward(self, x, layout, plan=None, return_attn=False):
        """x f32 [B,n,C], layout TokenLayout. -> (x, layout, x_down, layout_down, attn) or (x, layout, attn) for the last
        stage; attn (probabilities of the stage's last block) only when return_attn."""
        plan = plan if plan is not None else StagePlan(self, layout, x.device)
        B = x.shape[0]
        attn = None
        last = len(self.blocks) - 1
        for i, blk in enumerate(self.blocks):
            gm = plan.plain if i % 2 == 0 else plan.shifted
            x = gm.group(x)
            if return_attn and i == last:
                x, attn = blk(x, gm.rel, return_attn=True)
            else:
                x = blk(x, gm.rel)
            x = gm.merge(x, B)
        if self.downsample is not None:
            x_down, layout_down = self.downsample(x, layout, plan.merge)
            return x, layout, x_down, layout_down, attn
        return x, layout, attn

    def extra_repr(self):
        return (f"dim={self.dim}, input_resolution={self.input_resolution}, window_size={self.window_size},"
                f"shift_size={self.shift_size}, depth={self.depth}")
