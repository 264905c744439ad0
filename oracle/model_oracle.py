"""CPU restatement (torch fp32, functional) of the reference's pre-training hot path.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg. The product package (eventpretrain_amd/) never
imports this file; its ops raise when the HIP library is missing.

Pinned by tests/golden/*.npz, which oracle/gen_golden.py produced by running
the reference itself (tests/test_oracle_golden.py checks every function here
against them).

Every function takes a plain ``dict`` of tensors keyed by the reference's
state-dict names and cites the reference lines it restates.
Paths are relative to the reference root.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- utils/pos_embed.py
def sincos_1d(dim, pos):
    """utils/pos_embed.py:4-21 -- float32 omega, float32 positions, [sin | cos]."""
    assert dim % 2 == 0
    omega = np.arange(dim // 2, dtype=np.float32)
    omega /= dim / 2.0
    omega = 1.0 / 10000 ** omega
    ang = np.einsum("m,d->md", pos.reshape(-1), omega)
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)


def sincos_2d(dim, grid_size):
    """utils/pos_embed.py:23-55. meshgrid(w, h) "w goes first": the first half of the channels encodes the
    column index, the second half the row index (SURVEY.md section 4 known answers)."""
    ar = np.arange(grid_size, dtype=np.float32)
    col, row = np.meshgrid(ar, ar)         # col[i, j] = j, row[i, j] = i
    first = sincos_1d(dim // 2, col)
    second = sincos_1d(dim // 2, row)
    return np.concatenate([first, second], axis=1)


# ----------------------------------------------------------------------------- utils/reshape.py
def patchify(frame, p):
    """utils/reshape.py:15-22 frame2emb: (B,C,H,W) -> (B, L, p*p*C) with inner order (py, px, c)."""
    B, C, H, W = frame.shape
    gh, gw = H // p, W // p
    t = frame.reshape(B, C, gh, p, gw, p).permute(0, 2, 4, 3, 5, 1)
    return t.reshape(B, gh * gw, p * p * C)


# ----------------------------------------------------------------------------- model/backbone/vit.py:66-105
def masking_from_noise(noise, mask_ratio):
    """vit.py:75-103 with the noise as an explicit input. Stable ascending argsort (ties -> lower index first);
    ids_restore = inverse permutation; mask = 1 for removed tokens."""
    B, L = noise.shape
    keep = int(L * (1 - mask_ratio))
    order = torch.argsort(noise, dim=1, stable=True)
    restore = torch.argsort(order, dim=1, stable=True)
    mask = (restore >= keep).to(torch.float32)
    return order[:, :keep], mask, restore


def density_noise(x, patch, strategy):
    """vit.py:80-89: |sum over bins| average-pooled per patch; 'anti-density' negates."""
    d = F.avg_pool2d(x.sum(1).abs().unsqueeze(1), patch, patch).flatten(1)
    return d if strategy == "density" else -d


# ----------------------------------------------------------------------------- model/sub_module/vit_block.py
def layer_norm(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def attention(sd, pre, x, heads, attn_mask=None, attn_p=0.0):
    """vit_block.py:131-143: fused qkv Linear, scale d_h^-0.5, softmax, attn_drop, AV, proj. Returns (out, probs) -- the probabilities
    AFTER the dropout, as the reference returns them. `attn_mask` [B, heads, N, N] of 0 / 1 keeps with rate attn_p (training mode)."""
    B, N, C = x.shape
    dh = C // heads
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"]).reshape(B, N, 3, heads, dh)
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    p = torch.softmax((q @ k.transpose(-2, -1)) * dh ** -0.5, dim=-1)
    if attn_mask is not None:
        p = p * attn_mask.view_as(p) / (1.0 - attn_p)
    o = (p @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"]), p


def drop_path_scale(u, keep_prob):
    """timm 0.3.2 `drop_path` (the DropPath the reference imports, vit_block.py:5,241): per SAMPLE, random_tensor = floor(keep_prob +
    U[0,1)), output = x / keep_prob * random_tensor. timm is absent from /root/reference and from this image: published forward
    restated; given the draws `u` the arithmetic is pinned, the random stream itself is not (parity unpinned)."""
    return torch.floor(keep_prob + u) / keep_prob


def vit_block(sd, pre, x, heads, eps=1e-6, want_attn=False, drops=None):
    """vit_block.py:246-254 pre-LN residual block; Mlp = fc1, GELU(erf), fc2 (vit_block.py:225-231).
    `drops` (training-mode regularisers, vit_block.py:137-141,226-231,252-253): dict(u1, u2, keep_prob) for the two DropPath
    applications and optionally (p, proj, hidden, fc2) = dropout rate and 0/1 keep masks of the proj output, the MLP hidden
    and the fc2 output."""
    a, p = attention(sd, pre + "attn.", layer_norm(x, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], eps), heads,
                     attn_mask=None if drops is None else drops.get("attn"), attn_p=0.0 if drops is None else drops.get("attn_p", 0.0))
    s1 = s2 = 1.0
    if drops is not None:
        kp = drops.get("keep_prob", 1.0)
        if drops.get("u1") is not None:
            s1 = drop_path_scale(drops["u1"], kp).view(-1, 1, 1)
            s2 = drop_path_scale(drops["u2"], kp).view(-1, 1, 1)
        if drops.get("p"):
            a = a * drops["proj"].view_as(a) / (1.0 - drops["p"])
    x = x + s1 * a
    h = layer_norm(x, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], eps)
    h = F.gelu(F.linear(h, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"]))
    if drops is not None and drops.get("p"):
        h = h * drops["hidden"].view_as(h) / (1.0 - drops["p"])
    m = F.linear(h, sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    if drops is not None and drops.get("p"):
        m = m * drops["fc2"].view_as(m) / (1.0 - drops["p"])
    x = x + s2 * m
    return (x, p) if want_attn else x


def patch_embed(sd, pre, x, patch):
    """vit_block.py:60-68: Conv2d(k=s=p) -> LayerNorm over channels (eps 1e-5, nn.LayerNorm default) -> GELU.
    Returned token-major (B, L, D) (vit.py:111 flatten+permute)."""
    y = F.conv2d(x, sd[pre + "proj.weight"], sd[pre + "proj.bias"], stride=patch)
    y = y.flatten(2).transpose(1, 2)
    return F.gelu(layer_norm(y, sd[pre + "norm.weight"], sd[pre + "norm.bias"], 1e-5))


# ----------------------------------------------------------------------------- model/backbone/vit.py:107-156
def _depth(sd, pre):
    return 1 + max(int(k[len(pre):].split(".")[0]) for k in sd if k.startswith(pre))


def vit_masked(sd, x, noise, *, patch, heads, mask_ratio, fusion=True, pre="backbone."):
    """vit.py:108-130: mask ids, patch-embed all tokens, +pos, gather kept, blocks with taps after block index 1
    and 3, LN(sum of taps + last) when feature fusion is on."""
    ids_keep, mask, ids_restore = masking_from_noise(noise, mask_ratio)
    t = patch_embed(sd, pre + "patch_embed.", x, patch) + sd[pre + "pos_embed"]
    t = torch.gather(t, 1, ids_keep.unsqueeze(-1).expand(-1, -1, t.shape[-1]))
    taps = {}
    for i in range(_depth(sd, pre + "vit_block.")):
        t = vit_block(sd, f"{pre}vit_block.{i}.", t, heads)
        if i in (1, 3):
            taps[i] = t
    fused = taps[1] + taps[3] + t if fusion else t
    out = layer_norm(fused, sd[pre + "norm_layer.weight"], sd[pre + "norm_layer.bias"], 1e-6)
    return taps[1], taps[3], out, mask, ids_restore


def vit_dense(sd, x, *, patch, heads, pre="backbone."):
    """vit.py:132-156 (pretrain phases): all tokens, taps after block index 0 and 1, last block returns probs."""
    t = patch_embed(sd, pre + "patch_embed.", x, patch) + sd[pre + "pos_embed"]
    n = _depth(sd, pre + "vit_block.")
    taps = {}
    attn = None
    for i in range(n):
        if i < n - 1:
            t = vit_block(sd, f"{pre}vit_block.{i}.", t, heads)
        else:
            t, attn = vit_block(sd, f"{pre}vit_block.{i}.", t, heads, want_attn=True)
        if i in (0, 1):
            taps[i] = t
    return taps[0], taps[1], layer_norm(t, sd[pre + "norm_layer.weight"], sd[pre + "norm_layer.bias"], 1e-6), attn


# ----------------------------------------------------------------------------- model/backbone/convvit.py, conv_block.py
def conv_block(sd, pre, x, keep=None, drops=None):
    """conv_block.py:41-51 on NCHW maps: LN over channels (eps 1e-5), 1x1 conv, keep-mask multiply, depthwise 5x5
    (padding 2), 1x1 conv, residual; LN, 1x1 -> GELU -> 1x1, residual.
    `drops` (training mode; conv_block.py:19-21,35,43-49): dict(u1, u2, keep_prob) = the per-sample draws of the two DropPath
    applications, optionally p + 0/1 masks "hidden" / "fc2" [B, H*W, C'] (channels-last element order) of CMlp's two dropouts; the
    conv branch has no dropout of its own."""
    def ln(t, n):
        return layer_norm(t.permute(0, 2, 3, 1), sd[pre + n + ".weight"], sd[pre + n + ".bias"], 1e-5).permute(0, 3, 1, 2)

    def mask_map(t, key):            # channels-last [B, HW, C'] mask -> NCHW factor
        B_, C_, H_, W_ = t.shape
        return drops[key].view(B_, H_, W_, C_).permute(0, 3, 1, 2) / (1.0 - drops["p"])
    C = x.shape[1]
    s1 = s2 = 1.0
    if drops is not None and drops.get("u1") is not None:
        s1 = drop_path_scale(drops["u1"], drops["keep_prob"]).view(-1, 1, 1, 1)
        s2 = drop_path_scale(drops["u2"], drops["keep_prob"]).view(-1, 1, 1, 1)
    h = F.conv2d(ln(x, "norm1"), sd[pre + "conv1.weight"], sd[pre + "conv1.bias"])
    if keep is not None:
        h = keep * h
    h = F.conv2d(h, sd[pre + "attn.weight"], sd[pre + "attn.bias"], padding=2, groups=C)
    x = x + s1 * F.conv2d(h, sd[pre + "conv2.weight"], sd[pre + "conv2.bias"])
    h = F.gelu(F.conv2d(ln(x, "norm2"), sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"]))
    if drops is not None and drops.get("p"):
        h = h * mask_map(h, "hidden")
    m = F.conv2d(h, sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    if drops is not None and drops.get("p"):
        m = m * mask_map(m, "fc2")
    return x + s2 * m


def _patch_embed_map(sd, pre, x, p):
    """vit_block.py:60-68 kept as a (B,D,h,w) map."""
    y = F.conv2d(x, sd[pre + "proj.weight"], sd[pre + "proj.bias"], stride=p)
    y = layer_norm(y.permute(0, 2, 3, 1), sd[pre + "norm.weight"], sd[pre + "norm.bias"], 1e-5)
    return F.gelu(y).permute(0, 3, 1, 2)


def convvit_masked(sd, x, noise, *, heads, mask_ratio, fusion=True, pre="backbone."):
    """convvit.py:126-171: three stages with the keep mask up-sampled to 56x56 / 28x28 by block repetition, multi-scale
    fusion convs (k4s4, k2s2) gathered at the kept tokens, LN of the three-way sum."""
    ids_keep, mask, ids_restore = masking_from_noise(noise, mask_ratio)
    B = x.shape[0]
    g = int(round(mask.shape[1] ** 0.5))
    keep14 = (1 - mask).view(B, 1, g, g)
    gidx = lambda t: torch.gather(t, 1, ids_keep.unsqueeze(-1).expand(-1, -1, t.shape[-1]))
    t = _patch_embed_map(sd, pre + "patch_embed1.", x, 4)
    for i in range(_depth(sd, pre + "conv_block1.")):
        t = conv_block(sd, f"{pre}conv_block1.{i}.", t, keep14.repeat_interleave(4, 2).repeat_interleave(4, 3))
    l1 = t
    s1 = gidx(F.conv2d(t, sd[pre + "stage1_output_decode.weight"], sd[pre + "stage1_output_decode.bias"], stride=4).flatten(2).transpose(1, 2))
    t = _patch_embed_map(sd, pre + "patch_embed2.", t, 2)
    for i in range(_depth(sd, pre + "conv_block2.")):
        t = conv_block(sd, f"{pre}conv_block2.{i}.", t, keep14.repeat_interleave(2, 2).repeat_interleave(2, 3))
    l2 = t
    s2 = gidx(F.conv2d(t, sd[pre + "stage2_output_decode.weight"], sd[pre + "stage2_output_decode.bias"], stride=2).flatten(2).transpose(1, 2))
    t = _patch_embed_map(sd, pre + "patch_embed3.", t, 2).flatten(2).transpose(1, 2)
    t = F.linear(t, sd[pre + "patch_embed4.weight"], sd[pre + "patch_embed4.bias"]) + sd[pre + "pos_embed"]
    t = gidx(t)
    for i in range(_depth(sd, pre + "vit_block.")):
        t = vit_block(sd, f"{pre}vit_block.{i}.", t, heads)
    fused = s1 + s2 + t if fusion else t
    return l1, l2, layer_norm(fused, sd[pre + "norm_layer.weight"], sd[pre + "norm_layer.bias"], 1e-6), mask, ids_restore


def convvit_rec_step(sd, x, target, noise, cfg):
    l1, l2, lh, mask, ids_restore = convvit_masked(sd, x, noise, heads=cfg["heads"], mask_ratio=cfg["mask_ratio"])
    pred = rec_decoder(sd, lh, ids_restore, heads=cfg["dec_heads"])
    loss = rec_loss(pred, target, mask, cfg["patch"], cfg.get("norm_pix", True), cfg["mask_ratio"])
    return loss, l1, l2, lh, pred, mask, ids_restore


# ----------------------------------------------------------------------------- model/sub_module/swin_block.py, backbone/swin.py
def swin_knapsack(cap, wt):
    """swin_block.py:277-319: 0/1 knapsack with value == weight; back-tracking from the last item, an item is taken
    whenever the table value differs from the row above. Returns (best, increasing index list)."""
    n = len(wt)
    K = np.zeros((n + 1, cap + 1), dtype=np.int64)
    for i in range(1, n + 1):
        w = wt[i - 1]
        K[i] = K[i - 1]
        if w <= cap:
            K[i, w:] = np.maximum(K[i - 1, w:], K[i - 1, :cap + 1 - w] + w)
    res, c, picked = int(K[n, cap]), cap, []
    best = res
    for i in range(n, 0, -1):
        if res <= 0:
            break
        if res == K[i - 1, c]:
            continue
        picked.append(i - 1)
        res -= wt[i - 1]
        c -= wt[i - 1]
    return best, picked[::-1]


def swin_group_windows(cap, counts):
    """swin_block.py:322-347: greedy repetition of the knapsack over the windows that are still unassigned."""
    wt, ori = list(counts), list(range(len(counts)))
    sizes, groups = [], []
    while wt:
        best, idx = swin_knapsack(cap, wt)
        sizes.append(best)
        groups.append([ori[i] for i in idx])
        keep = [i for i in range(len(wt)) if i not in set(idx)]
        wt, ori = [wt[i] for i in keep], [ori[i] for i in keep]
    return sizes, groups


def _swin_mask_from_ids(gid):
    """swin_block.py:366-373: 0 where two slots carry the same non-negative window id, else -100."""
    same = (gid[:, :, None] == gid[:, None, :]) & ~((gid[:, :, None] == -1) & (gid[:, None, :] == -1))
    return torch.where(same, torch.zeros(()), torch.full((), -100.0))


def _swin_rel_idx(c, window):
    """swin_block.py:375-381 on coords [..., n, 2]."""
    d = c[..., :, None, :] - c[..., None, :, :] + (window - 1)
    return d[..., 0] * (2 * window - 1) + d[..., 1]


def swin_plan(coords, window, shift, n_tokens):
    """GroupingModule.prepare (swin_block.py:383-452). coords: LongTensor [n,2] (shared over the batch).
    Returns dict(mode, gather [nG*GS] (pads -> 0), scatter [n], mask [nG,GS,GS] f32, rel [nG,GS,GS] int64)."""
    wid = (coords + (window - shift) % window) // window
    wid = wid[:, 0] * coords.shape[0] + wid[:, 1]
    if n_tokens <= 2 * window * window:
        mask = _swin_mask_from_ids(wid[None])
        return dict(mode="masking", mask=mask, rel=_swin_rel_idx(coords[None], window), gs=n_tokens)
    order = torch.argsort(wid, stable=True)
    swid = wid[order]
    counts = torch.unique_consecutive(swid, return_counts=True)[1].tolist()
    gs = min(window * window, max(counts))
    sizes, groups = swin_group_windows(gs, counts)
    ord_spl, wid_spl = order.split(counts), swid.split(counts)
    gather, gids = [], []
    for n_el, g in zip(sizes, groups):
        pad = gs - n_el
        gather.append(F.pad(torch.cat([ord_spl[i] for i in g]), (0, pad), value=-1))
        gids.append(F.pad(torch.cat([wid_spl[i] for i in g]), (0, pad), value=-1))
    gather = torch.cat(gather)
    scatter = torch.argsort(gather, stable=True)[-sum(sizes):]
    gather = gather.clamp_min(0)
    mask = _swin_mask_from_ids(torch.stack(gids))
    rel = _swin_rel_idx(coords[gather].reshape(-1, gs, 2), window)
    rel = rel * (mask == 0)
    return dict(mode="grouping", gather=gather, scatter=scatter, mask=mask, rel=rel, gs=gs)


def window_attention(sd, pre, x, mask, rel, heads, attn_mask=None, attn_p=0.0):
    """swin_block.py:124-162: scaled q, + table[rel] (zeroed where masked) + mask, softmax, attn_drop (:152, for a GIVEN keep mask
    [Bg, heads, N, N]), AV, proj. Returns the dropped map, as the reference does (:157)."""
    Bg, N, C = x.shape
    dh = C // heads
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"]).reshape(Bg, N, 3, heads, dh)
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    att = (q * dh ** -0.5) @ k.transpose(-2, -1)
    blocked = mask != 0
    table = sd[pre + "relative_position_bias_table"]
    bias = table[rel.masked_fill(blocked, 0).reshape(-1)].view(-1, N, N, heads) * (~blocked).view(-1, N, N, 1).float()
    nG = bias.shape[0]
    att = att.view(Bg // nG, nG, heads, N, N) + bias.permute(0, 3, 1, 2).unsqueeze(0) + mask.view(1, nG, 1, N, N)
    p = torch.softmax(att.view(Bg, heads, N, N), dim=-1)
    if attn_mask is not None and attn_p > 0:
        p = p * attn_mask.view_as(p) / (1.0 - attn_p)
    o = (p @ v).transpose(1, 2).reshape(Bg, N, C)
    return F.linear(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"]), p


def swin_block(sd, pre, x, plan, heads, eps=1e-6, drops=None):
    """swin_block.py:260-273 wrapped in GroupingModule.group/merge (swin_block.py:454-466).
    `drops` (training mode; swin_block.py:157,257,270-271, Mlp.drop): as in vit_block -- the DropPath draws are per row of the GROUPED
    tensor (one per batch item x group: the reference applies drop_path to the (B * n_groups, gs, C) tensor), masks "proj" / "hidden"
    / "fc2" in the grouped element order; "attn" + "attn_p": keep mask and rate of the dropout on the attention probabilities."""
    B, n, C = x.shape
    if plan["mode"] == "grouping":
        x = x[:, plan["gather"]].reshape(-1, plan["gs"], C)
    a, p = window_attention(sd, pre + "attn.", layer_norm(x, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], eps),
                            plan["mask"], plan["rel"], heads, attn_mask=None if drops is None else drops.get("attn"),
                            attn_p=0.0 if drops is None else (drops.get("attn_p") or 0.0))
    s1 = s2 = 1.0
    if drops is not None:
        if drops.get("u1") is not None:
            s1 = drop_path_scale(drops["u1"], drops["keep_prob"]).view(-1, 1, 1)
            s2 = drop_path_scale(drops["u2"], drops["keep_prob"]).view(-1, 1, 1)
        if drops.get("p"):
            a = a * drops["proj"].view_as(a) / (1.0 - drops["p"])
    x = x + s1 * a
    h = layer_norm(x, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], eps)
    h = F.gelu(F.linear(h, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"]))
    if drops is not None and drops.get("p"):
        h = h * drops["hidden"].view_as(h) / (1.0 - drops["p"])
    m = F.linear(h, sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
    if drops is not None and drops.get("p"):
        m = m * drops["fc2"].view_as(m) / (1.0 - drops["p"])
    x = x + s2 * m
    if plan["mode"] == "grouping":
        x = x.reshape(B, -1, C)[:, plan["scatter"]]
    return x, p


def swin_patch_merging(sd, pre, x, vis, res, eps=1e-6):
    """swin_block.py:179-209. x [B,n,C] in row-major order of the visible cells, vis BoolTensor [res*res].
    2x2 neighbours are concatenated in the order (0,0),(1,0),(0,1),(1,1), LayerNorm(4C), Linear(4C->2C, no bias)."""
    B, n, C = x.shape
    rank = torch.cumsum(vis.long(), 0) - 1                       # row-major index among the visible cells
    yy, xx = torch.meshgrid(torch.arange(res), torch.arange(res), indexing="ij")
    blk = lambda t: t.reshape(res // 2, 2, res // 2, 2).permute(0, 2, 1, 3).reshape(-1, 4)
    vis_b, rank_b = blk(vis), blk(rank)
    rows = rank_b[vis_b.any(1)]                                   # [n/4, 4] in order (0,0),(0,1),(1,0),(1,1)
    t = torch.cat([x[:, rows[:, 0]], x[:, rows[:, 2]], x[:, rows[:, 1]], x[:, rows[:, 3]]], dim=-1)
    t = layer_norm(t, sd[pre + "norm.weight"], sd[pre + "norm.bias"], eps)
    t = F.linear(t, sd[pre + "reduction.weight"])
    vis_new = vis_b.any(1)
    c = torch.stack([yy[::2, ::2].reshape(-1) // 2, xx[::2, ::2].reshape(-1) // 2], -1)[vis_new]
    return t, c, vis_new


def swin_stages(sd, t, coords, vis, cfg, pre="backbone."):
    """BasicBlock.forward x4 (swin_block.py:523-553, swin.py:184-188). Returns per-stage (tokens, coords) and the
    last block's attention probabilities."""
    outs, attn = [], None
    res = cfg["input"] // 4
    W = cfg["window"]
    n_stage = len(cfg["depths"])
    for s in range(n_stage):
        win = min(W, res)
        plan0 = swin_plan(coords, win, 0, t.shape[1])
        plan1 = swin_plan(coords, win, W // 2, t.shape[1]) if win < res else plan0
        for i in range(cfg["depths"][s]):
            t, attn = swin_block(sd, f"{pre}swin_block.{s}.blocks.{i}.", t, plan0 if i % 2 == 0 else plan1, cfg["heads"][s])
        outs.append((t, coords))
        if s < n_stage - 1:
            t, coords, vis = swin_patch_merging(sd, f"{pre}swin_block.{s}.downsample.", t, vis, res)
            res //= 2
    return outs, attn


def swin_masked(sd, x, noise, cfg, fusion=True, pre="backbone."):
    """swin.py:174-246. The visibility pattern of sample 0 is applied to the WHOLE batch (swin.py:151), while the
    fusion convs' outputs are gathered with each sample's own ids_keep (swin.py:207) -- restated as is."""
    B = x.shape[0]
    ids_keep, mask, ids_restore = masking_from_noise(noise, cfg["mask_ratio"])
    t = F.conv2d(x, sd[pre + "patch_embed.proj.weight"], sd[pre + "patch_embed.proj.bias"], stride=4).flatten(2).transpose(1, 2)
    t = layer_norm(t, sd[pre + "patch_embed.norm.weight"], sd[pre + "patch_embed.norm.bias"], 1e-6)
    res = cfg["input"] // 4
    g = int(round(mask.shape[1] ** 0.5))
    rep = res // g
    vis = (mask[0] == 0).view(g, 1, g, 1).expand(g, rep, g, rep).reshape(-1)
    yy, xx = torch.meshgrid(torch.arange(res), torch.arange(res), indexing="ij")
    coords = torch.stack([yy.reshape(-1), xx.reshape(-1)], -1)[vis]
    t = t[:, vis]
    outs, attn = swin_stages(sd, t, coords, vis, cfg, pre)
    gidx = lambda e: torch.gather(e, 1, ids_keep.unsqueeze(-1).expand(-1, -1, e.shape[-1]))
    fused = outs[-1][0]
    if fusion:
        for s in range(len(outs) - 1):
            e, c = outs[s]
            r = res >> s
            dense = torch.zeros(B, r * r, e.shape[-1])
            dense[:, c[:, 0] * r + c[:, 1]] = e
            dense = dense.view(B, r, r, -1).permute(0, 3, 1, 2)
            k = r // g
            d = F.conv2d(dense, sd[f"{pre}stage{s + 1}_output_decode.weight"], sd[f"{pre}stage{s + 1}_output_decode.bias"], stride=k)
            fused = fused + gidx(d.flatten(2).transpose(1, 2))
    lh = layer_norm(fused, sd[pre + "norm_layer.weight"], sd[pre + "norm_layer.bias"], 1e-6)
    return outs, lh, mask, ids_restore, attn


def swin_rec_step(sd, x, target, noise, cfg):
    """PrHubModel.forward(is_rec=True) for the Swin backbone (pr_hub_model.py:194-204)."""
    outs, lh, mask, ids_restore, attn = swin_masked(sd, x, noise, cfg)
    pred = rec_decoder(sd, lh, ids_restore, heads=cfg["dec_heads"])
    loss = rec_loss(pred, target, mask, cfg["patch"], cfg.get("norm_pix", True), cfg["mask_ratio"])
    return loss, outs, lh, pred, mask, ids_restore, attn


# ----------------------------------------------------------------------------- model/pretrain/pr_rec_decoder.py:53-70
def rec_decoder(sd, x, ids_restore, *, heads, pre="pretrain_rec_decoder."):
    t = F.linear(x, sd[pre + "patch_embed.weight"], sd[pre + "patch_embed.bias"])
    B, n_keep, D = t.shape
    L = ids_restore.shape[1]
    filler = sd[pre + "mask_token"].expand(B, L - n_keep, D)
    t = torch.cat([t, filler], 1)
    t = torch.gather(t, 1, ids_restore.unsqueeze(-1).expand(-1, -1, D)) + sd[pre + "pos_embed"]
    for i in range(_depth(sd, pre + "vit_block.")):
        t = vit_block(sd, f"{pre}vit_block.{i}.", t, heads)
    t = layer_norm(t, sd[pre + "norm.weight"], sd[pre + "norm.bias"], 1e-6)
    return F.linear(t, sd[pre + "pred.weight"], sd[pre + "pred.bias"])


# ----------------------------------------------------------------------------- model/pretrain/pr_hub_model.py:125-141
def rec_loss(pred, target_frame, mask, patch, norm_pix=True, mask_ratio=0.5):
    """Per-patch normalised (UNBIASED variance, torch default) masked MSE."""
    tgt = patchify(target_frame, patch)
    if norm_pix:
        mu = tgt.mean(-1, keepdim=True)
        var = tgt.var(-1, keepdim=True)
        tgt = (tgt - mu) / (var + 1e-6) ** 0.5
    per = ((pred - tgt) ** 2).mean(-1)
    if mask_ratio == 0:
        return per.mean()
    return (mask * per).sum() / mask.sum()


def rec_step(sd, x, target, noise, cfg):
    """PrHubModel.forward(is_rec=True) for the ViT backbone (pr_hub_model.py:191-206)."""
    l1, l2, lh, mask, ids_restore = vit_masked(sd, x, noise, patch=cfg["patch"], heads=cfg["heads"],
                                               mask_ratio=cfg["mask_ratio"], fusion=cfg.get("fusion", True))
    pred = rec_decoder(sd, lh, ids_restore, heads=cfg["dec_heads"])
    loss = rec_loss(pred, target, mask, cfg["patch"], cfg.get("norm_pix", True), cfg["mask_ratio"])
    return loss, l1, l2, lh, pred, mask, ids_restore


# ----------------------------------------------------------------------------- contrastive stage
def batchnorm_tokens(x, w, b, rm, rv, training=True, eps=1e-5, momentum=0.1):
    """mlp_head.py:13,18 BatchNorm2d applied on the (B,C,h,w) view of (B,L,C) tokens
    (pr_hub_model.py:223-237): per-channel statistics over B*L rows, biased var for normalisation, unbiased for
    the running estimate. Returns (y, new_running_mean, new_running_var)."""
    B, L, C = x.shape
    flat = x.reshape(B * L, C)
    if training:
        mu = flat.mean(0)
        var = flat.var(0, unbiased=False)
        n = flat.shape[0]
        rm = (1 - momentum) * rm + momentum * mu.detach()
        rv = (1 - momentum) * rv + momentum * var.detach() * n / (n - 1)
    else:
        mu, var = rm, rv
    y = (flat - mu) / torch.sqrt(var + eps)
    if w is not None:
        y = y * w + b
    return y.reshape(B, L, C), rm, rv


def mlp_head(sd, pre, x, n_layers, training=True):
    """_build_mlp_2d (mlp_head.py:4-24): [Linear(no bias), BN, ReLU] x (n-1), Linear(no bias), BN(affine=False).
    Sequential indices: layer l -> Linear at 3l, BN at 3l+1."""
    stats = {}
    for l in range(n_layers):
        x = F.linear(x, sd[f"{pre}{3 * l}.weight"])
        bn = f"{pre}{3 * l + 1}."
        last = l == n_layers - 1
        x, rm, rv = batchnorm_tokens(x, None if last else sd[bn + "weight"], None if last else sd[bn + "bias"],
                                     sd[bn + "running_mean"], sd[bn + "running_var"], training)
        stats[bn + "running_mean"], stats[bn + "running_var"] = rm, rv
        if not last:
            x = torch.relu(x)
    return x, stats


def info_nce_queue(q, k, queue, T):
    """pr_hub_model.py:144-163: per-position InfoNCE, positive at class 0, negatives from queue (C, L, K)."""
    q = F.normalize(q, dim=-1)
    k = F.normalize(k, dim=-1)
    pos = (q * k).sum(-1, keepdim=True)
    neg = torch.einsum("blc,clk->blk", q, queue.detach())
    logits = torch.cat([pos, neg], -1) / T
    B, L, K1 = logits.shape
    loss = F.cross_entropy(logits.reshape(B * L, K1), torch.zeros(B * L, dtype=torch.long))
    return loss, k


def enqueue(queue, ptr, keys):
    """pr_hub_model.py:112-122: queue[:, :, ptr:ptr+B] = keys.T where .T on a 3-D tensor reverses all dims
    ((B,L,C) -> (C,L,B)); ptr advances modulo K; K % B must be 0."""
    B = keys.shape[0]
    K = queue.shape[2]
    assert K % B == 0
    queue = queue.clone()
    queue[:, :, ptr:ptr + B] = keys.detach().permute(2, 1, 0)
    return queue, (ptr + B) % K


def info_nce_inbatch(q, k_all, T, rank=0):
    """pr_hub_model.py:170-188: logits[n,l,m] = q[n,l,:].k_all[m,l,:] / T, label = n + N*rank."""
    q = F.normalize(q, dim=-1)
    k_all = F.normalize(k_all, dim=-1)
    logits = torch.einsum("nlc,mlc->nlm", q, k_all) / T
    N, L, M = logits.shape
    labels = (torch.arange(N) + N * rank).unsqueeze(-1).expand(N, L)
    return F.cross_entropy(logits.permute(0, 2, 1), labels)


def con_step(sd, x, clip_emb, cfg, training=True, rank=0, gather=None):
    """PrHubModel.forward(is_rec=False), ViT backbone (pr_hub_model.py:208-245).
    Returns (loss, emb_h_org, emb_h_proj, clip_org, clip_proj, attn, side) where `side` holds the buffer updates
    (BN running stats, queue, queue_ptr)."""
    _, _, emb_h, attn = vit_dense(sd, x, patch=cfg["patch"], heads=cfg["heads"])
    emb_h_org = emb_h.detach().clone()
    clip = layer_norm(clip_emb[:, 1:, :], sd["norm_clip_emb.weight"], sd["norm_clip_emb.bias"], 1e-5)
    clip_org = clip.detach().clone()
    clip_proj = F.linear(clip, sd["clip_emb_proj.weight"])
    h, s1 = mlp_head(sd, "emb_h_proj.", emb_h, 3, training)
    h, s2 = mlp_head(sd, "emb_h_pred.", h, 2, training)
    side = {**s1, **s2}
    if cfg.get("use_queue", True):
        loss, k = info_nce_queue(h, clip_proj, sd["queue"], cfg["T"])
        side["queue"], ptr = enqueue(sd["queue"], int(sd["queue_ptr"]), k)
        side["queue_ptr"] = torch.tensor([ptr])
    else:
        k = F.normalize(clip_proj, dim=-1)       # normalising twice is idempotent up to rounding; see info_nce_inbatch
        k_all = gather(clip_proj) if gather is not None else clip_proj
        loss = info_nce_inbatch(h, k_all, cfg["T"], rank)
    return loss, emb_h_org, h, clip_org, clip_proj, attn, side


def swin_dense(sd, x, cfg, pre="backbone."):
    """swin.py:248-290 (pretrain phases): every cell visible, LN of the last stage; returns the last block's probs."""
    t = F.conv2d(x, sd[pre + "patch_embed.proj.weight"], sd[pre + "patch_embed.proj.bias"], stride=4).flatten(2).transpose(1, 2)
    t = layer_norm(t, sd[pre + "patch_embed.norm.weight"], sd[pre + "patch_embed.norm.bias"], 1e-6)
    res = cfg["input"] // 4
    yy, xx = torch.meshgrid(torch.arange(res), torch.arange(res), indexing="ij")
    coords = torch.stack([yy.reshape(-1), xx.reshape(-1)], -1)
    outs, attn = swin_stages(sd, t, coords, torch.ones(res * res, dtype=torch.bool), cfg, pre)
    return outs, layer_norm(outs[-1][0], sd[pre + "norm_layer.weight"], sd[pre + "norm_layer.bias"], 1e-6), attn


def swin_con_step(sd, x, clip_emb, cfg, training=True):
    """PrHubModel.forward(is_rec=False) for the Swin backbone with the queue (pr_hub_model.py:208-245): the CLIP tokens
    go through Conv2d(512, 768, 2, stride 2) on their 14x14 grid (:219-220) to meet the 7x7 grid of the last stage."""
    _, emb_h, attn = swin_dense(sd, x, cfg)
    emb_h_org = emb_h.detach().clone()
    clip = layer_norm(clip_emb[:, 1:, :], sd["norm_clip_emb.weight"], sd["norm_clip_emb.bias"], 1e-5)
    clip_org = clip.detach().clone()
    B, L, Cc = clip.shape
    g = int(round(L ** 0.5))
    cmap = clip.transpose(1, 2).reshape(B, Cc, g, g)
    clip_proj = F.conv2d(cmap, sd["clip_emb_proj.weight"], sd["clip_emb_proj.bias"], stride=2).flatten(2).transpose(1, 2)
    h, s1 = mlp_head(sd, "emb_h_proj.", emb_h, 3, training)
    h, s2 = mlp_head(sd, "emb_h_pred.", h, 2, training)
    side = {**s1, **s2}
    loss, k = info_nce_queue(h, clip_proj, sd["queue"], cfg["T"])
    side["queue"], ptr = enqueue(sd["queue"], int(sd["queue_ptr"]), k)
    side["queue_ptr"] = torch.tensor([ptr])
    return loss, emb_h_org, h, clip_org, clip_proj, attn, side


# ----------------------------------------------------------------------------- model/finetune_cls/ft_cls_hub_model.py:118-139
def ft_cls_step(sd, x, label, cfg):
    """Dense backbone (vit.py:132-156 / swin.py:248-290) -> mean over tokens -> Linear head -> mean cross-entropy
    (trainer/finetune_cls/ft_cls_trainer.py:66). Returns (loss, pred, emb_h, attn)."""
    if cfg["backbone"] == "swin":
        _, emb_h, attn = swin_dense(sd, x, cfg)
    else:
        _, _, emb_h, attn = vit_dense(sd, x, patch=cfg["patch"], heads=cfg["heads"])
    pred = F.linear(emb_h.mean(dim=1), sd["classify_head.weight"], sd["classify_head.bias"])
    return F.cross_entropy(pred, label), pred, emb_h, attn


def label_smoothing_ce(pred, label, smoothing):
    """timm 0.3.2 `LabelSmoothingCrossEntropy(smoothing)` as trainer/finetune_cls/ft_cls_trainer.py:63-64 applies it. timm
    is a pip dependency that is absent from /root/reference and from this image; its published forward is restated:
    logp = log_softmax(x); loss = ((1-s) * -logp[label] + s * -logp.mean(-1)).mean(). Parity unpinned by a reference
    fixture; cross-checked against torch's F.cross_entropy(label_smoothing=s), the same formula
    (tests/test_oracle_golden.py)."""
    logp = F.log_softmax(pred, dim=-1)
    nll = -logp.gather(-1, label.view(-1, 1)).squeeze(1)
    smooth = -logp.mean(dim=-1)
    return ((1.0 - smoothing) * nll + smoothing * smooth).mean()


def clip_coef(total_norm, max_norm):
    """torch.nn.utils.clip_grad_norm_ (reference utils/misc.py:289-290): factor applied to every gradient."""
    return min(1.0, max_norm / (total_norm + 1e-6))


# ----------------------------------------------------------------------------- optimiser-side glue
def cosine_lr(epoch, lr, min_lr, warmup_epochs, epochs):
    """utils/lr_sched.py:3-16."""
    if epoch < warmup_epochs:
        return lr * epoch / warmup_epochs
    return min_lr + (lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * (epoch - warmup_epochs) / (epochs - warmup_epochs)))


def decay_split(named_shapes):
    """utils/lr_decay.py:41-47 with layer_decay=1: weight decay on every trainable parameter with ndim > 1
    (mask_token (1,1,D) included), none on 1-D ones."""
    decay = [n for n, s in named_shapes if len(s) > 1]
    no_decay = [n for n, s in named_shapes if len(s) <= 1]
    return decay, no_decay


def adamw_step(p, g, m, v, step, lr, wd, beta1=0.9, beta2=0.95, eps=1e-8):
    """torch.optim.AdamW single-tensor update (main_pretrain.py:341-343 betas): decoupled decay first, then
    bias-corrected Adam with denom = sqrt(v)/sqrt(1-b2^t) + eps."""
    p = p * (1 - lr * wd)
    m = beta1 * m + (1 - beta1) * g
    v = beta2 * v + (1 - beta2) * g * g
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v


def grad_norm(grads):
    """utils/misc.py:303-315: 2-norm of the per-parameter 2-norms."""
    return torch.norm(torch.stack([torch.norm(g.detach(), 2.0) for g in grads]), 2.0)
