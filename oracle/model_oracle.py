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


def attention(sd, pre, x, heads):
    """vit_block.py:131-143: fused qkv Linear, scale d_h^-0.5, softmax, AV, proj. Returns (out, probs)."""
    B, N, C = x.shape
    dh = C // heads
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"]).reshape(B, N, 3, heads, dh)
    q, k, v = qkv.permute(2, 0, 3, 1, 4)
    p = torch.softmax((q @ k.transpose(-2, -1)) * dh ** -0.5, dim=-1)
    o = (p @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"]), p


def vit_block(sd, pre, x, heads, eps=1e-6, want_attn=False):
    """vit_block.py:246-254 pre-LN residual block; Mlp = fc1, GELU(erf), fc2 (vit_block.py:225-231)."""
    a, p = attention(sd, pre + "attn.", layer_norm(x, sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], eps), heads)
    x = x + a
    h = layer_norm(x, sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], eps)
    h = F.gelu(F.linear(h, sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"]))
    x = x + F.linear(h, sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])
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
def conv_block(sd, pre, x, keep=None):
    """conv_block.py:41-51 on NCHW maps: LN over channels (eps 1e-5), 1x1 conv, keep-mask multiply, depthwise 5x5
    (padding 2), 1x1 conv, residual; LN, 1x1 -> GELU -> 1x1, residual."""
    def ln(t, n):
        return layer_norm(t.permute(0, 2, 3, 1), sd[pre + n + ".weight"], sd[pre + n + ".bias"], 1e-5).permute(0, 3, 1, 2)
    C = x.shape[1]
    h = F.conv2d(ln(x, "norm1"), sd[pre + "conv1.weight"], sd[pre + "conv1.bias"])
    if keep is not None:
        h = keep * h
    h = F.conv2d(h, sd[pre + "attn.weight"], sd[pre + "attn.bias"], padding=2, groups=C)
    x = x + F.conv2d(h, sd[pre + "conv2.weight"], sd[pre + "conv2.bias"])
    h = F.gelu(F.conv2d(ln(x, "norm2"), sd[pre + "mlp.fc1.weight"], sd[pre + "mlp.fc1.bias"]))
    return x + F.conv2d(h, sd[pre + "mlp.fc2.weight"], sd[pre + "mlp.fc2.bias"])


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
