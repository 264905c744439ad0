"""Plain ViT backbone without cls token, MAE-style masking and 3-tap feature fusion
(reference model/backbone/vit.py:11-171: same constructor, factories, return tuples and state-dict keys)."""
from functools import partial

import torch
import torch.nn as nn

from ... import ops
from ...utils.pos_embed import get_2d_sincos_pos_embed
from ...utils.reshape import emb2patch_frame
from ..sub_module.vit_block import PatchEmbed, ViTBlock


def init_linear_and_norm(m):
    """xavier-uniform Linear weights, zero biases, LayerNorm (1, 0)  (reference vit.py:56-64)."""
    if isinstance(m, nn.Linear):
        nn.init.xavier_uniform_(m.weight)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)


class ViT(nn.Module):
    def __init__(self, args, input_size=224, patch_size=16, embed_dim=1024,
                 depth=24, num_heads=16, mlp_ratio=4., out_indices=[3, 5, 7, 11], norm_layer=nn.LayerNorm,
                 num_bins=5, mask_ratio=0., drop_rate=0., attn_drop_rate=0., drop_path_rate=0.):
        super().__init__()
        self.args = args
        self.patch_size = patch_size
        self.out_indices = out_indices
        self.patch_embed = PatchEmbed(img_size=input_size, patch_size=patch_size, in_chans=num_bins, embed_dim=embed_dim)
        self.num_patches = self.patch_embed.num_patches
        self.pos_embed = nn.Parameter(torch.zeros(1, self.num_patches, embed_dim), requires_grad=False)
        self.drop_rate = float(drop_rate)          # pos_drop (vit.py:26,114,137)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, depth)]      # stochastic depth decay rule (vit.py:28)
        self.vit_block = nn.ModuleList([
            ViTBlock(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=True, qk_scale=None,
                     drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer) for i in range(depth)])
        if args.phase == "pretrain" and args.pr_phase in ("rec", "rec+con", "rec-n"):
            self.mask_ratio = mask_ratio
        self.norm_layer = norm_layer(embed_dim)
        self.initialize_weights()

    def initialize_weights(self):
        table = get_2d_sincos_pos_embed(self.pos_embed.shape[-1], int(self.num_patches ** .5), cls_token=False)
        self.pos_embed.data.copy_(torch.from_numpy(table).float().unsqueeze(0))
        w = self.patch_embed.proj.weight.data
        nn.init.xavier_uniform_(w.view(w.shape[0], -1))     # the conv is a Linear over flattened patches
        self.apply(init_linear_and_norm)

    # ------------------------------------------------------------------------------------------------ masking
    def masking_noise(self, x):
        """Noise whose ascending order decides what is kept (reference vit.py:77-89)."""
        strategy = self.args.masking_strategy
        if strategy == "random":
            return torch.rand(x.shape[0], self.num_patches, device=x.device)
        if strategy in ("density", "anti-density"):
            return ops.density_noise(x.detach(), self.patch_size, 1.0 if strategy == "density" else -1.0)
        raise ValueError(strategy)

    def random_masking(self, x, noise=None):
        """-> ids_keep (B, keep) int64, mask (B, L) f32 with 1 = removed, ids_restore (B, L) int64.
        `noise` may be passed explicitly (parity tests / reproducible runs); ties are broken by index (stable)."""
        if noise is None:
            noise = self.masking_noise(x)
        return ops.mask_from_noise(noise.contiguous().float(), self.mask_ratio)

    # ------------------------------------------------------------------------------------------------ forward
    def pos_drop(self, t):
        """nn.Dropout(p=drop_rate) on the embedded tokens (vit.py:114,137): training mode only."""
        if self.training and self.drop_rate > 0:
            t = ops.DropoutFn.apply(t, self.drop_rate, ops.draw_drop_seed(t.device))
        return t

    def forward(self, x, mask=False, noise=None):
        eps = self.norm_layer.eps
        if mask:
            ids_keep, mask_t, ids_restore = self.random_masking(x, noise)
            t = self.pos_drop(self.patch_embed.tokens(x, self.pos_embed, ids_keep))
            emb_l1 = emb_l2 = None
            for i, blk in enumerate(self.vit_block):
                t = blk(t)
                if i == 1:
                    emb_l1 = t
                elif i == 3:
                    emb_l2 = t
            if self.args.use_feature_fusion:
                emb_lh = ops.LayerNormFn.apply(emb_l1, emb_l2, t, self.norm_layer.weight, self.norm_layer.bias, eps)
            else:
                emb_lh = ops.LayerNormFn.apply(t, None, None, self.norm_layer.weight, self.norm_layer.bias, eps)
            return emb_l1, emb_l2, emb_lh, mask_t, ids_restore

        t = self.pos_drop(self.patch_embed.tokens(x, self.pos_embed, None))
        out_embs = []
        emb_l1 = emb_l2 = attn = None
        last = len(self.vit_block) - 1
        for i, blk in enumerate(self.vit_block):
            if i < last:
                t = blk(t)
            else:
                t, attn = blk(t, return_attn=True)
            if i == 0:
                emb_l1 = t
            elif i == 1:
                emb_l2 = t
            if i in self.out_indices:
                out_embs.append(emb2patch_frame(t))
        emb_h = ops.LayerNormFn.apply(t, None, None, self.norm_layer.weight, self.norm_layer.bias, eps)
        if self.args.phase in ("finetune_semseg", "finetune_flow"):
            return emb_l1, emb_l2, emb_h, out_embs, attn
        return emb_l1, emb_l2, emb_h, attn


def vit_small_patch16(args, **kwargs):
    return ViT(args=args, input_size=224, patch_size=16, embed_dim=384, depth=12, out_indices=[3, 5, 7, 11],
               num_heads=12, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_base_patch16(args, **kwargs):
    return ViT(args=args, input_size=224, patch_size=16, embed_dim=768, depth=12, out_indices=[3, 5, 7, 11],
               num_heads=12, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def vit_tiny_patch16_64(args, **kwargs):
    """BASELINE.json config 1 ("ViT-Tiny, 64x64 voxels"): not a reference factory (SURVEY.md header); built from the
    same class with the plumbing-test sizes the golden fixtures use."""
    return ViT(args=args, input_size=64, patch_size=16, embed_dim=192, depth=12, out_indices=[3, 5, 7, 11],
               num_heads=3, mlp_ratio=4, norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
