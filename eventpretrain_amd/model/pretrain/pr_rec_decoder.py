"""Temporal-difference-map decoder (reference model/pretrain/pr_rec_decoder.py:10-95): same constructor, factories
and state-dict keys; forward runs on the HIP kernels."""
from functools import partial

import torch
import torch.nn as nn

from ... import ops
from ...utils.pos_embed import get_2d_sincos_pos_embed
from ..backbone.vit import init_linear_and_norm
from ..sub_module.vit_block import ViTBlock


class PrRecDecoder(nn.Module):
    def __init__(self, patch_size=16, num_patches=196, encoder_embed_dim=768, embed_dim=512, depth=8, num_heads=16,
                 mlp_ratio=4., norm_layer=nn.LayerNorm, frame_chans=1):
        super().__init__()
        self.patch_size = patch_size
        self.num_patches = num_patches
        self.patch_embed = nn.Linear(encoder_embed_dim[-1], embed_dim, bias=True)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches, embed_dim), requires_grad=False)
        self.vit_block = nn.ModuleList([
            ViTBlock(embed_dim, num_heads, mlp_ratio[0], qkv_bias=True, qk_scale=None, norm_layer=norm_layer)
            for _ in range(depth)])
        self.norm = norm_layer(embed_dim)
        self.pred = nn.Linear(embed_dim, self.patch_size ** 2 * frame_chans, bias=True)
        self.initialize_weights()

    def initialize_weights(self):
        table = get_2d_sincos_pos_embed(self.pos_embed.shape[-1], int(self.num_patches ** .5), cls_token=False)
        self.pos_embed.data.copy_(torch.from_numpy(table).float().unsqueeze(0))
        self.apply(init_linear_and_norm)

    def forward(self, x, ids_restore=None):
        """x (B, n_keep, D_enc) f32 -> (B, L, p*p*frame_chans) f32."""
        t = ops.LinearFn.apply(x, self.patch_embed.weight, self.patch_embed.bias)
        if ids_restore is None:      # nothing was masked: identity order, only the positional table is added
            ids_restore = torch.arange(t.shape[1], device=t.device).unsqueeze(0).expand(t.shape[0], -1)
        t = ops.UnshuffleFn.apply(t, self.mask_token, self.pos_embed, ids_restore.contiguous())
        for blk in self.vit_block:
            t = blk(t)
        t = ops.LayerNormFn.apply(t, None, None, self.norm.weight, self.norm.bias, self.norm.eps)
        return ops.LinearFn.apply(t, self.pred.weight, self.pred.bias)


def pretrain_rec_decoder_small_patch16(**kwargs):
    return PrRecDecoder(patch_size=16, num_patches=196, encoder_embed_dim=[128, 256, 384], embed_dim=256, depth=8,
                        num_heads=8, mlp_ratio=[4, 4, 4], norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def pretrain_rec_decoder_swin_tiny_patch32(**kwargs):
    return PrRecDecoder(patch_size=32, num_patches=49, encoder_embed_dim=[96, 192, 384, 768], embed_dim=256, depth=8,
                        num_heads=8, mlp_ratio=[4, 4, 4], norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def pretrain_rec_decoder_swin_base_patch32(**kwargs):
    """Decoder for the Swin-Base backbone (BASELINE.json config 5). The reference ships only the Swin-T pair
    (pr_rec_decoder.py:81-87); this is the same class at the base decoder's width (pr_rec_decoder.py:89-95: 512 wide,
    16 heads) on the 49 cells of 32x32 pixels of the Swin hub."""
    return PrRecDecoder(patch_size=32, num_patches=49, encoder_embed_dim=[128, 256, 512, 1024], embed_dim=512, depth=8,
                        num_heads=16, mlp_ratio=[4, 4, 4], norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def pretrain_rec_decoder_base_patch16(**kwargs):
    return PrRecDecoder(patch_size=16, num_patches=196, encoder_embed_dim=[256, 384, 768], embed_dim=512, depth=8,
                        num_heads=16, mlp_ratio=[4, 4, 4], norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def pretrain_rec_decoder_tiny_patch16_64(**kwargs):
    """Decoder of the BASELINE.json config-1 plumbing model (64x64 input -> 16 patches)."""
    return PrRecDecoder(patch_size=16, num_patches=16, encoder_embed_dim=[192], embed_dim=128, depth=4, num_heads=4,
                        mlp_ratio=[4, 4, 4], norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
