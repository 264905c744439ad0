"""ConvViT backbone (ConvMAE-style 3-stage encoder; reference model/backbone/convvit.py:12-224): same constructor,
factories, return tuples and state-dict keys. Stage 1/2 feature maps live channels-last as token maps; every Conv2d
with kernel = stride is a patch-gather + MFMA GEMM, the depthwise 5x5 is its own kernel (csrc/conv.hip)."""
from functools import partial

import torch
import torch.nn as nn

from ... import ops
from ...utils.pos_embed import get_2d_sincos_pos_embed
from ..sub_module.conv_block import ConvBlock
from ..sub_module.vit_block import PatchEmbed, ViTBlock
from .vit import init_linear_and_norm


def _nchw(tokens, H, W):
    B, _, C_ = tokens.shape
    return tokens.view(B, H, W, C_).permute(0, 3, 1, 2)


class ConvViT(nn.Module):
    def __init__(self, args, input_size=224, patch_size=16, embed_dim=1024, depth=24, num_heads=16, mlp_ratio=4.,
                 norm_layer=nn.LayerNorm, num_bins=5, mask_ratio=0., drop_rate=0., attn_drop_rate=0., drop_path_rate=0.):
        super().__init__()
        self.drop_rate = float(drop_rate)          # pos_drop (convvit.py:30,132,176)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depth))]     # stochastic depth decay rule (convvit.py:32)
        self.args = args
        self.patch_size = patch_size
        self.sizes = list(input_size)
        self.patch_embed1 = PatchEmbed(img_size=input_size[0], patch_size=patch_size[0], in_chans=num_bins, embed_dim=embed_dim[0])
        self.patch_embed2 = PatchEmbed(img_size=input_size[1], patch_size=patch_size[1], in_chans=embed_dim[0], embed_dim=embed_dim[1])
        self.patch_embed3 = PatchEmbed(img_size=input_size[2], patch_size=patch_size[2], in_chans=embed_dim[1], embed_dim=embed_dim[2])
        self.patch_embed4 = nn.Linear(embed_dim[2], embed_dim[2])
        self.num_patches = self.patch_embed3.num_patches
        self.pos_embed = nn.Parameter(torch.zeros(1, self.num_patches, embed_dim[2]), requires_grad=False)
        self.conv_block1 = nn.ModuleList([ConvBlock(input_size=embed_dim[0], kernel_size=5, mlp_ratio=4., drop=drop_rate, drop_path=dpr[i])
                                          for i in range(depth[0])])
        # the reference sizes the second stage with depth[0] as well (convvit.py:36-38)
        self.conv_block2 = nn.ModuleList([ConvBlock(input_size=embed_dim[1], kernel_size=5, mlp_ratio=4., drop=drop_rate, drop_path=dpr[depth[0] + i])
                                          for i in range(depth[0])])
        self.vit_block = nn.ModuleList([
            ViTBlock(dim=embed_dim[2], num_heads=num_heads, mlp_ratio=mlp_ratio[2], qkv_bias=True, qk_scale=None, drop=drop_rate,
                     attn_drop=attn_drop_rate, drop_path=dpr[depth[0] + depth[1] + i], norm_layer=norm_layer)
            for i in range(depth[2])])
        if args.phase == "pretrain" and args.pr_phase in ("rec", "rec+con", "rec-n"):
            self.mask_ratio = mask_ratio
            self.stage1_output_decode = nn.Conv2d(embed_dim[0], embed_dim[2], 4, stride=4)
            self.stage2_output_decode = nn.Conv2d(embed_dim[1], embed_dim[2], 2, stride=2)
        self.norm_layer = norm_layer(embed_dim[-1])
        if args.phase in ("finetune_semseg", "finetune_flow"):
            raise NotImplementedError("dense-prediction fine-tuning heads are out of scope (SURVEY.md section 2, rows 18-21)")
        self.initialize_weights()

    def initialize_weights(self):
        table = get_2d_sincos_pos_embed(self.pos_embed.shape[-1], int(self.num_patches ** .5), cls_token=False)
        self.pos_embed.data.copy_(torch.from_numpy(table).float().unsqueeze(0))
        w = self.patch_embed3.proj.weight.data
        nn.init.xavier_uniform_(w.view(w.shape[0], -1))
        self.apply(init_linear_and_norm)

    # ------------------------------------------------------------------------------------------------ masking
    def masking_noise(self, x):
        strategy = self.args.masking_strategy
        if strategy == "random":
            return torch.rand(x.shape[0], self.num_patches, device=x.device)
        if strategy in ("density", "anti-density"):
            # the reference passes the *list* patch_size to AvgPool2d here (convvit.py:101); the pooling that yields the
            # 14x14 grid the rest of the code assumes is the product of the three stage strides
            p = int(self.sizes[0] // int(self.num_patches ** .5))
            return ops.density_noise(x.detach(), p, 1.0 if strategy == "density" else -1.0)
        raise ValueError(strategy)

    def random_masking(self, x, noise=None):
        if noise is None:
            noise = self.masking_noise(x)
        return ops.mask_from_noise(noise.contiguous().float(), self.mask_ratio)

    # ------------------------------------------------------------------------------------------------ forward
    def _stages(self, x, mask_t, ids_keep):
        s1, s2, s3 = self.sizes[1], self.sizes[2], int(self.num_patches ** .5)      # 56, 28, 14
        pe1, pe2, pe3 = self.patch_embed1, self.patch_embed2, self.patch_embed3
        t1 = ops.PatchEmbedFn.apply(x, None, pe1.proj.weight, pe1.proj.bias, pe1.norm.weight, pe1.norm.bias, None, pe1.patch_size[0])
        for blk in self.conv_block1:
            t1 = blk.forward_tokens(t1, s1, s1, mask_t, s1 // s3)
        t2 = ops.PatchEmbedNHWCFn.apply(t1, None, pe2.proj.weight, pe2.proj.bias, pe2.norm.weight, pe2.norm.bias, None,
                                        pe2.patch_size[0], s1, s1)
        for blk in self.conv_block2:
            t2 = blk.forward_tokens(t2, s2, s2, mask_t, s2 // s3)
        t3 = ops.PatchEmbedNHWCFn.apply(t2, ids_keep, pe3.proj.weight, pe3.proj.bias, pe3.norm.weight, pe3.norm.bias, None,
                                        pe3.patch_size[0], s2, s2)
        t3 = ops.LinearFn.apply(t3, self.patch_embed4.weight, self.patch_embed4.bias)
        t3 = ops.AddPosGatherFn.apply(t3, self.pos_embed, ids_keep)
        if self.training and self.drop_rate > 0:          # pos_drop (convvit.py:132,176)
            t3 = ops.DropoutFn.apply(t3, self.drop_rate, ops.draw_drop_seed(t3.device))
        return t1, t2, t3, (s1, s2)

    def forward(self, x, mask=False, noise=None):
        eps = self.norm_layer.eps
        if mask:
            ids_keep, mask_t, ids_restore = self.random_masking(x, noise)
            t1, t2, t3, (s1, s2) = self._stages(x, mask_t, ids_keep)
            emb_stage1 = ops.StridedConvTokensFn.apply(t1, ids_keep, self.stage1_output_decode.weight, self.stage1_output_decode.bias, 4, s1, s1)
            emb_stage2 = ops.StridedConvTokensFn.apply(t2, ids_keep, self.stage2_output_decode.weight, self.stage2_output_decode.bias, 2, s2, s2)
            for blk in self.vit_block:
                t3 = blk(t3)
            if self.args.use_feature_fusion:
                emb_lh = ops.LayerNormFn.apply(emb_stage1, emb_stage2, t3, self.norm_layer.weight, self.norm_layer.bias, eps)
            else:
                emb_lh = ops.LayerNormFn.apply(t3, None, None, self.norm_layer.weight, self.norm_layer.bias, eps)
            return _nchw(t1, s1, s1), _nchw(t2, s2, s2), emb_lh, mask_t, ids_restore

        t1, t2, t3, (s1, s2) = self._stages(x, None, None)
        attn = None
        last = len(self.vit_block) - 1
        for i, blk in enumerate(self.vit_block):
            if i < last:
                t3 = blk(t3)
            else:
                t3, attn = blk(t3, return_attn=True)
        emb_h = ops.LayerNormFn.apply(t3, None, None, self.norm_layer.weight, self.norm_layer.bias, eps)
        return _nchw(t1, s1, s1), _nchw(t2, s2, s2), emb_h, attn


def convvit_small_patch16(args, **kwargs):
    return ConvViT(args=args, input_size=[224, 56, 28], patch_size=[4, 2, 2], embed_dim=[128, 256, 384], depth=[2, 2, 11],
                   num_heads=12, mlp_ratio=[4, 4, 4], norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)


def convvit_base_patch16(args, **kwargs):
    return ConvViT(args=args, input_size=[224, 56, 28], patch_size=[4, 2, 2], embed_dim=[256, 384, 768], depth=[2, 2, 11],
                   num_heads=12, mlp_ratio=[4, 4, 4], norm_layer=partial(nn.LayerNorm, eps=1e-6), **kwargs)
