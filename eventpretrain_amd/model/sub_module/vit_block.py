"""ViT building blocks with the reference's class names, constructor arguments and state-dict keys
(reference model/sub_module/vit_block.py:44-68,118-143,215-254). The nn.Linear / nn.LayerNorm / nn.Conv2d members are
parameter containers only: every forward goes through the HIP kernels in libevtpretrain.so (eventpretrain_amd.ops)."""
import torch
import torch.nn as nn

from ... import ops


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class PatchEmbed(nn.Module):
    """Image to patch embedding: Conv2d(k=s=patch) -> LayerNorm over channels (eps 1e-5) -> GELU.
    forward(x) returns the reference's (B, D, H/p, W/p) layout; `tokens()` is the fused hot-path entry that also adds
    the positional table and keeps only `ids_keep`."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768):
        super().__init__()
        self.img_size = _pair(img_size)
        self.patch_size = _pair(patch_size)
        self.num_patches = (self.img_size[1] // self.patch_size[1]) * (self.img_size[0] // self.patch_size[0])
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.norm = nn.LayerNorm(embed_dim)
        self.act = nn.GELU()

    def tokens(self, x, pos_embed, ids_keep=None):
        """(B,C,H,W) f32 -> (B, n_keep, D) f32 = GELU(LN(conv(x))) + pos_embed, restricted to ids_keep."""
        B, Cc, H, W = x.shape
        if (H, W) != self.img_size:
            raise AssertionError(f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]}).")
        return ops.PatchEmbedFn.apply(x, ids_keep, self.proj.weight, self.proj.bias, self.norm.weight, self.norm.bias,
                                      pos_embed, self.patch_size[0])

    def forward(self, x):
        zero_pos = torch.zeros(1, self.num_patches, self.proj.weight.shape[0], device=x.device)
        t = self.tokens(x, zero_pos)
        B, L, D = t.shape
        return t.transpose(1, 2).reshape(B, D, self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1])


class Attention(nn.Module):
    """Parameter holder for the fused qkv / proj Linears; the math runs inside ViTBlock's fused function."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        self.attn_drop_rate = float(attn_drop)      # vit_block.py:127,138: applied by the block's fused function through the
        self.proj_drop_rate = float(proj_drop)      # materialised-probabilities path (ops.attention_dropout_fwd); ops.BlockDrop
        if qk_scale is not None:
            raise NotImplementedError("qk_scale override is not used on the pre-training path")
        if not qkv_bias:
            raise NotImplementedError("qkv_bias=False is not used on the pre-training path")
        self.num_heads = num_heads
        self.scale = (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        self.drop_rate = float(drop)                # applied by the block's fused function (ops.BlockDrop)
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)


class ViTBlock(nn.Module):
    """x = x + Attn(LN(x)); x = x + Mlp(LN(x))  -- one fused autograd node (ops.ViTBlockFn)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_scale=None, drop=0., attn_drop=0.,
                 drop_path=0., act_layer=nn.GELU, norm_layer=nn.LayerNorm):
        super().__init__()
        # stochastic depth (vit_block.py:241,252-253) and proj / Mlp dropout: training mode only, both branches; with both rates 0
        # -- the pre-training recipe -- the block takes the fused path with the residual add inside the GEMM epilogue
        self.drop_path_rate = float(drop_path)
        self.drop_rate = float(drop)
        self.attn_drop_rate = float(attn_drop)      # dropout on the attention probabilities (vit_block.py:127,138)
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=drop)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), drop=drop)
        if self.norm1.eps != self.norm2.eps:
            raise ValueError("norm1/norm2 eps differ")

    def forward(self, x, return_attn=False, block_drop=None):
        """`block_drop` (ops.BlockDrop): explicit draws / masks for this call (tests); default: drawn here in training mode."""
        rd = block_drop if block_drop is not None else ops.draw_block_drop(self, x.shape[0], x.device)
        return ops.vit_block(x, self, self.attn.num_heads, self.norm1.eps, want_attn=return_attn, rd=rd)
