"""ConvMAE-style convolution block (reference model/sub_module/conv_block.py:6-51): same classes, constructor arguments
and state-dict keys; the nn.Conv2d / nn.LayerNorm members only hold parameters, the math runs in ops.ConvBlockFn on
channels-last token maps."""
import torch.nn as nn


class CMlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        self.drop_rate = float(drop)
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Conv2d(in_features, hidden_features, 1)
        self.fc2 = nn.Conv2d(hidden_features, out_features or in_features, 1)


class ConvBlock(nn.Module):
    def __init__(self, input_size, kernel_size, mlp_ratio=4., drop=0., drop_path=0., act_layer=nn.GELU):
        super().__init__()
        self.drop_path_rate = float(drop_path)      # conv_block.py:35,43-49: both branches, training mode only
        self.drop_rate = float(drop)                # CMlp.drop (conv_block.py:19-21)
        if kernel_size != 5:
            raise NotImplementedError("the depthwise kernel is built for kernel_size=5 (the only size the reference uses)")
        self.norm1 = nn.LayerNorm(input_size)
        self.conv1 = nn.Conv2d(input_size, input_size, 1)
        self.attn = nn.Conv2d(input_size, input_size, kernel_size=kernel_size, padding=kernel_size // 2, groups=input_size)
        self.conv2 = nn.Conv2d(input_size, input_size, 1)
        self.norm2 = nn.LayerNorm(input_size)
        self.mlp = CMlp(in_features=input_size, hidden_features=int(input_size * mlp_ratio), drop=drop)

    def forward_tokens(self, x, H, W, mask=None, mask_scale=1, block_drop=None):
        """x (B, H*W, C) f32 channels-last; mask (B, L) with 1 = removed at the coarse 14x14 grid (mask_scale = how
        many map positions one coarse cell spans per side)."""
        from ... import ops
        rd = block_drop if block_drop is not None else ops.draw_block_drop(self, x.shape[0], x.device)
        return ops.conv_block(x, self, H, W, mask, mask_scale, rd=rd)

    def forward(self, x, mask=None):
        """Reference layout: x (B, C, H, W); mask (B, 1, H, W) keep factors (1 = keep) or None."""
        B, C_, H, W = x.shape
        t = x.permute(0, 2, 3, 1).reshape(B, H * W, C_).contiguous()
        coarse = None
        if mask is not None:
            # reference conv_block.py:41-44 multiplies conv1's output by this map; the kernel's mask argument is "1 = removed"
            # at mask_scale x mask_scale cells, so a dense keep map k enters as 1 - k at scale 1 (exact for 0 / 1 maps)
            if tuple(mask.shape) != (B, 1, H, W):
                raise ValueError("ConvBlock.forward: mask must be (B, 1, H, W)")
            coarse = (1.0 - mask.float()).reshape(B, H * W).contiguous()
        return self.forward_tokens(t, H, W, coarse, 1).reshape(B, H, W, C_).permute(0, 3, 1, 2)
