"""MoCo-v3 projection / prediction heads (reference model/sub_module/mlp_head.py:4-24): an nn.Sequential of
[Linear(no bias), BatchNorm2d, ReLU] x (n-1), Linear(no bias), BatchNorm2d(affine=False) -- kept as the parameter /
buffer container (same state-dict keys); `run_mlp_2d` executes it on (B, L, C) tokens with the HIP kernels, applying
each BatchNorm2d over the B*L rows exactly as the (B,C,h,w) round trip of pr_hub_model.py:223-237 does."""
import torch.nn as nn

from ... import ops


def _build_mlp_2d(num_layers, input_dim, mlp_dim, output_dim, last_bn=True, seq=True):
    layers = []
    for l in range(num_layers):
        d_in = input_dim if l == 0 else mlp_dim
        d_out = output_dim if l == num_layers - 1 else mlp_dim
        layers.append(nn.Linear(d_in, d_out, bias=False))
        if l < num_layers - 1:
            layers += [nn.BatchNorm2d(d_out), nn.ReLU(inplace=True)]
        elif last_bn:
            layers.append(nn.BatchNorm2d(d_out, affine=False))
    return nn.Sequential(*layers) if seq else layers


def run_mlp_2d(seq, x):
    mods = list(seq)
    i = 0
    while i < len(mods):
        lin = mods[i]
        bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm2d) else None
        relu = bn is not None and i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
        x = ops.linear_bn_tokens(x, lin, bn, relu)
        i += 1 + (bn is not None) + relu
    return x
