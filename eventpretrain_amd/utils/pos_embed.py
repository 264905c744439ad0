"""Fixed 2-D sin-cos positional table (host-side, init only).

Same numbers as the reference's utils/pos_embed.py:40-55 (float32 numpy arithmetic; the FIRST half of the channels
encodes the column index w, the second half the row index h; each half is [sin | cos])."""
import numpy as np


def _axis_table(dim, positions):
    """[len(positions), dim] = [sin(p*w_k) | cos(p*w_k)], w_k = 10000^(-k/(dim/2)), all in float32."""
    if dim % 2:
        raise ValueError("embed dim must be even")
    freq = np.arange(dim // 2, dtype=np.float32)
    freq /= dim / 2.0
    freq = 1.0 / 10000 ** freq
    phase = np.einsum("m,d->md", positions.reshape(-1), freq)
    return np.concatenate([np.sin(phase), np.cos(phase)], axis=1)


def get_2d_sincos_pos_embed(embed_dim, grid_size, cls_token=False):
    """-> float32 [grid_size**2 (+1), embed_dim]; row index = h * grid_size + w."""
    coord = np.arange(grid_size, dtype=np.float32)
    w_of, h_of = np.meshgrid(coord, coord)          # w_of[h, w] = w ; h_of[h, w] = h
    table = np.concatenate([_axis_table(embed_dim // 2, w_of), _axis_table(embed_dim // 2, h_of)], axis=1)
    if cls_token:
        table = np.concatenate([np.zeros([1, embed_dim]), table], axis=0)
    return table
