"""Layout helpers with the reference's names (utils/reshape.py). These are pure index permutations (views + one
copy), used at API boundaries and by tests; the hot path fuses them into kernel addressing instead
(csrc/loss.hip reads the target frame in patch order, csrc/tokens.hip gathers patches)."""
import torch


def frame2emb(patch_size, frame):
    """(B,C,H,W) -> (B, L, p*p*C), inner order (py, px, c)   [reference utils/reshape.py:15-22]"""
    B, Cc, H, W = frame.shape
    p = patch_size
    t = frame.reshape(B, Cc, H // p, p, W // p, p).permute(0, 2, 4, 3, 5, 1)
    return t.reshape(B, (H // p) * (W // p), p * p * Cc)


def emb2frame(args, emb, chans):
    """(B, L, p*p*chans) -> (B, chans, H, W)   [reference utils/reshape.py:5-13]"""
    B, L, _ = emb.shape
    g = int(L ** 0.5)
    if g * g != L:
        raise ValueError("token count is not a square")
    p = args.patch_size
    t = emb.reshape(B, g, g, p, p, chans).permute(0, 5, 1, 3, 2, 4)
    return t.reshape(B, chans, g * p, g * p)


def emb2patch_frame(emb):
    """(B, L, C) -> (B, C, g, g)   [reference utils/reshape.py:24-31]"""
    B, L, Cc = emb.shape
    g = int(L ** 0.5)
    if g * g != L:
        raise ValueError("token count is not a square")
    return emb.reshape(B, g, g, Cc).permute(0, 3, 1, 2)


def patch_frame2emb(patch_frame):
    """(B, C, h, w) -> (B, h*w, C)   [reference utils/reshape.py:33-38]"""
    B, Cc, h, w = patch_frame.shape
    return patch_frame.reshape(B, Cc, h * w).permute(0, 2, 1)


def resize(input, size, scale_factor=None, mode="bilinear", align_corners=None):
    """Visualisation-only helper of the reference (utils/reshape.py:40-43); not on the training path."""
    return torch.nn.functional.interpolate(input, size, scale_factor, mode, align_corners)
