"""Voxel-grid view augmentation on the GPU (reference dataset/augmentation/view_augment.py:9-95: `view_crop` ->
`view_resize(mode='nearest')` -> `view_horizontal_flip` -> `evg_time_flip`), batched: one kernel launch
(evp_view_augment_f32) for B grids that K1 left in HBM.

The reference draws its random decisions from the process-global legacy numpy stream inside each DataLoader worker.
Here they are explicit, host-drawn int32 rows {x0, y0, w, h, hflip, tflip}:
  * `draw_evg_params(rs, ...)`  -- the reference's exact call order on a `numpy.random.RandomState`, so
    `RandomState(seed)` reproduces `evg_augment(..., seed=seed)` bit for bit (pinned by tests/golden/evg_augment.npz);
  * `draw_evg_params_batch(seed, step, B, ...)` -- a counter-based stream (Philox keyed by (seed, step, sample)): the
    decisions of a sample do not depend on worker scheduling or on how many draws other samples consumed.
The event-level augmentations (erase / add correlated events) are in events_augment.py next to this file."""
import math

import numpy as np
import torch

from ... import _lib
from ..._lib import call, ptr, stream_ptr


def draw_evg_params(rs, H, W, crop_min=0.8, ratio=(3 / 4, 4 / 3)):
    """Decisions of one evg_augment call, in the reference's draw order (view_augment.py:14-31,41,49)."""
    x0, y0, w, h = 0, 0, W, H
    area = W * H
    for _ in range(10):
        target_area = rs.uniform(crop_min, 1.0) * area
        aspect = rs.uniform(W / H * ratio[0], W / H * ratio[1])
        cw, ch = int(round(math.sqrt(target_area * aspect))), int(round(math.sqrt(target_area / aspect)))
        if rs.randint(0, 10) < 5:
            cw, ch = ch, cw
        if cw < W and ch < H:
            x0, y0, w, h = rs.randint(0, W - cw), rs.randint(0, H - ch), cw, ch
            break
    hflip = int(rs.random_sample() < 0.5)
    tflip = int(rs.random_sample() < 0.5)
    return x0, y0, w, h, hflip, tflip


def draw_evg_params_batch(seed, step, B, H, W, crop_min=0.8, first_sample=0, ratio=(3 / 4, 4 / 3), U=None):
    """int32 [B,6] rows for samples first_sample .. first_sample+B-1 of optimizer step `step`: sample i uses the Philox
    stream keyed by (seed, step, i). Same distribution as the reference's (same accept / reject rule: up to ten tries of
    (area, aspect, swap coin), the first box that fits is placed uniformly, else the whole view; then the two flip coins), vectorised
    over the batch: every sample draws the uniforms of all ten tries, the first accepted one is used."""
    if U is None:
        U = evg_uniforms(seed, step, B, first_sample)
    T = U[:, :50].reshape(B, 10, 5)
    area = W * H
    target = (crop_min + T[..., 0] * (1.0 - crop_min)) * area
    aspect = W / H * ratio[0] + T[..., 1] * (W / H * ratio[1] - W / H * ratio[0])
    cw = np.rint(np.sqrt(target * aspect)).astype(np.int64)
    ch = np.rint(np.sqrt(target / aspect)).astype(np.int64)
    swap = (T[..., 2] * 10).astype(np.int64) < 5
    cw, ch = np.where(swap, ch, cw), np.where(swap, cw, ch)
    ok = (cw < W) & (ch < H)
    first = np.argmax(ok, axis=1)
    any_ok = ok.any(axis=1)
    r = np.arange(B)
    cw1, ch1 = cw[r, first], ch[r, first]
    x0 = np.minimum((T[r, first, 3] * np.maximum(W - cw1, 1)).astype(np.int64), np.maximum(W - cw1 - 1, 0))
    y0 = np.minimum((T[r, first, 4] * np.maximum(H - ch1, 1)).astype(np.int64), np.maximum(H - ch1 - 1, 0))
    out = np.zeros((B, 6), dtype=np.int32)
    out[:, 0], out[:, 1] = np.where(any_ok, x0, 0), np.where(any_ok, y0, 0)
    out[:, 2], out[:, 3] = np.where(any_ok, cw1, W), np.where(any_ok, ch1, H)
    out[:, 4], out[:, 5] = U[:, 50] < 0.5, U[:, 51] < 0.5
    return out


def evg_uniforms(seed, step, B, first_sample=0):
    """float64 [B, 52] in [0, 1): the uniforms behind one sample's crop box and flip coins (shared counter stream, purpose 4). The frame's
    box is drawn from the SAME uniforms scaled to the frame's size (the reference re-seeds numpy with the sample's seed before
    frame_augment), so a caller that needs both passes this array to draw_evg_params_batch twice."""
    from .events_augment import philox_words
    return philox_words(seed, step, first_sample + np.arange(B), 4, 10 * 5 + 2).astype(np.float64) * 2.3283064365386963e-10


def evg_augment_batch(voxels, params, size, negate=None, out=None):
    """voxels float32 [B,C,H,W] on the GPU, params int32 [B,6] (host array or device tensor) -> float32
    [B,C,size[0],size[1]]; negate=None applies the reference's rule (5- or 6-bin grids are negated on a time flip)."""
    _lib.require_device()
    if not voxels.is_cuda or voxels.dtype != torch.float32 or not voxels.is_contiguous():
        raise _lib.EvpError("evg_augment_batch: voxels must be a contiguous float32 tensor in device memory")
    B, C, H, W = voxels.shape
    if not torch.is_tensor(params):
        p = np.ascontiguousarray(params, dtype=np.int32).reshape(B, 6)
        if ((p[:, 0] < 0) | (p[:, 1] < 0) | (p[:, 2] < 1) | (p[:, 3] < 1) | (p[:, 0] + p[:, 2] > W) | (p[:, 1] + p[:, 3] > H)).any():
            raise ValueError("evg_augment_batch: crop box outside the view")
        params = torch.from_numpy(p).to(voxels.device, non_blocking=True)
    if params.dtype != torch.int32 or tuple(params.shape) != (B, 6) or not params.is_contiguous():
        raise ValueError("evg_augment_batch: params must be int32 [B,6]")
    Ho, Wo = int(size[0]), int(size[1])
    if out is None:
        out = torch.empty(B, C, Ho, Wo, dtype=torch.float32, device=voxels.device)
    if negate is None:
        negate = C in (5, 6)
    call("evp_view_augment_f32", ptr(voxels), ptr(params), ptr(out), B, C, H, W, Ho, Wo, int(bool(negate)), stream_ptr())
    return out


def frame_augment_batch(frames, params, size, out=None):
    """Difference-map targets of the batch (reference view_augment.py:79-89 `frame_augment`): frames float32 [B,C,H,W] on the
    GPU, params int32 [B,6] -- THE SAME rows that went to evg_augment_batch (the reference re-seeds numpy with the sample's
    seed, so the crop box and the flip coin repeat; column 5 is evg_augment's time-flip flag, which negates the frame) ->
    float32 [B,C,size[0],size[1]], bicubic (ATen upsample_bicubic2d, align_corners=False)."""
    _lib.require_device()
    if not frames.is_cuda or frames.dtype != torch.float32 or not frames.is_contiguous():
        raise _lib.EvpError("frame_augment_batch: frames must be a contiguous float32 tensor in device memory")
    B, C, H, W = frames.shape
    if not torch.is_tensor(params):
        p = np.ascontiguousarray(params, dtype=np.int32).reshape(B, 6)
        if ((p[:, 0] < 0) | (p[:, 1] < 0) | (p[:, 2] < 1) | (p[:, 3] < 1) | (p[:, 0] + p[:, 2] > W) | (p[:, 1] + p[:, 3] > H)).any():
            raise ValueError("frame_augment_batch: crop box outside the frame")
        params = torch.from_numpy(p).to(frames.device, non_blocking=True)
    if params.dtype != torch.int32 or tuple(params.shape) != (B, 6) or not params.is_contiguous():
        raise ValueError("frame_augment_batch: params must be int32 [B,6]")
    Ho, Wo = int(size[0]), int(size[1])
    if out is None:
        out = torch.empty(B, C, Ho, Wo, dtype=torch.float32, device=frames.device)
    call("evp_frame_augment_f32", ptr(frames), ptr(params), ptr(out), B, C, H, W, Ho, Wo, stream_ptr())
    return out
