"""CPU restatement (numpy) of the voxel-grid view augmentation of the pre-training loader.

TEST INFRASTRUCTURE ONLY (see oracle/model_oracle.py). Restates dataset/augmentation/view_augment.py:9-77
(`view_crop` -> `view_resize(mode='nearest')` -> `view_horizontal_flip` -> `evg_time_flip`) split into
  (1) the random DECISIONS, drawn from a legacy numpy stream in exactly the reference's call order, and
  (2) the deterministic pixel transform given those decisions.
Pinned by tests/golden/evg_augment.npz (outputs of the reference's own evg_augment under np.random.seed)."""
import math

import numpy as np


def draw_evg_params(rs, H, W, crop_min=0.8, ratio=(3 / 4, 4 / 3)):
    """The decisions of evg_augment (view_augment.py:84-95) from a numpy RandomState `rs`:
    up to 10 crop attempts (uniform area scale, uniform aspect ratio, a coin that swaps width/height, then the two
    start offsets; view_augment.py:14-31), a horizontal-flip coin (:41) and a time-flip coin (:49).
    Returns (x0, y0, w, h, hflip, tflip); no accepted attempt = the full view."""
    x0, y0, w, h = 0, 0, W, H
    area = W * H
    for _ in range(10):
        target = rs.uniform(crop_min, 1.0) * area
        aspect = rs.uniform(W / H * ratio[0], W / H * ratio[1])
        cw = int(round(math.sqrt(target * aspect)))
        ch = int(round(math.sqrt(target / aspect)))
        if rs.randint(0, 10) < 5:
            cw, ch = ch, cw
        if cw < W and ch < H:
            x0 = rs.randint(0, W - cw)
            y0 = rs.randint(0, H - ch)
            w, h = cw, ch
            break
    hflip = int(rs.random_sample() < 0.5)
    tflip = int(rs.random_sample() < 0.5)
    return x0, y0, w, h, hflip, tflip


def nearest_index(n_out, n_in):
    """Source index of F.interpolate(mode='nearest'): min(floor(dst * float32(n_in / n_out)), n_in - 1), the scale and the
    product in float32 as ATen computes them."""
    scale = np.float32(n_in) / np.float32(n_out)
    idx = np.floor(np.arange(n_out, dtype=np.float32) * scale).astype(np.int64)
    return np.minimum(idx, n_in - 1)


def evg_transform(view, params, out_hw, negate=True):
    """view float32 [C,H,W] -> [C,Ho,Wo]: crop box, nearest resize, horizontal flip, time flip (reversed bin order and,
    for 5/6-bin polarity grids, negated; view_augment.py:47-53)."""
    x0, y0, w, h, hflip, tflip = params
    Ho, Wo = out_hw
    ys = y0 + nearest_index(Ho, h)
    xs = x0 + nearest_index(Wo, w)
    if hflip:
        xs = xs[::-1]
    out = view[:, ys][:, :, xs]
    if tflip:
        out = out[::-1]
        if negate:
            out = -out
    return np.ascontiguousarray(out, dtype=np.float32)


# ---- event-level augmentation (dataset/augmentation/events_augment.py) ------------------------------------------------
def draw_window(rs, n, fix):
    """get_random_index(is_train=True) (events_augment.py:5-20): a clip longer than `fix` events gives up one randint for the start of
    a `fix`-long window [start, start + fix); a shorter one is taken whole and draws nothing."""
    if n > fix:
        s0 = int(rs.randint(0, n - fix))
        return s0, s0 + fix
    return 0, n


def n_imagenet_sample(rs, events, fix, sensor_hw, S, bins, crop_min=0.8):
    """The events half of PretrainNImageNetDataset.__getitem__ (dataset/pretrain/pr_n_imagenet_dataset.py:82-89) on the RUNNING stream
    `rs` -- window, erase / add, rescale, voxel grid, evg_augment without a seed -- in the dataset's own draw order.
    -> (window, rows after augmentation, evg params, voxel grid float32 [bins, S, S])."""
    from .voxel_oracle import voxel_grid
    sh, sw = sensor_hw
    s0, s1 = draw_window(rs, events.shape[0], fix)
    e = events[s0:s1].copy()
    e = erase_add_apply(e, draw_erase_add(rs, e.shape[0]), (sh, sw))
    g = voxel_grid(events_reshape(e, sw, sh, S, S), bins, (S, S))
    prm = draw_evg_params(rs, S, S, crop_min)
    return (s0, s1), e.shape[0], prm, evg_transform(g, prm, (S, S), negate=bins in (5, 6))


def draw_erase_add(rs, n):
    """The decisions of erase_and_add_events (events_augment.py:31-44) from a numpy RandomState `rs`, in the reference's
    call order: erase count, erased rows (sorted), add count, three per-row normal noise columns (x: sd 1.5, y: sd 1.5,
    t: sd 0.001 -- drawn for ALL n rows, as the reference does), added rows. Returns None when the clip is too short
    (`int(0.01 n) == 0`), else (erase_index int64 [E] ascending, add_index int64 [A], add_noise float64 [A,3])."""
    if int(0.01 * n) <= 0:
        return None
    lo, hi = int(0.001 * n), int(0.01 * n)
    erase_num = rs.randint(lo, hi)
    erase_index = np.sort(rs.choice(np.arange(n), size=erase_num, replace=False))
    add_num = rs.randint(lo, hi)
    nx = rs.normal(0, 1.5, size=(n, 1))
    ny = rs.normal(0, 1.5, size=(n, 1))
    nt = rs.normal(0, 0.001, size=(n, 1))
    add_index = rs.choice(np.arange(n), size=add_num, replace=False)
    noise = np.concatenate((nx[add_index], ny[add_index], nt[add_index]), 1)
    return erase_index.astype(np.int64), add_index.astype(np.int64), np.ascontiguousarray(noise, dtype=np.float64)


def erase_add_apply(events, decisions, size):
    """The deterministic part of erase_and_add_events (events_augment.py:38-53) given the decisions: float64 [n,4]
    (x,y,t,p) -> float64 [n-E+A,4], time-sorted (stable: on equal stamps original rows first)."""
    if decisions is None:
        return events
    erase_index, add_index, noise = decisions
    sensor_h, sensor_w = size
    add = events[add_index].copy()
    add[:, :3] += noise
    add[:, 0] = np.clip(add[:, 0], 0, sensor_w - 1)
    add[:, 1] = np.clip(add[:, 1], 0, sensor_h - 1)
    kept = np.delete(events, erase_index, axis=0)
    both = np.concatenate((kept, add))
    return both[np.argsort(both[:, 2], kind="stable")]


def events_reshape(events, sensor_w, sensor_h, input_w, input_h):
    """events_augment.py:22-26: x *= input_w / sensor_w, y *= input_h / sensor_h (float64)."""
    out = events.copy()
    out[:, 0] *= (input_w / sensor_w)
    out[:, 1] *= (input_h / sensor_h)
    return out


def frame_transform(frame, params, out_hw):
    """frame float32 [C,H,W] -> [C,Ho,Wo] as view_augment.py:79-89 `frame_augment` transforms it given the decisions
    (x0, y0, w, h, hflip, tflip): crop box, F.interpolate(mode='bicubic') -- the very ATen op the reference calls, on the CPU --,
    horizontal flip, negation when the voxel grid was time-flipped."""
    import torch
    import torch.nn.functional as F
    x0, y0, w, h, hflip, tflip = params
    t = torch.from_numpy(np.ascontiguousarray(frame[:, y0:y0 + h, x0:x0 + w], dtype=np.float32)).unsqueeze(0)
    t = F.interpolate(t, (int(out_hw[0]), int(out_hw[1])), None, "bicubic", None).squeeze(0)
    if hflip:
        t = torch.flip(t, dims=[2])
    if tflip:
        t = -t
    return t.numpy()
