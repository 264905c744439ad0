"""Event stream -> voxel grid on the GPU (csrc/voxel.hip), behind the reference's function name and arguments
(reference dataset/dataset_utils/events_to_voxel_grid.py:4-61), plus a batched device-resident entry point that the
input pipeline / bench uses (one launch for a whole batch of clips)."""
import numpy as np
import torch

from ... import _lib
from ..._lib import call, ptr, stream_ptr


def voxel_grid_batch(events, clip_offsets, num_bins, size, is_txyp=False, assume_sorted=True, algo=0, tile_rows=0, out=None,
                     scale=None):
    """events: float64 CUDA tensor [n_total,4]; clip_offsets: int64 CUDA tensor [n_clips+1] -> float32
    [n_clips, num_bins, H, W]. assume_sorted=True (default): the time-slab fast path, VERIFIED per clip on the device
    (evp_voxel_scatter_f32 assume_sorted = 2): a clip whose rows are not where sorted stamps would put them is redone by a
    full scan, so the result is the reference's for any row order without a host read-back. assume_sorted="trust": the
    unchecked fast path; assume_sorted=False: full scan for every clip.
    scale=(sx, sy): the loader's sensor -> input rescale (reference events_augment.py:22-26, sx = input_w / sensor_w,
    sy = input_h / sensor_h) applied to x and y inside the kernel, bit-identical to rescaling the array first."""
    _lib.require_device()
    if events.dtype != torch.float64 or events.dim() != 2 or events.shape[1] != 4 or not events.is_contiguous():
        raise _lib.EvpError("events must be a contiguous float64 [N,4] tensor")
    if clip_offsets.dtype != torch.int64:
        raise _lib.EvpError("clip_offsets must be int64")
    H, W = int(size[0]), int(size[1])
    n_clips = clip_offsets.numel() - 1
    dev = events.device
    if out is None:
        out = torch.empty(n_clips, num_bins, H, W, dtype=torch.float32, device=dev)
    n_total = int(events.shape[0])
    if n_total == 0:                    # every clip empty: the reference returns zero grids (no event touches any bin)
        return out.zero_()
    if assume_sorted == "trust":
        mode = 1
    else:
        mode = (2 if algo in (0, 3) else 1) if assume_sorted else 0
    # cuts [n_clips][bins + 2] int64, then the larger of the decode-once records (algo 2) and the verified mode's per-clip int32
    # flags -- with many empty / tiny clips (n_total << n_clips) the flags are the larger part
    ws = torch.empty(n_clips * (num_bins + 2) + max((3 * n_total + 1) // 2 + 2, (n_clips + 1) // 2), dtype=torch.int64, device=dev)
    if scale is None:
        call("evp_voxel_scatter_f32", ptr(events), ptr(clip_offsets), n_clips, n_total, int(num_bins), H, W, int(bool(is_txyp)),
             mode, int(algo), int(tile_rows), ptr(ws), ptr(out), stream_ptr())
    else:
        call("evp_voxel_scatter_scaled_f32", ptr(events), ptr(clip_offsets), n_clips, n_total, int(num_bins), H, W,
             int(bool(is_txyp)), mode, int(algo), int(tile_rows), float(scale[0]), float(scale[1]), ptr(ws),
             ptr(out), stream_ptr())
    return out


def events_to_voxel_grid(args, events, size, is_txyp=False):
    """Drop-in for the reference function: events is a numpy float64 [N,4] array (x,y,t,p) or (t,x,y,p); returns a
    float32 [num_bins,H,W] tensor -- in device memory (the next consumer is the GPU model). Sortedness of the
    stamps is checked on the host array so that unsorted input still gives the reference's result."""
    _lib.require_device()
    ev = np.ascontiguousarray(events, dtype=np.float64)
    if ev.ndim != 2 or ev.shape[1] != 4:
        raise AssertionError("events must be [N,4]")
    t = ev[:, 0 if is_txyp else 2]
    is_sorted = bool(np.all(t[1:] >= t[:-1])) if ev.shape[0] > 1 else True
    dev = torch.device("cuda", torch.cuda.current_device())
    ev_d = torch.from_numpy(ev).to(dev)
    off = torch.tensor([0, ev.shape[0]], dtype=torch.int64, device=dev)
    return voxel_grid_batch(ev_d, off, args.num_bins, size, is_txyp=is_txyp, assume_sorted="trust" if is_sorted else False)[0]
