"""The pre-training loader's per-sample chain as ONE batched call on clips already resident in HBM:

    get_random_index -> events_augment -> events_reshape -> events_to_voxel_grid -> evg_augment  (+ frame_augment for the target)

(reference dataset/pretrain/pr_n_imagenet_dataset.py:76-107; the seeded evg / frame pair as pr_ef_imagenet_dataset.py:187-206).
The reference runs it per sample in DataLoader workers at 10-27 ms per clip on a CPU core, which cannot feed a
5-12 k samples/s step; here the random DECISIONS are drawn on the host (a few hundred numbers per clip, overlappable with the
previous step) and every byte of event / voxel data stays on the device:

    window pick + erase / add  csrc/events.hip   evp_events_erase_add_win_f64  (2 launches: sort the <= 1 % added rows in LDS; merge)
    rescale + voxel scatter    csrc/voxel.hip    evp_voxel_scatter_scaled_f32  (K1, 3 launches in the verified mode)
    crop / resize / flips      csrc/augment.hip  evp_view_augment_f32, evp_frame_augment_f32

Decision streams: "counter" (default) = Philox keyed by (seed, step, sample): a sample's augmentation does not depend on worker
scheduling; "legacy" = the reference's process-global numpy stream in the reference's exact call order, so np.random.seed(s)
reproduces the reference's output for that sample (pinned by tests/golden/loader_chain.npz)."""
import numpy as np
import torch

from ... import _lib
from ..augmentation import events_augment as ea
from ..augmentation import view_augment as va
from ..dataset_utils.events_to_voxel_grid import voxel_grid_batch


class GpuInputPipeline:
    def __init__(self, args, seed=0, decision_stream="counter"):
        """args: the reference's namespace (fix_events_num, img_sensor_h / _w, input_size, num_bins, crop_min)."""
        if decision_stream not in ("counter", "legacy"):
            raise ValueError("decision_stream must be 'counter' or 'legacy'")
        self.args, self.seed, self.stream = args, int(seed), decision_stream
        self.sensor = (int(args.img_sensor_h), int(args.img_sensor_w))
        self.S = int(args.input_size)
        self.bins = int(args.num_bins)
        self.crop_min = float(getattr(args, "crop_min", 0.8))

    # ------------------------------------------------------------------------------------------------ host: decisions
    def draw(self, sizes, step, first_sample=0, sample_seeds=None, frame_size=None):
        """-> (windows int64 [B,2], erase/add decisions per clip, evg params int32 [B,6]) for clips of `sizes` rows; with
        `frame_size` = (Hf, Wf) of the target frames a fourth item, the frames' own params: the reference re-seeds numpy with the
        sample's seed before frame_augment (pr_ef_imagenet_dataset.py:194-206), so the same uniform draws are scaled to the FRAME's
        size -- the same box only when frame and grid have the same size -- and the time-flip flag is evg_augment's."""
        B = len(sizes)
        fix = int(self.args.fix_events_num)
        windows = np.zeros((B, 2), np.int64)
        params = np.zeros((B, 6), np.int32)
        fparams = np.zeros((B, 6), np.int32) if frame_size is not None else None
        H, W = self.sensor
        if self.stream == "legacy":
            # one sample after the other, the reference's call order (pr_n_imagenet_dataset.py:83-89 on the running stream; the
            # evg / frame pair re-seeded with np.random.randint(1000) as pr_ef_imagenet_dataset.py:187-195)
            dec = []
            for i, n in enumerate(int(v) for v in sizes):
                if sample_seeds is not None:
                    np.random.seed(int(sample_seeds[i]))
                if n > fix:
                    s0 = np.random.randint(0, n - fix)
                    windows[i] = (s0, s0 + fix)
                else:
                    windows[i] = (0, n)
                dec.append(ea.draw_erase_add(int(windows[i, 1] - windows[i, 0])))
                seed2 = np.random.randint(1000)
                np.random.seed(seed2)                           # evg_augment(..., seed=seed) re-seeds the global stream (view_augment.py:66-67)
                params[i] = va.draw_evg_params(np.random, self.S, self.S, self.crop_min)
                if fparams is not None:
                    np.random.seed(seed2)                       # frame_augment(..., seed=seed) does so again (view_augment.py:80-81)
                    fparams[i] = va.draw_evg_params(np.random, int(frame_size[0]), int(frame_size[1]), self.crop_min)
                    fparams[i, 5] = params[i, 5]
            return (windows, dec, params) if fparams is None else (windows, dec, params, fparams)
        for i, n in enumerate(int(v) for v in sizes):
            if n > fix:
                g = np.random.Generator(np.random.Philox(key=[self.seed & (2 ** 64 - 1), (((int(step) << 24) ^ (first_sample + i)) + (1 << 60)) & (2 ** 64 - 1)]))
                s0 = int(g.integers(0, n - fix))
                windows[i] = (s0, s0 + fix)
            else:
                windows[i] = (0, n)
        dec = ea.draw_erase_add_batch(self.seed, step, windows[:, 1] - windows[:, 0], first_sample)
        params = va.draw_evg_params_batch(self.seed, step, B, self.S, self.S, self.crop_min, first_sample)
        if fparams is None:
            return windows, dec, params
        fparams = va.draw_evg_params_batch(self.seed, step, B, int(frame_size[0]), int(frame_size[1]), self.crop_min, first_sample)
        fparams[:, 5] = params[:, 5]
        return windows, dec, params, fparams

    # ------------------------------------------------------------------------------------------------ device: data
    def run(self, events, clip_offsets, windows, decisions, params, frames=None, frame_params=None, assume_sorted=True):
        """events float64 CUDA [n_total,4] (x,y,t,p) sensor coordinates, clips time-sorted; clip_offsets int64 host [B+1].
        `frame_params`: the frames' own rows (see draw); default = `params` (frames of the grid's size).
        Returns (voxels float32 [B,bins,S,S], targets float32 [B,C,S,S] | None)."""
        _lib.require_device()
        H, W = self.sensor
        ev, off = ea.events_augment_batch(events, clip_offsets, decisions, (H, W), windows=windows)
        vox = voxel_grid_batch(ev, off, self.bins, (self.S, self.S), assume_sorted=assume_sorted, scale=(self.S / W, self.S / H))
        # the grid's and the frames' parameter rows in ONE upload (each pageable copy is a stall of the host)
        both = frames is not None and frame_params is not None and not torch.is_tensor(frame_params)
        rows = np.ascontiguousarray(params, dtype=np.int32).reshape(-1, 6)
        if both:
            fr = np.ascontiguousarray(frame_params, dtype=np.int32).reshape(-1, 6)
            B, Hf, Wf = rows.shape[0], int(frames.shape[-2]), int(frames.shape[-1])
            if ((fr[:, 0] < 0) | (fr[:, 1] < 0) | (fr[:, 2] < 1) | (fr[:, 3] < 1) | (fr[:, 0] + fr[:, 2] > Wf) | (fr[:, 1] + fr[:, 3] > Hf)).any():
                raise ValueError("GpuInputPipeline.run: frame crop box outside the frame")
            rows = np.concatenate([rows, fr], 0)
        p_all = torch.from_numpy(rows).to(events.device, non_blocking=True)
        p_dev = p_all[:len(params)]
        out = va.evg_augment_batch(vox, p_dev, (self.S, self.S))
        tgt = None
        if frames is not None:
            fp = p_all[len(params):] if both else (p_dev if frame_params is None else frame_params)
            tgt = va.frame_augment_batch(frames, fp, (self.S, self.S))
        return out, tgt

    def batch(self, events, clip_offsets, step, frames=None, first_sample=0, sample_seeds=None):
        """The whole chain for one batch: decisions on the host, data on the device."""
        offs = np.asarray(clip_offsets.cpu() if torch.is_tensor(clip_offsets) else clip_offsets, dtype=np.int64)
        if frames is None:
            windows, dec, params = self.draw(offs[1:] - offs[:-1], step, first_sample, sample_seeds)
            return self.run(events, offs, windows, dec, params)
        windows, dec, params, fparams = self.draw(offs[1:] - offs[:-1], step, first_sample, sample_seeds, frame_size=frames.shape[-2:])
        return self.run(events, offs, windows, dec, params, frames, fparams)

    def algorithmic_bytes(self, sizes):
        """HBM bytes the chain has to move for clips whose picked windows hold `sizes` rows (the figure bench.py prices the chain
        with): read the window (32 B / row) + write the augmented clip (32 B / row; +- 1 %) + K1 reads it again and writes the grid
        once + the view augmentation reads the grid and writes the view."""
        n = float(np.sum(sizes))
        grid = self.bins * self.S * self.S * 4.0 * len(sizes)
        return n * 32 + n * 32 + n * 32 + grid + grid + grid
