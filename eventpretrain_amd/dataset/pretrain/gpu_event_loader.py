"""A loader for the pre-training epoch loops that starts from RAW EVENT CLIPS: what the reference's Dataset.__getitem__ does per sample on
a DataLoader worker (dataset/pretrain/pr_n_imagenet_dataset.py:76-107: window pick -> erase / add -> rescale -> voxel grid -> view
augmentation; the seeded frame target of pr_ef_imagenet_dataset.py:187-206) happens here per BATCH on the GPU, by one replay of the
self-driven loader chain (gpu_input_pipeline.CapturedChain). The workers are left with what only they can do -- reading and decoding
the clip files.

    samples = iterable of (events float64 [n,4] (x,y,t,p) time-sorted numpy array, frame float32 [C,Hf,Wf] array / tensor or None, name)
    loader  = GpuEventLoader(args, samples, batch_size=64, n_batches=len(dataset) // 64, seed=args.seed, first_sample=rank * 64)
    pr_rec_one_epoch(args, model, loader, optimizer, epoch, loss_scaler)          # yields the dict batches the trainers take

Per batch: the clips are packed into a pinned slot on a worker thread (one batch ahead), uploaded on a copy stream while the previous
training step runs, and turned into (events_voxel_grid [B,bins,S,S], sub_frame [B,C,S,S]) by one graph replay. The decisions come
from the counter stream keyed by (seed, step, first_sample + i): reproducible per sample, independent of worker scheduling."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from ... import _lib
from .gpu_input_pipeline import GpuInputPipeline


class GpuEventLoader:
    yields_device_batches = True       # static device tensors, overwritten by the next batch: the epoch loop must not read one batch ahead

    def __init__(self, args, samples, batch_size, n_batches, seed=0, first_sample=0, max_events_per_clip=None, frame_shape=None,
                 frame_key="sub_frame", step0=0):
        """`samples`: a re-iterable (one pass per epoch) of (events, frame, name); `n_batches`: batches per epoch (a short last batch is
        dropped, as the reference's training loader does: drop_last=True). `max_events_per_clip`: capacity per clip of the device
        buffer (longer clips are cut to their FIRST rows before the window pick; default 2 x fix_events_num). `frame_shape` = (C,Hf,Wf)
        when the samples carry frame targets. `step0`: the counter stream's step of the first batch (continues across epochs)."""
        _lib.require_device()
        self.args, self.samples, self.B, self.n_batches = args, samples, int(batch_size), int(n_batches)
        self.dev = torch.device(args.device)
        self.cap = int(max_events_per_clip or 2 * int(args.fix_events_num))
        self.frame_key, self.frame_shape = frame_key, None if frame_shape is None else tuple(int(v) for v in frame_shape)
        self.pipe = GpuInputPipeline(args, seed=seed)
        self.first_sample, self.step = int(first_sample), int(step0)
        B, cap = self.B, self.cap
        self.ev = torch.zeros(B * cap, 4, dtype=torch.float64, device=self.dev)
        self.frames = None if self.frame_shape is None else torch.zeros(B, *self.frame_shape, dtype=torch.float32, device=self.dev)
        # two pinned slots: one being uploaded, one being packed
        self._pin_ev = [torch.zeros(B * cap, 4, dtype=torch.float64).pin_memory() for _ in range(2)]
        self._pin_off = [torch.zeros(B + 1, dtype=torch.int64).pin_memory() for _ in range(2)]
        self._pin_fr = [None if self.frames is None else torch.zeros(B, *self.frame_shape, dtype=torch.float32).pin_memory() for _ in range(2)]
        self._slot_free = [None, None]            # event: the upload out of this slot has run
        self.chain = self.pipe.capture(self.ev, B, frames=self.frames, clip_offsets=np.zeros(B + 1, np.int64))
        self.copy_stream = torch.cuda.Stream(self.dev)
        self._chain_done = None
        self._pool = ThreadPoolExecutor(max_workers=1)

    def __len__(self):
        return self.n_batches

    def _pack(self, it, slot):
        """Host half of one batch (worker thread): B samples -> the pinned slot. -> (rows, names) or None at the end of the pass."""
        if self._slot_free[slot] is not None:
            self._slot_free[slot].synchronize()
        ev_h, off_h, fr_h = self._pin_ev[slot].numpy(), self._pin_off[slot].numpy(), self._pin_fr[slot]
        names, n = [], 0
        off_h[0] = 0
        for i in range(self.B):
            try:
                events, frame, name = next(it)
            except StopIteration:
                return None
            e = np.asarray(events, dtype=np.float64)
            if e.ndim != 2 or e.shape[1] != 4:
                raise ValueError("GpuEventLoader: events must be float64 [n,4] (x,y,t,p)")
            k = min(e.shape[0], self.cap)
            ev_h[n:n + k] = e[:k]
            n += k
            off_h[i + 1] = n
            if fr_h is not None:
                f = frame.numpy() if torch.is_tensor(frame) else np.asarray(frame, dtype=np.float32)
                if tuple(f.shape) != self.frame_shape:
                    raise ValueError(f"GpuEventLoader: frame of shape {tuple(f.shape)}, expected {self.frame_shape}")
                fr_h[i].numpy()[...] = f
            names.append(name)
        return n, names

    def _upload(self, slot, n):
        cs = self.copy_stream
        if self._chain_done is not None:
            cs.wait_event(self._chain_done)           # the previous replay has read the event buffer, the frames and the offsets
        with torch.cuda.stream(cs):
            self.ev[:n].copy_(self._pin_ev[slot][:n], non_blocking=True)
            self.chain.d_off.copy_(self._pin_off[slot], non_blocking=True)
            if self.frames is not None:
                self.frames.copy_(self._pin_fr[slot], non_blocking=True)
            done = torch.cuda.Event()
            done.record(cs)
        self._slot_free[slot] = done
        return done

    def __iter__(self):
        it = iter(self.samples)
        self.chain.set_state(self.step, self.first_sample)
        fut = self._pool.submit(self._pack, it, 0)
        for b in range(self.n_batches):
            slot = b & 1
            packed = fut.result()
            if packed is None:
                return
            n, names = packed
            up = self._upload(slot, n)
            if b + 1 < self.n_batches:
                fut = self._pool.submit(self._pack, it, slot ^ 1)      # the next batch is packed while this one is uploaded and trained on
            cur = torch.cuda.current_stream(self.dev)
            cur.wait_event(up)
            vox, tgt = self.chain.run_next()
            ev = torch.cuda.Event()
            ev.record(cur)
            self._chain_done = ev
            self.step += 1
            batch = {"events_voxel_grid": vox}
            if tgt is not None:
                batch[self.frame_key] = tgt
            batch["image_name"] = names
            # static tensors: the consumer's launches are queued on this stream before the next replay overwrites them
            yield batch
