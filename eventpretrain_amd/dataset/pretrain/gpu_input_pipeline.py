"""The pre-training loader's per-sample chain as ONE batched call on clips already resident in HBM:

    get_random_index -> events_augment -> events_reshape -> events_to_voxel_grid -> evg_augment  (+ frame_augment for the target)

(reference dataset/pretrain/pr_n_imagenet_dataset.py:76-107; the seeded evg / frame pair as pr_ef_imagenet_dataset.py:187-206).
The reference runs it per sample in DataLoader workers at 10-27 ms per clip on a CPU core, which cannot feed a
5-12 k samples/s step; here the random DECISIONS are drawn on the host (a few hundred numbers per clip, overlappable with the
previous step) and every byte of event / voxel data stays on the device:

    window pick + erase / add  csrc/events.hip   evp_events_erase_add_win_f64  (2 launches: sort the <= 1 % added rows in LDS; merge)
    rescale + voxel scatter    csrc/voxel.hip    evp_voxel_scatter_scaled_f32  (K1, 3 launches in the verified mode)
    crop / resize / flips      csrc/augment.hip  evp_view_augment_f32, evp_frame_augment_f32

Decision streams: "device" (default, round 4) = Philox4x32-10 keyed by (seed, step, sample), the per-clip COUNTS, window starts and
crop boxes computed on the host with array arithmetic (~1 ms per batch) and the erase / add ROWS and noise drawn by a kernel
(evp_events_draw_erase_add) -- nothing per clip on the host, no decision tables to upload; "counter" = the same idea with numpy's own
Philox generator, everything drawn on the host (~13 ms per 64-clip batch: more than a training step); "legacy" = the reference's process-global numpy stream, in one of two orders (`legacy_order`):
  "n-imagenet"  PretrainNImageNetDataset.__getitem__'s own order (pr_n_imagenet_dataset.py:82-89): sample after sample on the RUNNING
                stream -- window, erase / add, evg_augment WITHOUT a seed; under np.random.seed(s) it reproduces what a DataLoader
                worker of the reference draws (tests/golden/loader_chain_nimagenet.npz, made by the reference's functions in that order);
  "chain"       the composition tests/golden/loader_chain.npz was made with (oracle/gen_golden.py gen_chain): the same events half, then
                the SEEDED evg_augment / frame_augment pair of pr_ef_imagenet_dataset.py:187-206 (`seed = np.random.randint(1000)`
                drawn at that point) -- it pins the chain's functions incl. the frame target, not a dataset's stream position (the
                EF-ImageNet dataset loads ready-made voxel grids and draws frame index and seed first).

Ways to run a batch, slowest first: `run(...)` does the host packing inline (simple, what the parity tests drive); `prepare(...)` /
`run_prepared(...)` split it -- every table of the batch (window bounds, erase / add indices, noise rows, offsets, crop rows for
grids and frames) is validated and packed into ONE pinned buffer by `prepare`, which a worker thread runs one batch ahead
(`prepare_async`); `capture(...)` -> CapturedChain replays the device half as one HIP graph, and with `clip_offsets` the chain is
SELF-DRIVEN: the plan itself (windows, counts, prefix sums, crop rows) is computed by a kernel of the graph (evp_events_plan_batch, bit
for bit the numpy form above) from a device-resident (step, first sample) pair the graph advances -- nothing per batch on the host.
The captured chain's default voxel stage is FUSED with the event augmentation and the view augmentation
(evp_voxel_scatter_fused_f32): it streams the original window rows, skips the erased ones, adds the built rows and flushes through
crop / resize / flips, so neither the merged clip nor the raw grids are ever written (round 4: 449 -> 262 us per 64-clip batch).
dataset.pretrain.gpu_event_loader.GpuEventLoader puts that chain at the DataLoader's place in the epoch loops."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from ... import _lib
from ..._lib import call, ptr, stream_ptr
from ..augmentation import events_augment as ea
from ..augmentation import view_augment as va
from ..dataset_utils.events_to_voxel_grid import voxel_grid_batch


class PreparedBatch:
    """Host half of one batch, ready to launch: `words` int64 [n] in pinned memory = erase indices | add indices | noise rows (float64
    bits) | five offset rows [n_clips + 1] (window begin, window end, erase / add / output offsets) | crop rows of the grids | crop rows
    of the frames (int32 [B,6] each, padded to 8 bytes), plus the few scalars the launches need. `busy` = event recorded after the
    upload that last read the pinned words (the slot is reused round-robin)."""
    __slots__ = ("words", "n_words", "o", "n_clips", "n_add", "n_out", "max_add", "windows", "params", "fparams", "slot", "sizes",
                 "n_erase", "max_cnt", "step", "first_sample", "on_device")


class GpuInputPipeline:
    def __init__(self, args, seed=0, decision_stream="device", ring=4, legacy_order="chain"):
        """args: the reference's namespace (fix_events_num, img_sensor_h / _w, input_size, num_bins, crop_min). `ring`: pinned slots
        for prepared batches (how many may exist at once before their run_prepared)."""
        self.RING = int(ring)
        if decision_stream not in ("device", "counter", "legacy"):
            raise ValueError("decision_stream must be 'device', 'counter' or 'legacy'")
        if legacy_order not in ("chain", "n-imagenet"):
            raise ValueError("legacy_order must be 'chain' or 'n-imagenet'")
        self.legacy_order = legacy_order
        self.args, self.seed, self.stream = args, int(seed), decision_stream
        self.sensor = (int(args.img_sensor_h), int(args.img_sensor_w))
        self.S = int(args.input_size)
        self.bins = int(args.num_bins)
        self.crop_min = float(getattr(args, "crop_min", 0.8))
        self._pins, self._busy, self._turn, self._pool = [None] * self.RING, [None] * self.RING, 0, None

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
            if self.legacy_order == "n-imagenet" and frame_size is not None:
                raise ValueError("legacy_order 'n-imagenet' has no frame target (that dataset pairs the grid with a CLIP image)")
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
                if self.legacy_order == "n-imagenet":           # evg_augment(args, grid, size): no seed, the stream runs on (:88-89)
                    params[i] = va.draw_evg_params(np.random, self.S, self.S, self.crop_min)
                    continue
                seed2 = np.random.randint(1000)
                np.random.seed(seed2)                           # evg_augment(..., seed=seed) re-seeds the global stream (view_augment.py:66-67)
                params[i] = va.draw_evg_params(np.random, self.S, self.S, self.crop_min)
                if fparams is not None:
                    np.random.seed(seed2)                       # frame_augment(..., seed=seed) does so again (view_augment.py:80-81)
                    fparams[i] = va.draw_evg_params(np.random, int(frame_size[0]), int(frame_size[1]), self.crop_min)
                    fparams[i, 5] = params[i, 5]
            return (windows, dec, params) if fparams is None else (windows, dec, params, fparams)
        sz = np.asarray([int(v) for v in sizes], dtype=np.int64)
        if self.stream == "device":
            words = ea.philox_words(self.seed, step, first_sample + np.arange(B), 0, 4)          # word 0: window start, 1 / 2: the two counts
            s0 = ((words[:, 0].astype(np.uint64) * np.maximum(sz - fix, 0).astype(np.uint64)) >> np.uint64(32)).astype(np.int64)
            windows[:, 0] = np.where(sz > fix, s0, 0)
            windows[:, 1] = np.where(sz > fix, s0 + fix, sz)
            # the rows and the noise are drawn on the device: the "decisions" are the two counts per clip
            dec = ea.draw_erase_add_counts(self.seed, step, windows[:, 1] - windows[:, 0], first_sample, words=words)
            U = va.evg_uniforms(self.seed, step, B, first_sample)
            params = va.draw_evg_params_batch(self.seed, step, B, self.S, self.S, self.crop_min, first_sample, U=U)
            if fparams is None:
                return windows, dec, params
            fparams = va.draw_evg_params_batch(self.seed, step, B, int(frame_size[0]), int(frame_size[1]), self.crop_min, first_sample, U=U)
            fparams[:, 5] = params[:, 5]
            return windows, dec, params, fparams
        for i, n in enumerate(int(v) for v in sz):
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
        if self.stream == "device" and isinstance(decisions, tuple) and len(decisions) == 2 and isinstance(decisions[0], np.ndarray):
            raise ValueError("GpuInputPipeline.run takes per-clip decision lists; with decision_stream='device' use prepare() / run_prepared() "
                             "(or batch()), or read the device's draws back with device_decisions()")
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

    # ------------------------------------------------------------------------------------------------ prepared form
    def prepare(self, clip_offsets, step, first_sample=0, sample_seeds=None, frame_size=None):
        """Host work of one batch (decisions, range checks, packing into a pinned slot). Thread-safe with respect to the device:
        touches no stream except to wait for the slot's previous upload."""
        offs = np.asarray(clip_offsets, dtype=np.int64)
        n_clips = offs.shape[0] - 1
        drawn = self.draw(offs[1:] - offs[:-1], step, first_sample, sample_seeds, frame_size=frame_size)
        windows, dec, params = drawn[0], drawn[1], drawn[2]
        fparams = drawn[3] if frame_size is not None else None
        H, W = self.sensor
        win = windows.reshape(n_clips, 2)
        if (win[:, 0] < 0).any() or (win[:, 1] < win[:, 0]).any() or (win[:, 1] > offs[1:] - offs[:-1]).any():
            raise _lib.EvpError("GpuInputPipeline.prepare: a window leaves its clip")
        w_beg, w_end = offs[:-1] + win[:, 0], offs[:-1] + win[:, 1]
        er_l, ai_l, nz_l = [], [], []
        tab = np.zeros((5, n_clips + 1), np.int64)
        tab[0, :n_clips], tab[1, :n_clips] = w_beg, w_end
        max_add = 0
        on_device = self.stream == "device"
        if on_device:
            e_num, a_num = dec
            tab[2, 1:], tab[3, 1:] = np.cumsum(e_num), np.cumsum(a_num)
            tab[4, 1:] = np.cumsum((w_end - w_beg) - e_num + a_num)
            max_add = int(max(a_num.max(), 0)) if n_clips else 0
            max_cnt = int(max(max_add, e_num.max())) if n_clips else 0
            dec = []
        for c, d in enumerate(dec):
            n = int(w_end[c] - w_beg[c])
            e = a = 0
            if d is not None:
                er, ai, nz = d
                if er.size and (er[0] < 0 or er[-1] >= n or np.any(np.diff(er) <= 0)):
                    raise _lib.EvpError("GpuInputPipeline.prepare: erase_index of clip %d must be strictly ascending inside [0, %d)" % (c, n))
                if ai.size and (ai.min() < 0 or ai.max() >= n):
                    raise _lib.EvpError("GpuInputPipeline.prepare: add_index of clip %d out of range" % c)
                e, a = int(er.size), int(ai.size)
                er_l.append(er), ai_l.append(ai), nz_l.append(nz.reshape(-1))
            max_add = max(max_add, a)
            tab[2, c + 1], tab[3, c + 1], tab[4, c + 1] = tab[2, c] + e, tab[3, c] + a, tab[4, c] + n - e + a
        if max_add > ea.MAX_ADD_PER_CLIP:
            raise _lib.EvpError("GpuInputPipeline.prepare: at most %d added rows per clip (got %d)" % (ea.MAX_ADD_PER_CLIP, max_add))
        if not on_device:
            max_cnt = max_add
        rows = np.ascontiguousarray(params, dtype=np.int32).reshape(n_clips, 6)
        S = self.S
        if ((rows[:, 0] < 0) | (rows[:, 1] < 0) | (rows[:, 2] < 1) | (rows[:, 3] < 1) | (rows[:, 0] + rows[:, 2] > S) | (rows[:, 1] + rows[:, 3] > S)).any():
            raise ValueError("GpuInputPipeline.prepare: crop box outside the view")
        n_e, n_a = int(tab[2, -1]), int(tab[3, -1])
        pw = (n_clips * 6 + 1) // 2                      # int32 [B,6] in 8-byte words
        o = [0, n_e, n_e + n_a, n_e + n_a + 3 * n_a, 0, 0, 0] if not on_device else [0, 0, 0, 0, 0, 0, 0]     # (device stream: no decision tables)
        o[4] = o[3] + 5 * (n_clips + 1)
        o[5] = o[4] + pw
        o[6] = o[5] + (pw if fparams is not None else 0)
        slot = self._turn
        self._turn = (slot + 1) % self.RING
        if self._busy[slot] is not None:
            self._busy[slot].synchronize()               # the upload that last read this slot has run
        pin = self._pins[slot]
        if pin is None or pin.numel() < o[6]:
            pin = self._pins[slot] = torch.empty(max(o[6] * 2, 1 << 16), dtype=torch.int64).pin_memory()
        w = pin.numpy()
        if n_e and not on_device:
            np.concatenate(er_l, out=w[o[0]:o[1]])
        if n_a and not on_device:
            np.concatenate(ai_l, out=w[o[1]:o[2]])
            np.concatenate(nz_l, out=w[o[2]:o[3]].view(np.float64))
        w[o[3]:o[4]] = tab.reshape(-1)
        w[o[4]:o[5]].view(np.int32)[:n_clips * 6] = rows.reshape(-1)
        if fparams is not None:
            fr = np.ascontiguousarray(fparams, dtype=np.int32).reshape(n_clips, 6)
            Hf, Wf = int(frame_size[0]), int(frame_size[1])
            if ((fr[:, 0] < 0) | (fr[:, 1] < 0) | (fr[:, 2] < 1) | (fr[:, 3] < 1) | (fr[:, 0] + fr[:, 2] > Wf) | (fr[:, 1] + fr[:, 3] > Hf)).any():
                raise ValueError("GpuInputPipeline.prepare: frame crop box outside the frame")
            w[o[5]:o[6]].view(np.int32)[:n_clips * 6] = fr.reshape(-1)
        pb = PreparedBatch()
        pb.words, pb.n_words, pb.o, pb.n_clips, pb.n_add, pb.n_out, pb.max_add = pin, o[6], o, n_clips, n_a, int(tab[4, -1]), max_add
        pb.n_erase, pb.max_cnt, pb.step, pb.first_sample, pb.on_device = n_e, max_cnt, int(step), int(first_sample), on_device
        pb.windows, pb.params, pb.fparams, pb.slot, pb.sizes = windows, params, fparams, slot, win[:, 1] - win[:, 0]
        return pb

    def prepare_async(self, *a, **kw):
        """prepare(...) on the pipeline's worker thread (one batch ahead of the device); returns a Future."""
        if self._pool is None:
            self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="evp-loader")
        return self._pool.submit(self.prepare, *a, **kw)

    def run_prepared(self, events, pb, frames=None, assume_sorted=True):
        """Device half: ONE upload of the packed tables, then erase / add merge (2 launches), K1 with the sensor -> input rescale (3),
        view augmentation (1) and the target's frame augmentation (1). No host read-back, no per-batch host arithmetic."""
        _lib.require_device()
        if not events.is_cuda or events.dtype != torch.float64 or events.dim() != 2 or events.shape[1] != 4 or not events.is_contiguous():
            raise _lib.EvpError("run_prepared: events must be a contiguous float64 [N,4] tensor in device memory")
        dev, o, nc = events.device, pb.o, pb.n_clips
        d = torch.empty(pb.n_words, dtype=torch.int64, device=dev)
        d.copy_(pb.words[:pb.n_words], non_blocking=True)
        busy = self._busy[pb.slot] or torch.cuda.Event()
        busy.record()
        self._busy[pb.slot] = busy
        tabs = d[o[3]:o[4]].view(5, nc + 1)
        H, W = self.sensor
        ws = torch.empty(max(pb.n_add, 1), 4, dtype=torch.float64, device=dev)
        ev = torch.empty(pb.n_out, 4, dtype=torch.float64, device=dev)
        if pb.on_device:
            er_d, ai_d, nz_d = self._draw_on_device(pb, tabs, dev)
        else:
            er_d, ai_d, nz_d = d[o[0]:o[1]], d[o[1]:o[2]], d[o[2]:o[3]]
        call("evp_events_erase_add_win_f64", ptr(events), ptr(tabs[0]), ptr(tabs[1]), nc, ptr(er_d), ptr(tabs[2]), ptr(ai_d),
             ptr(nz_d), ptr(tabs[3]), pb.max_add, float(W), float(H), ptr(ws), ptr(tabs[4]), ptr(ev), stream_ptr())
        vox = voxel_grid_batch(ev, tabs[4], self.bins, (self.S, self.S), assume_sorted=assume_sorted, scale=(self.S / W, self.S / H))
        p_dev = d[o[4]:o[5]].view(torch.int32)[:nc * 6].view(nc, 6)
        out = va.evg_augment_batch(vox, p_dev, (self.S, self.S))
        tgt = None
        if frames is not None:
            fp = d[o[5]:o[6]].view(torch.int32)[:nc * 6].view(nc, 6) if pb.fparams is not None else p_dev
            tgt = va.frame_augment_batch(frames, fp, (self.S, self.S))
        return out, tgt

    def _draw_on_device(self, pb, tabs, dev):
        """The erase / add rows and the noise of a prepared batch, drawn by evp_events_draw_erase_add into fresh device arrays."""
        er_d = torch.empty(max(pb.n_erase, 1), dtype=torch.int64, device=dev)
        ai_d = torch.empty(max(pb.n_add, 1), dtype=torch.int64, device=dev)
        nz_d = torch.empty(max(pb.n_add, 1) * 3, dtype=torch.float64, device=dev)
        call("evp_events_draw_erase_add", ptr(tabs[0]), ptr(tabs[1]), pb.n_clips, ptr(tabs[2]), ptr(tabs[3]), self.seed & (2 ** 64 - 1),
             pb.step & (2 ** 64 - 1), pb.first_sample, None, pb.max_cnt, ptr(er_d), ptr(ai_d), ptr(nz_d), stream_ptr())
        return er_d, ai_d, nz_d

    def device_decisions(self, pb, device):
        """Read the device stream's draws of a prepared batch back as the per-clip lists run() / events_augment_batch take (tests, debugging:
        one upload, one launch, one read-back)."""
        d = torch.empty(pb.n_words, dtype=torch.int64, device=device)
        d.copy_(pb.words[:pb.n_words])
        tabs = d[pb.o[3]:pb.o[4]].view(5, pb.n_clips + 1)
        er_d, ai_d, nz_d = self._draw_on_device(pb, tabs, d.device)
        t = tabs.cpu().numpy()
        er, ai, nz = er_d.cpu().numpy(), ai_d.cpu().numpy(), nz_d.cpu().numpy().reshape(-1, 3)
        out = []
        for c in range(pb.n_clips):
            e, a = slice(t[2, c], t[2, c + 1]), slice(t[3, c], t[3, c + 1])
            out.append(None if (e.stop == e.start and a.stop == a.start) else (er[e].copy(), ai[a].copy(), nz[a].copy()))
        return out

    def capture(self, events, n_clips, frames=None, clip_offsets=None, fused_voxel=True, out=None, tgt_out=None):
        """The device half as ONE HIP graph (device decision stream): `events` is the buffer every batch's raw rows will sit in (fixed
        address: the loader uploads into it), `frames` likewise. Launched from Python the eight kernels of a batch cost more host time
        (~0.9 ms with the worker thread competing for the interpreter) than the GPU needs for them (~0.42 ms); replayed they cost one
        call. Every buffer gets its upper bound (windows of at most fix_events_num rows, counts below int(0.01 fix_events_num)); what
        varies per batch -- offsets, crop rows, (step, first sample) -- travels through one pinned table the graph's upload node
        re-reads. Returns a CapturedChain; its run(prepared) hands out the two STATIC output tensors (consume or clone them before the
        next run).
        With `clip_offsets` (int64 [n_clips + 1], host or device) the chain is SELF-DRIVEN: the batch's plan (windows, counts, offsets,
        crop rows) is computed by a kernel inside the graph (evp_events_plan_batch) from the clip offsets and a device-resident (step,
        first sample) pair that the graph itself advances -- `run_next()` is one replay, with nothing prepared, packed or uploaded on the
        host; `set_clip_offsets` / `set_state` change the inputs between replays.
        `fused_voxel` (default): the voxel grids are binned straight from the window rows, the erase list and the added rows
        (evp_voxel_scatter_fused_f32) -- the merged clip, which only K1 would read, is never written.
        `out` / `tgt_out`: tensors the chain writes its grids [n_clips, bins, S, S] / targets [n_clips, C, S, S] INTO -- the static inputs
        of a step executor, say, which then needs no copy between the two graphs."""
        return CapturedChain(self, events, int(n_clips), frames, clip_offsets, fused_voxel, out, tgt_out)

    def batch(self, events, clip_offsets, step, frames=None, first_sample=0, sample_seeds=None):
        """The whole chain for one batch: decisions on the host, data on the device."""
        if self.stream == "device":
            offs = np.asarray(clip_offsets.cpu() if torch.is_tensor(clip_offsets) else clip_offsets, dtype=np.int64)
            pb = self.prepare(offs, step, first_sample, sample_seeds, frame_size=None if frames is None else frames.shape[-2:])
            return self.run_prepared(events, pb, frames=frames)
        offs = np.asarray(clip_offsets.cpu() if torch.is_tensor(clip_offsets) else clip_offsets, dtype=np.int64)
        if frames is None:
            windows, dec, params = self.draw(offs[1:] - offs[:-1], step, first_sample, sample_seeds)
            return self.run(events, offs, windows, dec, params)
        windows, dec, params, fparams = self.draw(offs[1:] - offs[:-1], step, first_sample, sample_seeds, frame_size=frames.shape[-2:])
        return self.run(events, offs, windows, dec, params, frames, fparams)

    def algorithmic_bytes(self, sizes, fused=False):
        """HBM bytes the chain has to move for clips whose picked windows hold `sizes` rows (the figure bench.py prices the chain
        with): read the window (32 B / row) + write the augmented clip (32 B / row; +- 1 %) + K1 reads it again and writes the grid
        once + the view augmentation reads the grid and writes the view. `fused` (CapturedChain's default form): neither the augmented
        clip nor the raw grid exists -- the window is read once, the augmented view written once."""
        n = float(np.sum(sizes))
        grid = self.bins * self.S * self.S * 4.0 * len(sizes)
        return (n * 32 + grid) if fused else (n * 32 + n * 32 + n * 32 + grid + grid + grid)


class CapturedChain:
    """GpuInputPipeline.capture: the chain's device half captured once, replayed per batch -- with the batch's tables prepared on the host
    (run(prepared)) or, self-driven, planned by a kernel of the graph (run_next())."""

    def __init__(self, pipe, events, n_clips, frames, clip_offsets=None, fused_voxel=True, out=None, tgt_out=None):
        if pipe.stream != "device":
            raise ValueError("CapturedChain needs decision_stream='device' (host-drawn decision lists change size per batch)")
        _lib.require_device()
        self.pipe, self.events, self.frames, self.nc = pipe, events, frames, n_clips
        dev = events.device
        fix = int(pipe.args.fix_events_num)
        self.kmax = max(int(0.01 * fix), 1)
        pw = (n_clips * 6 + 1) // 2
        # table layout of a device-stream PreparedBatch: five offset rows | crop rows | frame crop rows; two more words: step, first sample
        self.n_tab = 5 * (n_clips + 1) + pw + (pw if frames is not None else 0)
        self.h_tab = torch.zeros(self.n_tab + 2, dtype=torch.int64).pin_memory()      # what the graph's upload node reads (prepared form)
        self.d_tab = torch.zeros(self.n_tab + 2, dtype=torch.int64, device=dev)
        self.er = torch.zeros(n_clips * self.kmax, dtype=torch.int64, device=dev)
        self.ai = torch.zeros(n_clips * self.kmax, dtype=torch.int64, device=dev)
        self.nz = torch.zeros(n_clips * self.kmax * 3, dtype=torch.float64, device=dev)
        self.ws = torch.zeros(n_clips * self.kmax, 4, dtype=torch.float64, device=dev)
        self.fused = bool(fused_voxel) and fix <= 48 * 1024 * 8
        self.ev = None if self.fused else torch.zeros(n_clips * (fix + self.kmax), 4, dtype=torch.float64, device=dev)
        self.kws = torch.zeros(n_clips * (pipe.bins + 5), dtype=torch.int64, device=dev) if self.fused else None
        self._busy = None
        for t_, shp, who in ((out, (n_clips, pipe.bins, pipe.S, pipe.S), "out"),
                             (tgt_out, None if frames is None else (n_clips, frames.shape[1], pipe.S, pipe.S), "tgt_out")):
            if t_ is not None and (shp is None or tuple(t_.shape) != shp or t_.dtype != torch.float32 or not t_.is_contiguous() or t_.device != dev):
                raise ValueError(f"CapturedChain: {who} must be a contiguous float32 tensor of shape {shp} on the events' device")
        self._out_given, self._tgt_given = out, tgt_out
        self.self_driven = clip_offsets is not None
        if self.self_driven:
            self.d_off = torch.zeros(n_clips + 1, dtype=torch.int64, device=dev)
            self.state = torch.zeros(2, dtype=torch.int64, device=dev)          # (step, first sample): advanced by the graph itself
            self.set_clip_offsets(clip_offsets)
        # warm-up (allocator, kernel attributes) on a side stream, then capture
        sizes = np.full(n_clips, min(fix, int(events.shape[0]) // max(n_clips, 1)), dtype=np.int64)
        off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            if not self.self_driven:
                self._stage(pipe.prepare(off, step=0, frame_size=None if frames is None else frames.shape[-2:]))
            self._launches()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):      # (RCCL's watchdog thread may poll its events meanwhile)
            self.out, self.tgt = self._launches()
        if self.self_driven:
            self.set_state(0, 0)                      # the warm-up advanced it

    def set_clip_offsets(self, clip_offsets):
        """Self-driven chain: the clip offsets of the rows now in the event buffer (int64 [n_clips + 1], host or device). Every clip's
        window is at most fix_events_num rows by construction, so any ascending offsets inside the captured buffer are valid."""
        off = clip_offsets if torch.is_tensor(clip_offsets) else torch.from_numpy(np.ascontiguousarray(clip_offsets, dtype=np.int64))
        if off.numel() != self.nc + 1 or off.dtype != torch.int64:
            raise ValueError("CapturedChain.set_clip_offsets: int64 [n_clips + 1] expected")
        if not off.is_cuda:
            o = off.numpy()
            if o[0] < 0 or (np.diff(o) < 0).any() or o[-1] > self.events.shape[0]:
                raise ValueError("CapturedChain.set_clip_offsets: offsets must ascend inside the event buffer")
        self.d_off.copy_(off)

    def set_state(self, step, first_sample=0):
        """Self-driven chain: the (step, first sample) the NEXT replay draws for (every replay then advances step by one)."""
        self.state.copy_(torch.tensor([int(step), int(first_sample)], dtype=torch.int64))

    def _stage(self, pb):
        if pb.n_clips != self.nc or not pb.on_device or pb.n_words != self.n_tab:
            raise ValueError("CapturedChain.run: the prepared batch does not have the captured shape (clips, frames, decision stream)")
        if pb.max_cnt > self.kmax or int(pb.sizes.max()) > int(self.pipe.args.fix_events_num):
            raise ValueError("CapturedChain.run: a window or a count exceeds the captured upper bounds")
        h = self.h_tab.numpy()
        h[:self.n_tab] = pb.words.numpy()[:self.n_tab]
        h[self.n_tab], h[self.n_tab + 1] = pb.step, pb.first_sample

    def _launches(self):
        pipe, nc = self.pipe, self.nc
        d = self.d_tab
        tabs = d[:5 * (nc + 1)].view(5, nc + 1)
        pw = (nc * 6 + 1) // 2
        o4 = 5 * (nc + 1)
        H, W = pipe.sensor
        fr = self.frames
        if self.self_driven:
            # the plan of the batch, on the device: windows, counts and their prefix sums, crop rows; (step, first sample) -> d[n_tab:]
            call("evp_events_plan_batch", ptr(self.d_off), nc, int(pipe.args.fix_events_num), pipe.seed & (2 ** 64 - 1), ptr(self.state), 1,
                 ptr(d[self.n_tab:]), pipe.S, pipe.S, 0 if fr is None else int(fr.shape[-2]), 0 if fr is None else int(fr.shape[-1]),
                 float(pipe.crop_min), ptr(tabs), ptr(d[o4:]), None if fr is None else ptr(d[o4 + pw:]), stream_ptr())
        else:
            d.copy_(self.h_tab, non_blocking=True)        # an upload node of the graph
        call("evp_events_draw_erase_add", ptr(tabs[0]), ptr(tabs[1]), nc, ptr(tabs[2]), ptr(tabs[3]), pipe.seed & (2 ** 64 - 1), 0, 0,
             ptr(d[self.n_tab:]), self.kmax, ptr(self.er), ptr(self.ai), ptr(self.nz), stream_ptr())
        if self.fused:
            # the added rows built and time-sorted (in the draw kernel's add workgroup they buy nothing: 261 vs 262 us per batch), then the
            # grids straight from window rows + erase list + added rows (no merged clip) ...
            call("evp_events_build_added_f64", ptr(self.events), ptr(tabs[0]), nc, ptr(self.ai), ptr(self.nz), ptr(tabs[3]), self.kmax, float(W),
                 float(H), ptr(self.ws), stream_ptr())
            # ... and they leave through the view augmentation (crop / nearest resize / flips), so the raw grids are never stored either
            p_dev = d[o4:o4 + pw].view(torch.int32)[:nc * 6].view(nc, 6)
            out = self._out_given if self._out_given is not None else torch.empty(nc, pipe.bins, pipe.S, pipe.S, dtype=torch.float32, device=d.device)
            call("evp_voxel_scatter_fused_f32", ptr(self.events), ptr(tabs[0]), ptr(tabs[1]), nc, ptr(self.er), ptr(tabs[2]), ptr(self.ws),
                 ptr(tabs[3]), int(pipe.args.fix_events_num), pipe.bins, pipe.S, pipe.S, pipe.S / W, pipe.S / H, ptr(p_dev), pipe.S, pipe.S,
                 int(pipe.bins in (5, 6)), ptr(self.kws), ptr(out), stream_ptr())
        else:
            call("evp_events_erase_add_win_f64", ptr(self.events), ptr(tabs[0]), ptr(tabs[1]), nc, ptr(self.er), ptr(tabs[2]), ptr(self.ai),
                 ptr(self.nz), ptr(tabs[3]), self.kmax, float(W), float(H), ptr(self.ws), ptr(tabs[4]), ptr(self.ev), stream_ptr())
            vox = voxel_grid_batch(self.ev, tabs[4], pipe.bins, (pipe.S, pipe.S), assume_sorted=True, scale=(pipe.S / W, pipe.S / H))
            p_dev = d[o4:o4 + pw].view(torch.int32)[:nc * 6].view(nc, 6)
            out = va.evg_augment_batch(vox, p_dev, (pipe.S, pipe.S), out=self._out_given)
        tgt = None
        if fr is not None:
            # (as a parallel branch of the graph the frame targets cost more than they hide: 408 vs 389 us per batch with fork + join)
            fp = d[o4 + pw:o4 + 2 * pw].view(torch.int32)[:nc * 6].view(nc, 6)
            tgt = va.frame_augment_batch(fr, fp, (pipe.S, pipe.S), out=self._tgt_given)
        return out, tgt

    def run(self, pb):
        """One batch of the PREPARED form: the tables into the pinned slot, one replay. -> (voxels, targets): static tensors."""
        if self.self_driven:
            raise ValueError("CapturedChain.run: this chain plans its batches on the device; use run_next()")
        if self._busy is not None:
            self._busy.synchronize()                  # the previous replay is done: its upload node has read the pinned table and its
        self._stage(pb)                               # outputs have been consumed by whatever the caller queued behind it
        self.graph.replay()
        ev = self._busy or torch.cuda.Event()
        ev.record()
        self._busy = ev
        return self.out, self.tgt

    def run_next(self):
        """Self-driven chain: one replay = one batch; nothing else happens on the host. -> (voxels, targets): static tensors, overwritten
        by the next replay in stream order."""
        if not self.self_driven:
            raise ValueError("CapturedChain.run_next: capture with clip_offsets for the self-driven form")
        self.graph.replay()
        return self.out, self.tgt
