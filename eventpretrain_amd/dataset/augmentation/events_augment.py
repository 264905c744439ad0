"""Event-level augmentation on the GPU behind the reference's function names (reference
dataset/augmentation/events_augment.py): window pick `get_random_index` (:5-20), sensor -> input rescale `events_reshape`
(:22-26), erase / add-correlated-events `erase_and_add_events` (:28-55) and `events_augment` (:80-86).

Split as in view_augment.py: the random DECISIONS are drawn on the host from the legacy numpy stream in the reference's
exact call order (so `np.random.seed(seed)` reproduces the reference's augmented clip bit for bit -- pinned by
tests/golden/events_augment.npz), the DATA MOVEMENT runs on the device (csrc/events.hip: the few added rows are sorted in
LDS and merged into the already time-sorted kept rows; no delete / concatenate / argsort of the whole clip).

`events_augment_batch` handles a batch of clips resident in HBM in two launches; together with
`voxel_grid_batch(..., scale=(sx, sy))` (the rescale fused into K1) and `view_augment.evg_augment_batch` this is the
loader's chain events -> events_augment -> events_reshape -> events_to_voxel_grid -> evg_augment
(pr_n_imagenet_dataset.py:82-89) without leaving the GPU."""
import numpy as np
import torch

from ... import _lib
from ..._lib import call, ptr, stream_ptr

MAX_ADD_PER_CLIP = 8192     # csrc/events.hip: the added rows of one clip are sorted in LDS


def get_random_index(args, events, is_train, seed=None):
    """Window of at most args.fix_events_num (train) / args.val_fix_events_num rows, start drawn from np.random
    (events_augment.py:5-20). Returns [start, end). Host-only: slicing a device tensor with it is free."""
    if seed is not None:
        np.random.seed(seed)
    fix_events_num = args.fix_events_num if is_train else args.val_fix_events_num
    n = int(events.shape[0])
    if n > fix_events_num:
        start_index = np.random.randint(0, n - fix_events_num)
        return start_index, start_index + fix_events_num
    return 0, n


def events_reshape(events, sensor_w, sensor_h, input_w, input_h):
    """x *= input_w / sensor_w, y *= input_h / sensor_h in place (float64; events_augment.py:22-26). Works on numpy
    arrays and on device tensors; in the batched pipeline pass scale=(input_w / sensor_w, input_h / sensor_h) to
    voxel_grid_batch instead -- K1 applies it while it reads the rows."""
    events[:, 0] *= (input_w / sensor_w)
    events[:, 1] *= (input_h / sensor_h)
    return events


def draw_erase_add(n, rs=None):
    """Decisions of erase_and_add_events for a clip of n rows, in the reference's draw order (events_augment.py:31-44):
    erase count, erased rows, add count, the three normal noise columns for ALL n rows, added rows. `rs`: a
    numpy.random.RandomState, default the process-global legacy stream (what the reference uses).
    Returns None for n < 100 (`int(0.01 n) == 0`: the reference leaves such clips alone), else
    (erase_index int64 [E] ascending, add_index int64 [A], add_noise float64 [A,3])."""
    rs = np.random if rs is None else rs
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


# ---- counter-based streams shared by host and device (csrc/events.hip ev_philox / ev_rand_word) ----------------------------------
_M32 = np.uint64(0xFFFFFFFF)


def philox_words(seed, step, samples, purpose, n_words):
    """uint32 [len(samples), n_words]: word w of sample s = Philox4x32-10 with counter (w // 4, purpose, s lo, s hi) and key
    (seed lo ^ hi32(step * 0x9E3779B97F4A7C15), seed hi ^ lo32(step)), output lane w % 4 -- the stream csrc/events.hip draws from on the
    device, restated with numpy array arithmetic (a few microseconds for a batch's worth of counts and crop boxes)."""
    samples = np.asarray(samples, dtype=np.uint64).reshape(-1, 1)
    nb = (int(n_words) + 3) // 4
    seed, step = int(seed) & (2 ** 64 - 1), int(step) & (2 ** 64 - 1)
    k0 = np.uint64((seed & 0xFFFFFFFF) ^ (((step * 0x9E3779B97F4A7C15) & (2 ** 64 - 1)) >> 32))
    k1 = np.uint64((seed >> 32) ^ (step & 0xFFFFFFFF))
    c0 = np.broadcast_to(np.arange(nb, dtype=np.uint64)[None, :], (samples.shape[0], nb)).copy()
    c1 = np.full_like(c0, np.uint64(purpose))
    c2 = np.broadcast_to(samples & _M32, c0.shape).copy()
    c3 = np.broadcast_to(samples >> np.uint64(32), c0.shape).copy()
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c0, np.uint64(0xCD9E8D57) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & _M32
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & _M32
        c1, c3, c0, c2 = p1 & _M32, p0 & _M32, n0, n2
        k0 = (k0 + np.uint64(0x9E3779B9)) & _M32
        k1 = (k1 + np.uint64(0xBB67AE85)) & _M32
    return np.stack([c0, c1, c2, c3], axis=2).reshape(samples.shape[0], nb * 4)[:, :n_words].astype(np.uint32)


def draw_erase_add_counts(seed, step, sizes, first_sample=0, words=None):
    """(erase_num, add_num) int64 [B] for windows of `sizes` rows: uniform in [int(0.001 n), int(0.01 n)) as the reference draws them
    (events_augment.py:31-33,38), 0 where int(0.01 n) == 0; from the shared counter stream (purpose 0, words 1 and 2; word 0 is the
    window start). The rows themselves and the noise are drawn on the device (evp_events_draw_erase_add)."""
    n = np.asarray(sizes, dtype=np.int64)
    w = (philox_words(seed, step, first_sample + np.arange(n.shape[0]), 0, 4) if words is None else words).astype(np.uint64)
    lo, hi = (0.001 * n).astype(np.int64), (0.01 * n).astype(np.int64)
    span = np.maximum(hi - lo, 0).astype(np.uint64)
    e = lo + ((w[:, 1] * span) >> np.uint64(32)).astype(np.int64)
    a_ = lo + ((w[:, 2] * span) >> np.uint64(32)).astype(np.int64)
    live = hi > 0
    return np.where(live, e, 0), np.where(live, a_, 0)


def draw_erase_add_batch(seed, step, sizes, first_sample=0):
    """Counter-based decisions for a batch drawn on the HOST: sample i of optimizer step `step` uses the Philox stream keyed by
    (seed, step, first_sample + i), so a clip's augmentation does not depend on worker scheduling. Same distribution as
    the reference's (counts uniform in [int(0.001 n), int(0.01 n)), rows without replacement, N(0,1.5) / N(0,1.5) /
    N(0,0.001) noise), but the noise is drawn for the added rows only (the reference draws 3 n normals and keeps <= 1 %).
    ~130 us of numpy calls per clip: the batched pipeline draws on the device instead (GpuInputPipeline decision_stream="device")."""
    out = []
    for i, n in enumerate(int(v) for v in sizes):
        if int(0.01 * n) <= 0:
            out.append(None)
            continue
        g = np.random.Generator(np.random.Philox(key=[int(seed) & (2 ** 64 - 1), ((int(step) << 24) ^ (first_sample + i)) & (2 ** 64 - 1)]))
        lo, hi = int(0.001 * n), int(0.01 * n)
        erase_index = np.sort(g.choice(n, size=int(g.integers(lo, hi)), replace=False)).astype(np.int64)
        add_index = g.choice(n, size=int(g.integers(lo, hi)), replace=False).astype(np.int64)
        noise = g.normal(0.0, 1.0, size=(add_index.size, 3)) * np.array([1.5, 1.5, 0.001])
        out.append((erase_index, add_index, np.ascontiguousarray(noise, dtype=np.float64)))
    return out


def events_augment_batch(events, clip_offsets, decisions, size, windows=None):
    """events: float64 CUDA tensor [n_total,4] (x,y,t,p), every clip time-sorted; clip_offsets: int64 host sequence /
    array [n_clips+1]; decisions: one `draw_erase_add` result (or None) per clip; size = (sensor_h, sensor_w).
    `windows`: optional int64 host array [n_clips,2] of clip-relative [start, end) rows -- `get_random_index`'s pick; the
    decisions' indices are then relative to the window and the output holds the augmented windows only.
    Returns (augmented events float64 CUDA [n_total',4], new clip_offsets int64 CUDA [n_clips+1])."""
    _lib.require_device()
    if not events.is_cuda or events.dtype != torch.float64 or events.dim() != 2 or events.shape[1] != 4 or not events.is_contiguous():
        raise _lib.EvpError("events_augment_batch: events must be a contiguous float64 [N,4] tensor in device memory")
    offs = np.asarray(clip_offsets.cpu() if torch.is_tensor(clip_offsets) else clip_offsets, dtype=np.int64)
    n_clips = offs.shape[0] - 1
    if windows is not None:
        win = np.asarray(windows, dtype=np.int64).reshape(n_clips, 2)
        if (win[:, 0] < 0).any() or (win[:, 1] < win[:, 0]).any() or (win[:, 1] > offs[1:] - offs[:-1]).any():
            raise _lib.EvpError("events_augment_batch: a window leaves its clip")
        w_beg, w_end = offs[:-1] + win[:, 0], offs[:-1] + win[:, 1]
    else:
        w_beg, w_end = offs[:-1], offs[1:]
    if len(decisions) != n_clips:
        raise _lib.EvpError("events_augment_batch: one decision entry per clip")
    dev = events.device
    er_l, ai_l, nz_l = [], [], []
    er_off, ad_off, out_off = np.zeros(n_clips + 1, np.int64), np.zeros(n_clips + 1, np.int64), np.zeros(n_clips + 1, np.int64)
    max_add = 0
    for c, d in enumerate(decisions):
        n = int(w_end[c] - w_beg[c])
        e = a = 0
        if d is not None:
            er, ai, nz = d
            if er.size and (er[0] < 0 or er[-1] >= n or np.any(np.diff(er) <= 0)):
                raise _lib.EvpError("events_augment_batch: erase_index of clip %d must be strictly ascending inside [0, %d)" % (c, n))
            if ai.size and (ai.min() < 0 or ai.max() >= n):
                raise _lib.EvpError("events_augment_batch: add_index of clip %d out of range" % c)
            e, a = int(er.size), int(ai.size)
            er_l.append(er), ai_l.append(ai), nz_l.append(nz.reshape(-1, 3))
        max_add = max(max_add, a)
        er_off[c + 1], ad_off[c + 1], out_off[c + 1] = er_off[c] + e, ad_off[c] + a, out_off[c] + n - e + a
    if max_add > MAX_ADD_PER_CLIP:
        raise _lib.EvpError("events_augment_batch: at most %d added rows per clip (got %d)" % (MAX_ADD_PER_CLIP, max_add))

    # ONE upload for the four tables (a pageable source makes every copy wait for the stream: four of them were four stalls of the host
    # per batch): erase indices | add indices | noise rows (float64 bits) | the five offset rows, all 8-byte words
    er_a = np.concatenate(er_l).astype(np.int64, copy=False) if er_l else np.zeros(0, np.int64)
    ai_a = np.concatenate(ai_l).astype(np.int64, copy=False) if ai_l else np.zeros(0, np.int64)
    nz_a = np.ascontiguousarray(np.concatenate(nz_l), dtype=np.float64).reshape(-1) if nz_l else np.zeros(0, np.float64)
    pad = lambda v: np.concatenate([v, np.zeros(n_clips + 1 - v.shape[0], np.int64)])
    tab_a = np.stack([pad(w_beg), pad(w_end), er_off, ad_off, out_off]).astype(np.int64, copy=False).reshape(-1)
    packed = torch.from_numpy(np.concatenate([er_a, ai_a, nz_a.view(np.int64), tab_a])).to(dev)
    o1, o2, o3 = er_a.size, er_a.size + ai_a.size, er_a.size + ai_a.size + nz_a.size
    er_d, ai_d = packed[:o1], packed[o1:o2]
    nz_d = packed[o2:o3].view(torch.float64).view(-1, 3)
    tabs = packed[o3:].view(5, n_clips + 1)
    n_add, n_out = int(ad_off[-1]), int(out_off[-1])
    ws = torch.empty(max(n_add, 1), 4, dtype=torch.float64, device=dev)
    out = torch.empty(n_out, 4, dtype=torch.float64, device=dev)
    call("evp_events_erase_add_win_f64", ptr(events), ptr(tabs[0]), ptr(tabs[1]), n_clips, ptr(er_d), ptr(tabs[2]), ptr(ai_d), ptr(nz_d),
         ptr(tabs[3]), max_add, float(size[1]), float(size[0]), ptr(ws), ptr(tabs[4]), ptr(out), stream_ptr())
    return out, tabs[4]


def erase_and_add_events(args, events, size=None):
    """Drop-in for the reference function (events_augment.py:28-55): events is a numpy float64 [N,4] (x,y,t,p) array or a
    device tensor, time-sorted; decisions come from the process-global numpy stream like the reference's. Returns the
    augmented, time-sorted clip in device memory."""
    _lib.require_device()
    if not torch.is_tensor(events):
        events = torch.from_numpy(np.ascontiguousarray(events, dtype=np.float64)).to(torch.device("cuda", torch.cuda.current_device()))
    n = int(events.shape[0])
    d = draw_erase_add(n)
    if d is None:
        return events
    out, _ = events_augment_batch(events, [0, n], [d], size)
    return out


def events_augment(args, events, size, seed=None):
    """Drop-in for events_augment.py:80-86."""
    if seed is not None:
        np.random.seed(seed)
    return erase_and_add_events(args, events, size=size)
