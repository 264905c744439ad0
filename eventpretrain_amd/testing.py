"""Deterministic, RNG-free input/weight generators shared by the golden-vector
generator (oracle/gen_golden.py), the parity tests and bench.py.

Everything here is exact integer hashing followed by one float64 division, so
the same bytes come out in the build container (where the reference is
imported to make the fixtures) and on the GPU box (where it is not present).
Nothing in this file is taken from the reference; it only fixes *inputs*.
"""
import zlib
from types import SimpleNamespace

import numpy as np
import torch

_M1 = np.uint64(0xFF51AFD7ED558CCD)
_M2 = np.uint64(0xC4CEB9FE1A85EC53)


def _mix64(x: np.ndarray) -> np.ndarray:
    """murmur3 finaliser on a uint64 array (wrapping arithmetic)."""
    x = x.copy()
    x ^= x >> np.uint64(33)
    x *= _M1
    x ^= x >> np.uint64(33)
    x *= _M2
    x ^= x >> np.uint64(33)
    return x


def det_uniform(name: str, shape, lo: float = -1.0, hi: float = 1.0) -> torch.Tensor:
    """float32 tensor of `shape`, uniform-looking in [lo, hi), keyed by `name`."""
    n = int(np.prod(shape)) if len(shape) else 1
    seed = np.uint64(zlib.crc32(name.encode()) * 0x9E3779B1 + 0x7F4A7C15)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + seed
        u = _mix64(idx) >> np.uint64(11)
    v = u.astype(np.float64) / float(1 << 53)          # [0,1)
    v = lo + (hi - lo) * v
    return torch.from_numpy(v.astype(np.float32).reshape(tuple(shape)))


def det_normalish(name: str, shape) -> torch.Tensor:
    """Sum of 4 uniforms, centred, unit variance: a cheap bell-shaped fill."""
    acc = sum(det_uniform(f"{name}#{i}", shape, 0.0, 1.0).double() for i in range(4))
    return ((acc - 2.0) * (3.0 ** 0.5)).float()


def det_value_for(name: str, shape) -> torch.Tensor:
    """Closed-form value for a parameter/buffer called `name` (state-dict key).

    Magnitudes follow what a trained-ish ViT looks like so that softmax,
    LayerNorm and GELU all operate away from degenerate regimes:
      * LayerNorm/BatchNorm weight: 1 +- 0.2; any bias: +-0.05
      * >=2-D weights: xavier-uniform bound sqrt(6/(fan_in+fan_out))
      * mask_token: +-0.05
    """
    shape = tuple(shape)
    leaf = name.split(".")[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_var":
        return det_uniform(name, shape, 0.5, 1.5)
    if leaf == "running_mean":
        return det_uniform(name, shape, -0.1, 0.1)
    if leaf == "bias":
        return det_uniform(name, shape, -0.05, 0.05)
    if leaf == "mask_token":
        return det_uniform(name, shape, -0.05, 0.05)
    if leaf == "weight" and len(shape) == 1:
        return det_uniform(name, shape, 0.8, 1.2)
    if len(shape) >= 2:
        fan_out = shape[0]
        fan_in = int(np.prod(shape[1:]))
        bound = (6.0 / (fan_in + fan_out)) ** 0.5
        return det_uniform(name, shape, -bound, bound)
    return det_uniform(name, shape, -0.05, 0.05)


@torch.no_grad()
def det_fill_module_(module: torch.nn.Module, skip=("pos_embed", "queue_ptr")) -> None:
    """Overwrite every parameter and float buffer of `module` in place with
    `det_value_for(key)`. Keys are state-dict keys, so the reference classes
    and this repo's mirrors get bit-identical weights."""
    for key, t in module.state_dict().items():
        leaf = key.split(".")[-1]
        if leaf in skip:
            continue
        if leaf == "queue":
            q = det_uniform(key, t.shape, -1.0, 1.0)
            q = torch.nn.functional.normalize(q, dim=0)
            t.copy_(q.to(t.dtype))
            continue
        t.copy_(det_value_for(key, t.shape).to(t.dtype))


def synthetic_events(i: int, n: int = 100_000, width: int = 224, height: int = 224,
                     duration: float = 0.05, signed_polarity: bool = False) -> np.ndarray:
    """SURVEY.md 8(d) synthetic clip `i`: float64 [n,4] rows (x, y, t[s], p),
    time-sorted, integer-valued pixel coordinates, p in {0,1} (or +-1)."""
    rng = np.random.default_rng(1234 + i)
    x = np.floor(rng.uniform(0, width, n))
    y = np.floor(rng.uniform(0, height, n))
    t = np.sort(rng.uniform(0.0, duration, n))
    p = (rng.uniform(0, 1, n) < 0.5).astype(np.float64)
    if signed_polarity:
        p = 2.0 * p - 1.0
    return np.stack([x, y, t, p], axis=1).astype(np.float64)


def make_args(**overrides) -> SimpleNamespace:
    """The flat argparse namespace the reference threads through every
    constructor (main_pretrain.py:32-169), with the same defaults for the
    fields read on the hot path."""
    a = dict(
        phase="pretrain", pr_phase="rec", backbone_type="vit", model_size="small",
        patch_size=16, num_bins=5, frame_chans=1, input_size=224,
        mask_ratio=0.5, masking_strategy="random", use_feature_fusion=True,
        norm_pix_loss=True, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0,
        use_queue=True, queue_length=1024, T=0.07, emb_frames_dim=512,
        distributed=False, device="cpu", accum_iter=1, print_freq=1000, log_freq=10,
        test_experiment=False, visualize=False, vis_train_freq=100, backward=True,
        blr=1e-3, lr=None, min_lr=0.0, warmup_epochs=20, epochs=5, weight_decay=0.05,
        layer_decay=0.75, use_layer_decay=False, use_layer_grafted=False,
        batch_size=2, seed=0, start_epoch=0, world_size=1, rank=0,
        # loader-side fields (main_pretrain.py:56,61,141-142)
        crop_min=0.8, fix_events_num=15000, val_fix_events_num=15000, img_sensor_w=640, img_sensor_h=480, graph_step=True,
    )
    a.update(overrides)
    return SimpleNamespace(**a)
