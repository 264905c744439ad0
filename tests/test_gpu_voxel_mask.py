"""K1 (voxel histogram) and K2 (mask ids) on the GPU, through the C-ABI, against the golden fixtures (made by the
reference) and the CPU oracle. Float tolerance for K1: 1e-5 abs (f32 accumulation order), exact on the
known-answer clip. K2 is integer work: bit-exact."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import jl

pytestmark = pytest.mark.gpu


def _batch(evs, bins, size, **kw):
    from eventpretrain_amd.dataset.dataset_utils.events_to_voxel_grid import voxel_grid_batch
    off = np.concatenate([[0], np.cumsum([e.shape[0] for e in evs])]).astype(np.int64)
    ev = torch.from_numpy(np.concatenate(evs, 0)).cuda()
    return voxel_grid_batch(ev, torch.from_numpy(off).cuda(), bins, size, **kw).cpu().numpy()


@pytest.mark.parametrize("algo", [0, 1, 2])
def test_voxel_kat_exact(algo):
    d = load_golden("voxel")
    g = _batch([d["kat_events"]], 5, (4, 4), algo=algo)[0]
    assert np.array_equal(g, d["kat_grid"])


@pytest.mark.parametrize("algo,tile_rows", [(0, 0), (0, 1), (0, 5), (0, 1000), (2, 0), (2, 3), (1, 0)])
def test_voxel_cases_vs_reference(algo, tile_rows):
    from oracle.voxel_oracle import voxel_grid
    d = load_golden("voxel")
    for c in jl(d["cases"]):
        ev = d[c["tag"] + "_events"]
        tr = min(tile_rows, c["H"]) if tile_rows else 0
        g = _batch([ev], c["bins"], (c["H"], c["W"]), is_txyp=c["is_txyp"], algo=algo, tile_rows=tr)[0]
        ref = d[c["tag"] + "_grid"]
        assert np.abs(g - ref).max() <= 1e-5, (c["tag"], np.abs(g - ref).max())
        assert np.abs(g - voxel_grid(ev, c["bins"], (c["H"], c["W"]), c["is_txyp"])).max() <= 1e-5


def test_voxel_drop_in_function_and_unsorted():
    from eventpretrain_amd.dataset.dataset_utils.events_to_voxel_grid import events_to_voxel_grid
    from eventpretrain_amd.testing import make_args
    from oracle.voxel_oracle import voxel_grid
    d = load_golden("voxel")
    ev = d["rand0_events"]
    g = events_to_voxel_grid(make_args(num_bins=5), ev, (32, 48))
    assert g.is_cuda and g.dtype == torch.float32 and tuple(g.shape) == (5, 32, 48)
    assert np.abs(g.cpu().numpy() - d["rand0_grid"]).max() <= 1e-5
    # shuffled rows: the reference takes t0/t1 from the first/last ROW; result must still agree with the oracle
    rng = np.random.default_rng(5)
    sh = ev[rng.permutation(ev.shape[0])]
    g2 = events_to_voxel_grid(make_args(num_bins=5), sh, (32, 48)).cpu().numpy()
    assert np.abs(g2 - voxel_grid(sh, 5, (32, 48))).max() <= 1e-5


def test_voxel_ragged_batch_and_empty_clip():
    from eventpretrain_amd.testing import synthetic_events
    from oracle.voxel_oracle import voxel_grid
    evs = [synthetic_events(200 + i, n, width=64, height=40) for i, n in enumerate([1, 7000, 0, 333, 20000, 64, 2, 5000, 123])]
    evs[2] = np.zeros((0, 4))
    for tr in (0, 7):
        g = _batch(evs, 5, (40, 64), tile_rows=tr)
        for i, e in enumerate(evs):
            ref = voxel_grid(e, 5, (40, 64)) if e.shape[0] else np.zeros((5, 40, 64), np.float32)
            assert np.abs(g[i] - ref).max() <= 1e-5, (tr, i)


def test_voxel_full_size_batch():
    """BASELINE config size: 64 clips x 100k events -> 5x224x224. Checked against the reference-made fixture for
    clip 0, the C oracle for a few clips, and a size-independent property for all of them: the grid total equals the
    sum of the per-event contributions (each in-range event adds p*(1-dt) + p*dt)."""
    from eventpretrain_amd.testing import synthetic_events
    from oracle.voxel_oracle import voxel_grid
    d = load_golden("voxel")
    evs = [synthetic_events(i) for i in range(64)]
    g = _batch(evs, 5, (224, 224))
    assert np.abs(g[0][:, ::7, ::5] - d["full0_sample"]).max() <= 1e-5
    for i in (0, 17, 63):
        assert np.abs(g[i] - voxel_grid(evs[i], 5, (224, 224))).max() <= 1e-5
    for i in range(64):
        e = evs[i]
        p = np.where(e[:, 3] == 0, -1.0, e[:, 3])
        ts = 4 * (e[:, 2] - e[0, 2]) / (e[-1, 2] - e[0, 2])
        tf = np.floor(ts)
        dtf = (ts - tf).astype(np.float32).astype(np.float64)
        expect = np.sum(p * (1 - dtf) * (tf < 5)) + np.sum(p * dtf * (tf + 1 < 5))
        assert abs(float(g[i].astype(np.float64).sum()) - expect) <= 2e-2, i
    for algo in (1, 2):
        ga = _batch(evs[:8], 5, (224, 224), algo=algo)
        assert np.abs(ga - g[:8]).max() <= 2e-5, algo


def test_sorted_check_kernel():
    from eventpretrain_amd._lib import call, ptr, stream_ptr
    from eventpretrain_amd.testing import synthetic_events
    a, b = synthetic_events(1, 500), synthetic_events(2, 700)
    b[[10, 400]] = b[[400, 10]]
    ev = torch.from_numpy(np.concatenate([a, b])).cuda()
    off = torch.tensor([0, 500, 1200], dtype=torch.int64).cuda()
    flags = torch.empty(2, dtype=torch.int32).cuda()
    call("evp_events_sorted_check", ptr(ev), ptr(off), 2, 0, ptr(flags), stream_ptr())
    assert flags.cpu().tolist() == [1, 0]


# --------------------------------------------------------------------------------------------------- masking
def test_mask_ids_bit_exact_vs_reference():
    from eventpretrain_amd import ops
    d = load_golden("masking")
    for c in jl(d["cases"]):
        t = c["tag"]
        keep, mask, restore = ops.mask_from_noise(torch.from_numpy(d[t + "_noise"]).cuda(), c["ratio"])
        assert keep.dtype == torch.int64 and restore.dtype == torch.int64 and mask.dtype == torch.float32
        assert np.array_equal(keep.cpu().numpy(), d[t + "_ids_keep"])
        assert np.array_equal(restore.cpu().numpy(), d[t + "_ids_restore"])
        assert np.array_equal(mask.cpu().numpy(), d[t + "_mask"])


def test_mask_ties_nan_and_sizes():
    from eventpretrain_amd import ops
    from oracle.model_oracle import masking_from_noise
    g = torch.Generator().manual_seed(0)
    for B, L, r in [(3, 49, 0.75), (2, 1024, 0.5), (5, 196, 0.0), (1, 7, 0.9), (64, 196, 0.5)]:
        noise = torch.rand(B, L, generator=g)
        noise[:, ::3] = noise[:, :1]            # massive ties, as density masking produces (empty patches)
        if L > 5:
            noise[0, 4] = float("nan")
        k, m, rs = ops.mask_from_noise(noise.cuda(), r)
        ko, mo_, ro = masking_from_noise(noise, r)
        assert torch.equal(k.cpu(), ko) and torch.equal(rs.cpu(), ro) and torch.equal(m.cpu(), mo_)


def test_density_noise_matches_oracle():
    from eventpretrain_amd import ops
    from oracle.model_oracle import density_noise
    x = torch.randn(3, 5, 64, 96, generator=torch.Generator().manual_seed(1))
    for strat, sign in (("density", 1.0), ("anti-density", -1.0)):
        got = ops.density_noise(x.cuda(), 16, sign).cpu()
        assert torch.allclose(got, density_noise(x, 16, strat), atol=1e-5, rtol=1e-5)


def test_view_augment_matches_reference_and_oracle():
    """evp_view_augment_f32 (crop box -> nearest resize -> h-flip -> time flip/negate) through the C-ABI: bit-exact against
    the reference's own evg_augment outputs (fixture) with the decisions drawn from RandomState(seed) in the reference's
    order, and against the oracle on a batch of random boxes incl. all four flip combinations and up/down-scaling."""
    from eventpretrain_amd.dataset.augmentation.view_augment import draw_evg_params, draw_evg_params_batch, evg_augment_batch
    from eventpretrain_amd.testing import det_normalish
    from helpers import jl
    from oracle import augment_oracle as ao
    d = load_golden("evg_augment")
    for tag in jl(d["tags"]):
        shp, size, seed = tuple(int(v) for v in d[f"{tag}_shape"]), tuple(int(v) for v in d[f"{tag}_size"]), int(d[f"{tag}_seed"])
        v = det_normalish(f"aug.view.{tag}", shp)
        prm = draw_evg_params(np.random.RandomState(seed), shp[1], shp[2], 0.8)
        assert prm == ao.draw_evg_params(np.random.RandomState(seed), shp[1], shp[2], 0.8)
        out = evg_augment_batch(v.unsqueeze(0).cuda(), np.array([prm]), size).cpu().numpy()[0]
        if f"{tag}_out" in d.files:
            assert np.array_equal(out, d[f"{tag}_out"]), tag
        else:
            assert np.array_equal(out.reshape(-1)[::7], d[f"{tag}_sample"]), tag
    rng = np.random.default_rng(3)
    B, C, H, W = 16, 5, 120, 160
    x = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32))
    prm = np.zeros((B, 6), dtype=np.int32)
    for i in range(B):
        w, h = int(rng.integers(1, W + 1)), int(rng.integers(1, H + 1))
        prm[i] = (int(rng.integers(0, W - w + 1)), int(rng.integers(0, H - h + 1)), w, h, i & 1, (i >> 1) & 1)
    for size in ((224, 224), (64, 96)):
        out = evg_augment_batch(x.cuda(), prm, size).cpu().numpy()
        for i in range(B):
            assert np.array_equal(out[i], ao.evg_transform(x[i].numpy(), tuple(prm[i]), size, negate=True)), (size, i)
    # counter-based decisions: reproducible per (seed, step, sample), independent of batch composition
    p1 = draw_evg_params_batch(7, 3, 8, 224, 224)
    p2 = draw_evg_params_batch(7, 3, 4, 224, 224, first_sample=4)
    assert np.array_equal(p1[4:], p2) and not np.array_equal(p1, draw_evg_params_batch(7, 4, 8, 224, 224))
    assert ((p1[:, 0] + p1[:, 2] <= 224) & (p1[:, 1] + p1[:, 3] <= 224)).all()
    with pytest.raises(ValueError):
        evg_augment_batch(x.cuda(), np.array([[0, 0, W + 1, H, 0, 0]] * B), (8, 8))


def test_events_augment_matches_reference_and_oracle():
    """Event-level augmentation through the C-ABI (evp_events_erase_add_f64) + the rescale fused into K1
    (evp_voxel_scatter_scaled_f32): the drop-in `events_augment(args, events, size, seed)` reproduces the reference's
    augmented clip bit for bit (integer/index work + float64 adds: exact), the voxel grid of the rescaled clip matches
    the reference's within the K1 tolerance, a ragged batch matches the oracle."""
    from types import SimpleNamespace
    from eventpretrain_amd.dataset.augmentation import events_augment as ea
    from eventpretrain_amd.dataset.dataset_utils.events_to_voxel_grid import voxel_grid_batch
    from oracle import augment_oracle as ao
    d = load_golden("events_augment")
    for tag in jl(d["tags"]):
        seed, n, sh, sw, S = (int(v) for v in d[tag + "_meta"])
        out = ea.events_augment(SimpleNamespace(), d[tag + "_events_in"].copy(), size=(sh, sw), seed=seed)
        assert out.is_cuda and np.array_equal(out.cpu().numpy(), d[tag + "_events_out"]), tag
        off = torch.tensor([0, out.shape[0]], dtype=torch.int64, device="cuda")
        g = voxel_grid_batch(out, off, 5, (S, S), scale=(S / sw, S / sh))[0].cpu().numpy()
        assert np.abs(g - d[tag + "_voxel"]).max() <= 1e-5, tag
        # the drop-in events_reshape on the device tensor, then plain K1: the same grid
        g2 = voxel_grid_batch(ea.events_reshape(out.clone(), sw, sh, S, S), off, 5, (S, S))[0].cpu().numpy()
        assert np.abs(g2 - d[tag + "_voxel"]).max() <= 1e-5, tag

    # ragged batch with its own decision streams, clips with nothing erased/added and an empty clip, against the oracle
    from eventpretrain_amd.testing import synthetic_events
    sizes = [20000, 150, 0, 99, 5000, 100000]
    evs = [synthetic_events(300 + i, n, width=346, height=260) if n else np.zeros((0, 4)) for i, n in enumerate(sizes)]
    decs = [ao.draw_erase_add(np.random.RandomState(70 + i), n) for i, n in enumerate(sizes)]
    decs[4] = None
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    out, out_off = ea.events_augment_batch(torch.from_numpy(np.concatenate(evs)).cuda(), offs, decs, (260, 346))
    out, out_off = out.cpu().numpy(), out_off.cpu().numpy()
    for i, (e, dec) in enumerate(zip(evs, decs)):
        ref = ao.erase_add_apply(e, dec, (260, 346))
        assert np.array_equal(out[out_off[i]:out_off[i + 1]], ref), i
    assert out_off[-1] == out.shape[0]

    # counter-based decision stream: reproducible per (seed, step, sample), valid tables, same result as the oracle merge
    decs2 = ea.draw_erase_add_batch(9, 3, sizes)
    again = ea.draw_erase_add_batch(9, 3, sizes[3:], first_sample=3)
    assert decs2[2] is None and decs2[3] is None and np.array_equal(decs2[5][0], again[2][0]) and np.array_equal(decs2[5][2], again[2][2])
    out2, off2 = ea.events_augment_batch(torch.from_numpy(np.concatenate(evs)).cuda(), offs, decs2, (260, 346))
    out2, off2 = out2.cpu().numpy(), off2.cpu().numpy()
    for i, (e, dec) in enumerate(zip(evs, decs2)):
        assert np.array_equal(out2[off2[i]:off2[i + 1]], ao.erase_add_apply(e, dec, (260, 346))), i

    # host-side argument errors surface as EvpError
    from eventpretrain_amd import EvpError
    with pytest.raises(EvpError):
        ea.events_augment_batch(torch.from_numpy(evs[0]).cuda(), [0, 20000], [(np.array([5, 5]), np.array([1]), np.zeros((1, 3)))], (260, 346))
    with pytest.raises(EvpError):
        ea.events_augment_batch(torch.from_numpy(evs[0]).cuda(), [0, 20000], [(np.array([5]), np.array([20000]), np.zeros((1, 3)))], (260, 346))
