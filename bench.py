#!/usr/bin/env python3
"""Headline benchmark: pretrain samples/s of the masked-ViT optimiser step (BASELINE.json configs[1]:
ViT-Base masked modeling with the difference-map decoder, 224x224 5-bin voxels, bf16, batch 64 per MI355X).

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

One "step" = mask-noise draw -> forward -> backward -> (N>1: bucketed RCCL gradient all-reduce, overlapped with
backward) -> fused AdamW -> zero_grad, on a synthetic batch that is already resident in HBM (voxel grids made by the
K1 kernel from synthetic event clips before the timed region). Prints ONE JSON line on rank 0.

Extra objects on the same line:
  roofline      dominant kernel = the forward / data-gradient bf16 MFMA GEMM family; achieved = algorithmic FLOPs of its launches in
                one step / the sum of their durations INSIDE the replayed step (in-kernel wall-clock stamps, StampedStep);
                `frac_warm` = the same launches re-launched back to back between HIP events; peak = 2.5 PFLOP/s dense bf16
                (MI355X_MICROARCH.md). The same object is emitted for every --config.
  voxel         K1 event->voxel scatter: achieved GB/s on 64 clips x 100k events (4.2 MB algorithmic bytes per clip)
                against the 8 TB/s HBM peak, HIP-event timed.
  cpu_baseline  the CPU oracle (oracle/model_oracle.py: torch fp32 restatement, kind "port") timed on this box's host
                cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0   # dense bf16, MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBS = 8000.0            # HBM3E spec, same table


def step_flops_per_sample(cfg):
    """Algorithmic FLOPs of one optimiser step per sample (SURVEY.md 8d conventions: multiply-add = 2, backward = 2x
    forward, softmax/LN/GELU/optimizer excluded). Patch-embed is counted on the kept tokens only, because this
    implementation gathers before the conv GEMM (identical result; the reference computes all 196)."""
    L, keep = cfg["L"], cfg["keep"]
    D, Dd = cfg["dim"], cfg["dec_dim"]
    blk = lambda n, d: 24 * n * d * d + 4 * n * n * d
    fwd = 2 * keep * cfg["patch_k"] * D + cfg["depth"] * blk(keep, D) + 2 * keep * D * Dd + cfg["dec_depth"] * blk(L, Dd) \
        + 2 * L * Dd * cfg["pred"]
    return 3 * fwd


CONFIGS = {
    # name: (workload label, backbone_type, model_size, pr_phase, hub factory, supp kind, noise cells, graphable)
    "vit_base_rec": ("ViT-Base masked modeling (diff-map decoder)", "vit", "base", "rec", "pretrain_hub_model_base_patch16", "frame", 196, True),
    "vit_base_con": ("ViT-Base contrastive stage (MoCo-v3 heads, CLIP tokens as input, queue 1024)", "vit", "base", "con",
                     "pretrain_hub_model_base_patch16", "clip", 0, True),
    "vit_base_adj": ("ViT-Base Trans stage (backbone frozen except norm_layer, MoCo-v3 heads, queue 1024)", "vit", "base", "adj",
                     "pretrain_hub_model_base_patch16", "clip", 0, True),
    "convvit_base_rec": ("ConvViT-Base masked modeling", "convvit", "base", "rec", "pretrain_hub_model_base_patch16", "frame", 196, True),
    "swin_tiny_rec": ("Swin-T masked modeling (window 7)", "swin", "tiny", "rec", "pretrain_hub_model_swin_tiny_patch16", "frame", 49, False),
    "swin_base_rec": ("Swin-Base masked modeling (window 7)", "swin", "base", "rec", "pretrain_hub_model_swin_base_patch16", "frame", 49, False),
}


def build(args, device):
    from eventpretrain_amd import ops
    from eventpretrain_amd.model.pretrain import pr_hub_model as hub
    from eventpretrain_amd.optim import FusedAdamW
    from eventpretrain_amd.testing import make_args
    from eventpretrain_amd.utils import lr_decay as lrd
    label, bb, size, phase, fac_name, supp, cells, graphable = CONFIGS[args.config]
    if args.config == "vit_base_rec" and args.model != "base":
        size = args.model
        fac_name = {"small": "pretrain_hub_model_small_patch16", "tiny": "pretrain_hub_model_tiny_patch16_64"}[args.model]
    a = make_args(model_size=size, pr_phase=phase, backbone_type=bb, device="cuda", batch_size=args.batch,
                  use_queue=True, mask_ratio=0.5 if phase == "rec" else 0.0)
    torch.manual_seed(1234)       # identical initial weights on every rank (DDP broadcasts them in the reference)
    model = getattr(hub, fac_name)(a, emb_frames_dim=512, queue_length=1024, T=0.07).to(device).train()
    if phase == "adj":            # main_pretrain.py:281-284
        for k, v in model.backbone.named_parameters():
            if "norm_layer" not in k:
                v.requires_grad = False
    world = dist.get_world_size() if dist.is_initialized() else 1
    a.distributed = world > 1        # queue policy "all_gather": the contrastive keys of every rank enter every queue
    a.lr = a.blr * args.batch * world / 256
    groups = lrd.param_groups_lrd(a, model, a.weight_decay, layer_decay=1)
    opt = FusedAdamW(groups, lr=a.lr, betas=(0.9, 0.95), grad_scale=1.0 / world)
    ops.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    return a, model, opt


def make_batch(args, device, rank):
    """Synthetic clips -> voxel grids through the K1 kernel; difference-map targets ~ N(0,1)."""
    from eventpretrain_amd.dataset.dataset_utils.events_to_voxel_grid import voxel_grid_batch
    from eventpretrain_amd.testing import synthetic_events
    S = 64 if args.model == "tiny" else 224
    n_ev = 10_000 if args.model == "tiny" else 100_000
    B = args.batch
    base = synthetic_events(1000 * rank, n_ev, width=S, height=S)
    rng = np.random.default_rng(77 + rank)
    evs = []
    for i in range(B):             # cheap per-clip variation of one generated clip (generation is host-side numpy)
        e = base.copy()
        e[:, 0] = (e[:, 0] + rng.integers(0, S)) % S
        e[:, 1] = (e[:, 1] + rng.integers(0, S)) % S
        if i % 2:
            e[:, 3] = 1.0 - e[:, 3]
        evs.append(e)
    ev = torch.from_numpy(np.concatenate(evs, 0)).to(device)
    off = torch.arange(0, (B + 1) * n_ev, n_ev, dtype=torch.int64, device=device)
    vox = voxel_grid_batch(ev, off, 5, (S, S))
    g = torch.Generator(device="cpu").manual_seed(1 + rank)
    if CONFIGS[args.config][5] == "clip":
        tgt = torch.randn(B, 197, 512, generator=g).to(device)      # frozen-CLIP image tokens: an input tensor
    else:
        tgt = torch.randn(B, 1, S, S, generator=g).to(device)
    return ev, off, vox, tgt, S, n_ev


class GemmTimer:
    """Kernel-level HIP-event timing of the GEMM family inside bench.py. One eager step is run with ops.gemm wrapped to
    RECORD every call (arguments and operand tensors); every distinct call signature is then re-launched `reps` times
    between two HIP events on the launch stream (torch's current stream), so the figure is a pure kernel duration
    (no event-per-launch overhead, same shapes / layouts / epilogues / cache-resident operands as the step). Results
    are aggregated per kernel instantiation, i.e. per rocprofv3 kernel name."""

    def __init__(self):
        self.calls = []

    def install(self):
        from eventpretrain_amd import ops
        self._orig = ops.gemm
        timer = self

        def rec(a, b, out, **kw):
            timer.calls.append((a, b, out, kw))
            return timer._orig(a, b, out, **kw)

        ops.gemm = rec

    def remove(self):
        from eventpretrain_amd import ops
        ops.gemm = self._orig

    @staticmethod
    def _inst(a, out, kw):
        """Kernel instantiation a call lands on = the launcher's own rule (gemm.hip pick_tile, gemm_g4.hip evp_g4_gemm_pick)."""
        M, N, K = kw["M"], kw["N"], kw["K"]
        nb = kw.get("batch", (1, 1))
        act = kw.get("act", 0)
        epi = 1 if act in (1, 3) else 2 if act in (2, 4) else 0
        bf = a.dtype == torch.bfloat16
        names = ("bf16" if bf else "f32", "f32" if out.dtype == torch.float32 else "bf16", epi, int(bool(kw.get("trans_a"))), int(bool(kw.get("trans_b"))))
        tile = kw.get("tile", 0)
        g4_ok = (bf and out.dtype == torch.bfloat16 and not kw.get("trans_a") and nb == (1, 1) and K % 32 == 0 and K >= 96 and N % 8 == 0
                 and kw.get("residual") is None and not kw.get("accumulate") and not (epi == 1 and kw.get("trans_b")) and not (epi == 2 and not kw.get("trans_b")))
        if tile == 0 and g4_ok and N >= 1024 and M >= 2048 and GemmTimer.g4_fwd and epi != 1:
            one_round = ((M + 255) // 256) * ((N + 255) // 256) <= 256
            tile = 20 if one_round else (22 if kw.get("trans_b") else 0)
        if tile in (20, 21, 22):
            return ("g4x",) + names + ({20: "256x256", 21: "256x128", 22: "128x256"}[tile],)
        t128 = ((M + 127) // 128) * ((N + 127) // 128) * nb[0] * nb[1]
        splittable = bool(kw.get("trans_a")) and out.dtype == torch.float32 and kw.get("bias") is None and act == 0 and kw.get("residual") is None \
            and kw.get("aux") is None and nb == (1, 1) and K >= 2048
        tile = tile or (1 if (M >= 128 and N >= 128 and (t128 >= 192 or splittable)) else 2)
        t96 = ((M + 95) // 96) * ((N + 127) // 128)
        if not kw.get("tile", 0) and tile == 1 and bf and not kw.get("trans_a") and nb == (1, 1) and t128 > 256 and t96 <= 512:
            tile = 4
        return ("gemm",) + names + ({1: "128x128", 2: "64x64", 4: "96x128"}.get(tile, str(tile)),)

    g4_fwd = False          # evp_gemm_set_variant(11) is not the default (gemm.hip g_gemm_g4_fwd)

    @staticmethod
    def kernel_name(inst):
        return "%s_kernel<in=%s,out=%s,epi=%d,transA=%d,transB=%d,tile=%s>" % inst

    def summary(self, reps=10):
        sigs = {}
        for a, b, out, kw in self.calls:
            key = (self._inst(a, out, kw), kw["M"], kw["N"], kw["K"], kw.get("batch", (1, 1)), kw.get("bias") is not None,
                   kw.get("residual") is not None, kw.get("aux") is not None, bool(kw.get("accumulate")))
            ent = sigs.setdefault(key, [0, (a, b, out, kw)])
            ent[0] += 1
        agg = {}
        for key, (count, (a, b, out, kw)) in sigs.items():
            for _ in range(2):
                self._orig(a, b, out, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                self._orig(a, b, out, **kw)
            e1.record()
            torch.cuda.synchronize()
            sec = e0.elapsed_time(e1) * 1e-3 / reps
            nb = key[4]
            fl = 2.0 * key[1] * key[2] * key[3] * nb[0] * nb[1]
            g = agg.setdefault(key[0], [0.0, 0.0, 0])
            g[0] += fl * count
            g[1] += sec * count
            g[2] += count
        out = []
        for inst, (fl, sec, n) in agg.items():
            out.append(dict(kernel=GemmTimer.kernel_name(inst), launches_per_step=n,
                            avg_us=sec / n * 1e6, tflops=fl / sec / 1e12, ms_per_step=sec * 1e3, flops_per_step=fl))
        out.sort(key=lambda d: -d["ms_per_step"])
        return out


class StampedStep:
    """IN-STEP kernel durations of every GEMM launch of one optimiser step, measured where the launches run: inside a
    replayed HIP graph, operands just produced by the previous kernel. A second step executor is captured with the
    library's in-kernel stamps on (evp_gemm_set_stamp_buffer: every GEMM launch gets a slot, each of its workgroups writes
    the chip-wide 100 MHz counter when it starts and after its last store was acknowledged); ops.gemm and the grouped
    weight-gradient entries are wrapped during that capture to record WHAT each stamped launch computes. The graph is then
    replayed and max(end) - min(start) per slot read back. HIP events cannot do this (they would need a node between any
    two kernels of the graph); the warm re-launch figure of GemmTimer is printed beside it as `frac_warm`."""
    STRIDE = 2 * 4096        # uint64 per slot (EVP_STAMP_WGS workgroups x {start, end}), csrc/gemm_common.h
    GROUPED = {"evp_gemm_grouped_tn_bf16": ("gemm_grouped_tn_kernel<128,128>", 128),
               "evp_gemm_grouped_tn_g4_bf16": ("gemm_g4_grouped_tn_kernel (256x256, one wave per SIMD, 32x32x16)", 256)}

    def __init__(self, make_executor, device, n_slots=2048):
        from eventpretrain_amd import ops as _ops
        from eventpretrain_amd._lib import call
        self.buf = torch.zeros(n_slots * self.STRIDE, dtype=torch.int64, device=device)
        self.n_slots = n_slots
        self.launches = []                      # (slot, kernel name, flops) of the launches made while the graph was captured
        pending_grouped = {}
        orig_gemm, orig_call, orig_flush = _ops.gemm, _ops.call, _ops._deferred.flush

        def rec_gemm(a, b, out, **kw):
            capturing = torch.cuda.is_current_stream_capturing()
            idx = call("evp_gemm_stamp_count")
            r = orig_gemm(a, b, out, **kw)
            if capturing and call("evp_gemm_stamp_count") == idx + 1:
                nb = kw.get("batch", (1, 1))
                self.launches.append((idx % n_slots, GemmTimer.kernel_name(GemmTimer._inst(a, out, kw)), 2.0 * kw["M"] * kw["N"] * kw["K"] * nb[0] * nb[1]))
            return r

        def rec_call(name, *a_):
            if name not in self.GROUPED:
                return orig_call(name, *a_)
            capturing = torch.cuda.is_current_stream_capturing()
            idx = call("evp_gemm_stamp_count")
            r = orig_call(name, *a_)
            fl, tiles = pending_grouped.pop(name, (0.0, 0))
            if capturing and call("evp_gemm_stamp_count") == idx + 1:
                self.launches.append((idx % n_slots, "%s (weight gradients of the step, %d tiles)" % (self.GROUPED[name][0], tiles), fl))
            return r

        def rec_flush():
            for (_, _, _, n_out, k_in, rows, _bp) in _ops._deferred.w:      # the routing rule of _DeferredGrads._build
                big = _ops._use_wgrad_g4 and rows % 32 == 0 and rows >= 96 and n_out >= 256 and k_in >= 256
                name = "evp_gemm_grouped_tn_g4_bf16" if big else "evp_gemm_grouped_tn_bf16"
                T_ = self.GROUPED[name][1]
                fl, tiles = pending_grouped.get(name, (0.0, 0))
                pending_grouped[name] = (fl + 2.0 * n_out * k_in * rows, tiles + ((n_out + T_ - 1) // T_) * ((k_in + T_ - 1) // T_))
            return orig_flush()

        call("evp_gemm_set_stamp_buffer", self.buf.data_ptr(), n_slots)
        _ops.gemm, _ops.call, _ops._deferred.flush = rec_gemm, rec_call, rec_flush
        try:
            self.executor = make_executor()
        finally:
            _ops.gemm, _ops.call, _ops._deferred.flush = orig_gemm, orig_call, orig_flush
            call("evp_gemm_set_stamp_buffer", None, 0)       # later launches are not stamped; the captured ones keep their slots
        self.ok = self.executor.graph is not None and len(self.launches) > 0 and len({l[0] for l in self.launches}) == len(self.launches)

    def measure(self, replays=6):
        """-> per-launch mean duration in seconds (list aligned with self.launches), averaged over `replays` replays."""
        slots = torch.tensor([l[0] for l in self.launches], dtype=torch.int64, device=self.buf.device)
        view = self.buf.view(self.n_slots, self.STRIDE)
        acc = torch.zeros(len(self.launches), dtype=torch.float64)
        for _ in range(2):
            self.executor.step()
        n = 0
        for _ in range(replays):
            self.buf.zero_()
            self.executor.step()
            torch.cuda.synchronize()
            st = view.index_select(0, slots).cpu().numpy().view(np.uint64).reshape(len(self.launches), -1, 2)
            start = ~st[:, :, 0]                     # stored complemented: 0 = not written
            have = st[:, :, 0] != 0
            t0 = np.where(have, start, np.uint64(0xFFFFFFFFFFFFFFFF)).min(axis=1)
            t1 = st[:, :, 1].max(axis=1)
            if not (have.any(axis=1).all() and (t1 > t0).all()):
                continue
            acc += torch.from_numpy((t1 - t0).astype(np.float64) * 1e-8)       # 100 MHz ticks
            n += 1
        return (acc / max(n, 1)).tolist() if n else None

    def summary(self, replays=6):
        durs = self.measure(replays)
        if durs is None:
            return None
        agg = {}
        for (slot, name, fl), sec in zip(self.launches, durs):
            g = agg.setdefault(name, [0.0, 0.0, 0])
            g[0] += fl
            g[1] += sec
            g[2] += 1
        out = [dict(kernel=k, launches_per_step=n, avg_us=sec / n * 1e6, tflops=fl / sec / 1e12, ms_per_step=sec * 1e3, flops_per_step=fl)
               for k, (fl, sec, n) in agg.items()]
        out.sort(key=lambda d: -d["ms_per_step"])
        return out


def cpu_baseline(cfgs, n_threads, Bc=16, n_steps=6):
    """Oracle (torch-CPU fp32 restatement + its own AdamW) on a bounded sample: ViT-Base, Bc samples x n_steps optimiser steps."""
    from oracle import model_oracle as mo
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import rec_state_dict
    torch.set_num_threads(n_threads)
    cfg = dict(input=224, patch=16, dim=768, depth=12, heads=12, dec_dim=512, dec_depth=8, dec_heads=16, mask_ratio=0.5, B=Bc)
    sd = rec_state_dict(cfg)
    train = [k for k in sd if "pos_embed" not in k]
    for k in train:
        sd[k].requires_grad_(True)
    g = torch.Generator().manual_seed(0)
    x, y, noise = torch.randn(Bc, 5, 224, 224, generator=g), torch.randn(Bc, 1, 224, 224, generator=g), torch.rand(Bc, 196, generator=g)
    decay, _ = mo.decay_split([(k, tuple(sd[k].shape)) for k in train])
    m = {k: torch.zeros_like(sd[k]) for k in train}
    v = {k: torch.zeros_like(sd[k]) for k in train}
    t0 = time.time()
    for it in range(n_steps):
        for k in train:
            sd[k] = sd[k].detach().requires_grad_(True)
        loss = mo.rec_step(sd, x, y, noise, cfg)[0]
        loss.backward()
        with torch.no_grad():
            for k in train:
                sd[k], m[k], v[k] = mo.adamw_step(sd[k], sd[k].grad, m[k], v[k], it + 1, 1e-4, 0.05 if k in decay else 0.0)
    dt_ = time.time() - t0
    return dict(value=Bc * n_steps / dt_, unit="samples/s", cores=n_threads, kind="port",
                sample="oracle/model_oracle.py (torch fp32 CPU restatement), ViT-Base+dec-Base masked step fwd+bwd+AdamW, "
                       "B=%d, %d steps, %.1f s" % (Bc, n_steps, dt_))


_PMC_CONFIG_OK = True      # the committed counters are of the headline workload: other --config runs report traffic null


def pmc_traffic(kernel_key):
    """(HBM-side bytes per launch, source file) of a kernel from the newest committed PMC summary under profiles/ (separate rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE passes of THIS command with the guide's gfx950 correction, tools/make_profiles.py), or (None, None).
    bench.py cannot collect PMC counters itself; the file the figure comes from is named next to it (`traffic_source`)."""
    if not _PMC_CONFIG_OK and "voxel" not in kernel_key:
        return None, None
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    for fn in ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            with open(os.path.join(here, fn)) as f:
                k = json.load(f)["kernels"]
            for name, v in k.items():
                if name in kernel_key:
                    return float(v["traffic_bytes"]), "profiles/" + fn
        except Exception:
            pass
    return None, None


def cpu_voxel_baseline():
    from eventpretrain_amd.testing import synthetic_events
    from oracle.voxel_oracle import voxel_grid_batch
    evs = [synthetic_events(i) for i in range(8)]
    ev = np.concatenate(evs, 0)
    off = np.arange(0, 9 * 100_000, 100_000, dtype=np.int64)
    voxel_grid_batch(ev[:100_000], off[:2], 5, (224, 224))
    reps, t0 = 0, time.time()
    while time.time() - t0 < 3.0:          # a bounded sample of the same workload: ~3 s of scalar C
        voxel_grid_batch(ev, off, 5, (224, 224))
        reps += 1
    dt_ = time.time() - t0
    return dict(value=8 * reps / dt_, unit="clips/s", cores=1, kind="port",
                sample="oracle/voxel_oracle.c, %d x (8 clips x 100k events), %.2f s" % (reps, dt_))


def self_launch(n, argv):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same args>
    as a child process, let rank 0's JSON line through on stdout and return the child's exit code. Nothing here initialises the
    GPU: the parent only counts devices (EVP_BENCH_SHARE_DEVICE=1 skips the count: gloo rehearsals with the ranks on one card)."""
    import socket
    import subprocess
    if not os.environ.get("EVP_BENCH_SHARE_DEVICE"):
        n_dev = torch.cuda.device_count()
        if n_dev < n:
            print("bench.py: --gpus %d needs %d HIP devices on this node, %d visible" % (n, n, n_dev), file=sys.stderr)
            return 2
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    print("bench.py: launching %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (SURVEY.md 8d: 20 warm-up + 100 timed when not overridden)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="vit_base_rec", choices=sorted(CONFIGS),
                    help="vit_base_rec = the headline workload (BASELINE.json configs[1]); the others are configs[2-4] on one GPU")
    ap.add_argument("--batch", type=int, default=64, help="per-GPU batch")
    ap.add_argument("--model", default="base", choices=["base", "small", "tiny"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--bucket-mb", type=float, default=64.0)
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from Python each step (no HIP graph)")
    ap.add_argument("--wgrad-chunks", type=int, default=4,
                    help="N > 1: weight-gradient launches per step (each chunk's buffer is all-reduced while the next computes)")
    ap.add_argument("--gemm-variant", type=int, action="append", default=[],
                    help="A/B aid: evp_gemm_set_variant codes applied before the model is built (11/12/13: G4 forward / data-gradient routing)")
    ap.add_argument("--force-dist", action="store_true",
                    help="take the multi-GPU code path (process group, reducer, split graphs) even with one rank")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the product path); gloo = rehearsal of the N > 1 control flow with several ranks on ONE card")
    args = ap.parse_args()

    # ---- N > 1 from a plain `python bench.py --gpus N`: start the ranks as a CHILD process group (one rank per GPU over RCCL) and
    # relay. This runs before anything touches the GPU runtime: no HIP call, no torch.cuda.is_available(), no library load
    # (torch.cuda.device_count() does not initialise the device on this image); never os.exec*.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    from eventpretrain_amd import _lib
    _lib.require_device()
    for v_ in args.gemm_variant:
        _lib.call("evp_gemm_set_variant", v_)
    GemmTimer.g4_fwd = 11 in args.gemm_variant
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    if args.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    n_dev = torch.cuda.device_count()
    if local >= n_dev and not os.environ.get("EVP_BENCH_SHARE_DEVICE"):
        raise SystemExit("rank %d: LOCAL_RANK %d but only %d HIP device(s) visible" % (rank, local, n_dev))
    local_dev = local % max(n_dev, 1)            # EVP_BENCH_SHARE_DEVICE=1: several ranks on one card (gloo rehearsal only)
    torch.cuda.set_device(local_dev)
    device = torch.device("cuda", local_dev)
    multi = world > 1 or args.force_dist
    rccl_world = 1
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world)     # "nccl" is RCCL over xGMI on ROCm
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        rccl_world = dist.get_world_size()
        if rccl_world != args.gpus:
            raise SystemExit("process group has %d ranks, --gpus %d" % (rccl_world, args.gpus))

    a, model, opt = build(args, device)
    ev, off, vox, tgt, S, n_ev = make_batch(args, device, rank)
    reducer = None
    if multi:
        from eventpretrain_amd.parallel import BucketedGradReducer
        reducer = BucketedGradReducer([p for p in model.parameters() if p.requires_grad], bucket_mb=args.bucket_mb)
    gen = torch.Generator(device=device).manual_seed(100 + rank)     # seed + rank, as main_pretrain.py:174
    L = model.backbone.num_patches
    label, _bb, _size, phase, _fac, _supp, cells, graphable = CONFIGS[args.config]
    global _PMC_CONFIG_OK
    _PMC_CONFIG_OK = args.config == "vit_base_rec" and args.batch == 64 and args.dtype == "bf16"

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- the step executor (eventpretrain_amd/engine.py): the ~600 kernel launches of forward + backward (+ AdamW) are
    # captured once in HIP graphs and replayed; per-step scalars (lr, Adam bias corrections) travel through pinned host
    # tables that the graph's own H2D copy nodes re-read, the mask noise is drawn into a static buffer before each
    # replay. With N > 1 the collectives stay outside the graphs: [forward+backward graph] -> in-place RCCL all-reduce of
    # the flat gradient buffers -> [AdamW graph].
    from eventpretrain_amd.engine import GraphedStep
    is_swin = not graphable      # CONFIGS' flag: the Swin step's launch geometry follows the mask pattern
    use_graph = not args.no_graph
    n_warm_eager = min(args.warmup, 3) if use_graph else 0
    step_prepare = None
    if phase == "rec":
        fwd, noise_shape = (lambda m, x, y, noise: m(x, y, is_rec=True, noise=noise)), (args.batch, cells)
        if is_swin:
            # Swin: the window plan is host work per pattern. The noise is drawn on the HOST (seeded per rank), the backbone's
            # fixed-shape plan tables are refreshed by one H2D copy before each replay (SwinTransformer.enable_static_plan), so
            # ONE captured graph serves every pattern; a pattern that overflows the fixed group count runs that step eagerly --
            # with N > 1 on every rank at once (engine.GraphedStep._vote), in the data-parallel eager form.
            step_prepare = model.backbone.enable_static_plan(device)
    else:
        fwd, noise_shape = (lambda m, x, y, noise: m(x, y)), None
    executor = GraphedStep(model, opt, fwd, [vox, tgt], noise_shape=noise_shape, generator=gen, reducer=reducer, use_graph=use_graph,
                           warmup=max(n_warm_eager, 2), wgrad_chunks=args.wgrad_chunks, step_prepare=step_prepare,
                           host_generator=torch.Generator().manual_seed(100 + rank))
    graph_note = executor.note
    step, eager_step = executor.step, executor.eager_step
    loss = None

    for _ in range(args.warmup - n_warm_eager):
        loss = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss.item())

    ms = elapsed / args.steps * 1e3
    value = args.batch * world * args.steps / elapsed
    dims = dict(base=(768, 12, 512, 8), small=(384, 12, 256, 8), tiny=(192, 12, 128, 4))[args.model]
    fcfg = dict(L=196 if S == 224 else 16, keep=98 if S == 224 else 8, dim=dims[0], depth=dims[1], dec_dim=dims[2], dec_depth=dims[3],
                patch_k=5 * 256, pred=256)
    headline = args.config == "vit_base_rec"
    fl_sample = step_flops_per_sample(fcfg) if headline else None
    workload = label if not headline else "ViT-%s masked modeling (diff-map decoder)" % args.model.capitalize()
    result = {
        "metric": "pretrain samples/sec (masked-ViT step, B=64 224^2)", "value": value, "unit": "samples/s",
        "n_gpus": world, "rccl_world": rccl_world, "dist_backend": (args.dist_backend if multi else None), "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "%s, %dx%d 5-bin voxels, batch=%d per GPU, AdamW step included" % (workload, S, S, args.batch),
                   "name": args.config, "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                   "mask_ratio": 0.5 if phase == "rec" else 0.0,
                   "params_M": sum(p.numel() for p in model.parameters()) / 1e6},
        "final_loss": final_loss, "launch_mode": graph_note, "eager_fallback_steps": executor.eager_fallbacks,
    }
    if headline:
        result["step_tflops_per_gpu"] = fl_sample * args.batch / (ms * 1e-3) / 1e12
        result["step_mfma_frac"] = result["step_tflops_per_gpu"] / MFMA_BF16_PEAK_TFLOPS

    # ---- median of per-step times (HIP events around single steps, after the timed region; SURVEY.md 8d asks for the
    # median; `value` above stays the mean over exactly K steps as the driver's contract says)
    n_med = min(args.steps, 50)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n_med + 1)]
    barrier()
    evs[0].record()
    for i in range(n_med):
        step()
        evs[i + 1].record()
    barrier()
    per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(n_med))
    result["ms_per_step_median"] = per[n_med // 2]
    result["ms_per_step_min"] = per[0]

    if multi and getattr(executor, "plan", None) is not None:
        # per-rank tail of the data-parallel step: from "last weight-gradient chunk computed" to "every all-reduce and update
        # part done" on the step's stream -- what is NOT hidden behind compute (diagnoses the 1 -> 8 scaling from one run)
        executor.plan.timing = []
        for _ in range(10):
            step()
        barrier()
        tail = [a_.elapsed_time(b_) for a_, b_ in executor.plan.timing]
        executor.plan.timing = None
        t_ = torch.tensor([sum(tail) / max(len(tail), 1)], dtype=torch.float64, device=device)
        gathered = [torch.zeros_like(t_) for _ in range(world)]
        dist.all_gather(gathered, t_)
        result["dp_tail_ms_per_rank"] = [round(float(g_.item()), 4) for g_ in gathered]

    if rank == 0 and not args.no_kernel_timing:
        # ---- dominant kernel, HIP events around every GEMM launch of real steps (instrumented, after the timed region)
        from eventpretrain_amd import ops as _ops
        timer = GemmTimer()
        opt.zero_grad(set_to_none=True)
        # the grouped weight-gradient launch is one kernel per step: time it directly with HIP events around the call
        grouped = {}
        orig_call = _ops.call

        GROUPED = {"evp_gemm_grouped_tn_bf16": ("gemm_grouped_tn_kernel<128,128>", 128),
                   "evp_gemm_grouped_tn_g4_bf16": ("gemm_g4_grouped_tn_kernel (256x256, one wave per SIMD, 32x32x16)", 256)}

        def timed_call(name, *a_):
            if name not in GROUPED:
                return orig_call(name, *a_)
            # One kernel per step, milliseconds long: SINGLE launches, each between its own pair of events, median of 7 (its tables
            # are still alive here; the step's gradients are not used after this instrumented pass, so re-accumulating into them is
            # harmless). Ten launches back to back -- the form used for the short GEMMs -- read 20 % long for this MFMA-dense
            # kernel (1.97 ms against 1.65 ms inside the step by rocprofv3): the chip clocks down under the sustained load.
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            r = orig_call(name, *a_)                      # the launch in its place in the step (operands just produced)
            c1.record()
            torch.cuda.synchronize()
            grouped.setdefault(name, {})["ctx_ms"] = c0.elapsed_time(c1)
            singles = []
            for _ in range(7):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                orig_call(name, *a_)
                e1.record()
                torch.cuda.synchronize()
                singles.append(e0.elapsed_time(e1))
            grouped.setdefault(name, {})["ms"] = sorted(singles)[len(singles) // 2]
            return r

        orig_flush = _ops._deferred.flush

        def counting_flush():
            for (_, _, _, n_out, k_in, rows, _bp) in _ops._deferred.w:
                big = _ops._use_wgrad_g4 and rows % 32 == 0 and rows >= 96 and n_out >= 256 and k_in >= 256      # the routing rule of flush()
                name = "evp_gemm_grouped_tn_g4_bf16" if big else "evp_gemm_grouped_tn_bf16"
                T_ = GROUPED[name][1]
                g_ = grouped.setdefault(name, {})
                g_["flops"] = g_.get("flops", 0.0) + 2.0 * n_out * k_in * rows
                g_["tiles"] = g_.get("tiles", 0) + ((n_out + T_ - 1) // T_) * ((k_in + T_ - 1) // T_)
            return orig_flush()

        attn_flops = [0.0]
        orig_af, orig_ab = _ops.attention_fused_fwd, _ops.attention_fused_bwd

        def af(qkv, B_, N_, heads_, dh_, **kw):
            attn_flops[0] += 4.0 * B_ * heads_ * N_ * N_ * dh_
            return orig_af(qkv, B_, N_, heads_, dh_, **kw)

        def ab(qkv, out_, dout_, lse_, B_, N_, heads_, dh_):
            attn_flops[0] += 8.0 * B_ * heads_ * N_ * N_ * dh_       # backward = 2x forward (SURVEY.md 8d convention)
            return orig_ab(qkv, out_, dout_, lse_, B_, N_, heads_, dh_)

        _ops.attention_fused_fwd, _ops.attention_fused_bwd = af, ab
        _ops.call, _ops._deferred.flush = timed_call, counting_flush
        timer.install()
        saved_reducer, executor.reducer = executor.reducer, None      # rank-0-only pass: no collective may run in it
        eager_step()
        executor.reducer = saved_reducer
        timer.remove()
        _ops.call, _ops._deferred.flush = orig_call, orig_flush
        _ops.attention_fused_fwd, _ops.attention_fused_bwd = orig_af, orig_ab
        ks = timer.summary()
        timer.calls = []
        torch.cuda.synchronize()
        for name, g_ in grouped.items():
            if "ms" not in g_:
                continue
            # the launch in its place in the (instrumented, eager) step is the figure; the re-launch median is kept beside it
            sec = (g_.get("ctx_ms") or g_["ms"]) * 1e-3
            ks.append(dict(kernel="%s (weight gradients of the step, %d tiles)" % (GROUPED[name][0], g_["tiles"]),
                           relaunch_median_us=g_["ms"] * 1e3, launches_per_step=1, avg_us=sec * 1e6, tflops=g_["flops"] / sec / 1e12, ms_per_step=sec * 1e3,
                           flops_per_step=g_["flops"]))
        ks.sort(key=lambda d: -d["ms_per_step"])
        # ---- the same launches timed IN the step (in-kernel stamps inside a replayed graph); see StampedStep
        ks_step = None
        if use_graph and executor.graph is not None:
            try:
                st = StampedStep(lambda: GraphedStep(model, opt, fwd, [vox, tgt], noise_shape=noise_shape, generator=gen, reducer=None, use_graph=True,
                                                     warmup=2, step_prepare=step_prepare, host_generator=torch.Generator().manual_seed(100 + rank)), device)
                ks_step = st.summary() if st.ok else None
                del st
            except Exception as e:      # a measurement aid must not lose the bench line
                result["in_step_timing_error"] = repr(e)
        if ks:
            peak = MFMA_BF16_PEAK_TFLOPS if args.dtype == "bf16" else 157.3
            # dominant kernel = the FAMILY with the most time: the forward / data-gradient GEMMs (128x128 / 96x128 tiles of the
            # 16x16x32 body, 256x256 / 128x256 tiles of the G4 body), whose instantiations rocprofv3 lists under separate names
            is_fam = lambda d: d["kernel"].startswith(("gemm_kernel<", "g4x_kernel<"))
            is_grp = lambda d: "grouped" in d["kernel"]

            def fam_entry(ds_step, ds_warm, name, key):
                ds = ds_step if ds_step else ds_warm
                fl_, sec_, n_ = sum(d["flops_per_step"] for d in ds), sum(d["ms_per_step"] for d in ds) * 1e-3, sum(d["launches_per_step"] for d in ds)
                tr, src = pmc_traffic(key)
                e_ = {"bound": "mfma", "achieved": fl_ / sec_ / 1e12, "peak": peak, "unit": "TFLOP/s", "frac": fl_ / sec_ / 1e12 / peak,
                      "traffic": tr, "traffic_source": src, "kernel": name, "avg_launch_us": sec_ / n_ * 1e6, "launches_per_step": n_,
                      "ms_per_step_in_kernel": sec_ * 1e3}
                if ds_step:
                    e_["frac_in_step"] = e_["frac"]
                    e_["method"] = ("achieved = algorithmic 2MNK FLOPs of these launches in one step / the sum of their durations INSIDE the replayed "
                                    "step graph (in-kernel wall-clock stamps, max(end) - min(start) per launch, mean of 6 replays; bench.py StampedStep); "
                                    "frac_warm = the same launches re-launched back to back between two HIP events (operands cache-warm)")
                if ds_warm:
                    fw, sw = sum(d["flops_per_step"] for d in ds_warm), sum(d["ms_per_step"] for d in ds_warm) * 1e-3
                    e_["frac_warm"] = fw / sw / 1e12 / peak
                    e_["ms_per_step_in_kernel_warm"] = sw * 1e3
                return e_
            entries = []
            fam_w, grp_w = [d for d in ks if is_fam(d)], [d for d in ks if is_grp(d)]
            fam_s = [d for d in ks_step if is_fam(d)] if ks_step else None
            grp_s = [d for d in ks_step if is_grp(d)] if ks_step else None
            if fam_w or fam_s:
                entries.append(fam_entry(fam_s, fam_w, "gemm_kernel<...> + g4x_kernel<...> (forward + data-gradient GEMM family, %d instantiations)"
                                         % len(fam_s or fam_w), "gemm_kernel<"))
            if grp_w or grp_s:
                gs = sorted(grp_s, key=lambda d: -d["ms_per_step"])[:1] if grp_s else None
                gw = [d for d in grp_w if gs is None or d["kernel"].split(" (")[0] == gs[0]["kernel"].split(" (")[0]][:1] or grp_w[:1]
                entries.append(fam_entry(gs, gw, (gs or gw)[0]["kernel"], (gs or gw)[0]["kernel"]))
            entries.sort(key=lambda e_: -e_["ms_per_step_in_kernel"])
            if not entries:
                entries.append(fam_entry(None, ks[:1], ks[0]["kernel"], ks[0]["kernel"]))
            result["roofline"] = entries[0]
            if len(entries) > 1:
                result["roofline_second"] = entries[1]
            launched = sum(d["flops_per_step"] for d in ks) + attn_flops[0]
            result["step_tflops_launched"] = launched / (ms * 1e-3) / 1e12
            result["step_mfma_frac_launched"] = result["step_tflops_launched"] / peak
            if not headline:
                result["step_tflops_per_gpu"], result["step_mfma_frac"] = result["step_tflops_launched"], result["step_mfma_frac_launched"]
            rnd = lambda ds: [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items() if k != "flops_per_step"} for d in ds]
            result["gemm_kernels_in_step"] = rnd(ks_step) if ks_step else None
            result["gemm_kernels_warm"] = rnd(ks)
        # ---- K1 voxel scatter (HBM-bound)
        from eventpretrain_amd.dataset.dataset_utils.events_to_voxel_grid import voxel_grid_batch
        out = torch.empty_like(vox)
        for _ in range(3):
            voxel_grid_batch(ev, off, 5, (S, S), out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            voxel_grid_batch(ev, off, 5, (S, S), out=out)
        e1.record()
        torch.cuda.synchronize()
        sec = e0.elapsed_time(e1) * 1e-3 / reps
        bytes_ = args.batch * (n_ev * 32 + 5 * S * S * 4)
        result["voxel"] = {"bound": "hbm", "achieved": bytes_ / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": bytes_ / sec / 1e9 / HBM_PEAK_GBS,
                           "traffic": (pmc_traffic("voxel_bin_kernel")[0] or 0) + (pmc_traffic("voxel_cuts_kernel")[0] or 0) or None,
                           "traffic_source": pmc_traffic("voxel_bin_kernel")[1],
                           "kernel": "voxel_cuts_kernel+voxel_bin_kernel",
                           "us_per_batch": sec * 1e6, "clips_per_s": args.batch / sec, "events_per_s": args.batch * n_ev / sec,
                           "algorithmic_bytes_per_clip": n_ev * 32 + 5 * S * S * 4}
        # SURVEY.md 8(d) "end-to-end" line: voxelisation of the batch (K1, events resident in HBM) + the step, back to back
        result["end_to_end"] = {"value": args.batch * world / (ms * 1e-3 + sec), "unit": "samples/s",
                                "includes": "K1 voxel scatter of the batch (%.0f us) + optimiser step, serial, per GPU" % (sec * 1e6)}
        # ---- the loader's whole chain on the GPU: window pick -> erase / add -> rescale + K1 -> crop / resize /
        # flips (+ the target's bicubic frame augmentation), sensor-shaped clips (640 x 480) of 150 k events with a 100 k-event
        # window, decisions from the counter-based stream drawn on the host per batch (their host time is reported, it overlaps
        # the previous step in a real loop)
        try:
            from eventpretrain_amd.dataset.pretrain.gpu_input_pipeline import GpuInputPipeline
            from eventpretrain_amd.testing import make_args as _mk, synthetic_events as _syn
            pa = _mk(crop_min=0.8, input_size=S, fix_events_num=100_000, img_sensor_w=640, img_sensor_h=480, device="cuda")
            base_clip = _syn(4242, 150_000, width=640, height=480)
            evs2 = torch.from_numpy(np.concatenate([base_clip] * args.batch, 0)).to(device)
            off2 = np.arange(0, (args.batch + 1) * 150_000, 150_000, dtype=np.int64)
            frames2 = torch.randn(args.batch, 1, 480, 640, device=device)
            pipe = GpuInputPipeline(pa, seed=1, ring=4)
            # (what the host-planned form -- prepare() on a worker thread + run() -- would spend per batch on the host: reported beside)
            th = time.perf_counter()
            for s_ in range(8):
                pipe.prepare(off2, step=100 + s_, frame_size=(480, 640))
            host_ms = (time.perf_counter() - th) / 8 * 1e3
            # the device half as one HIP graph, SELF-DRIVEN: the batch plan (windows, counts, offsets, crop rows) is computed by a kernel
            # inside the graph from the clip offsets and a device-resident step counter -- per batch the host does one replay
            chain = pipe.capture(evs2, args.batch, frames=frames2, clip_offsets=off2)
            for s_ in range(3):
                chain.run_next()
            torch.cuda.synchronize()
            n_rep = 24
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            tw = time.perf_counter()
            c0.record()
            for s_ in range(n_rep):
                chain.run_next()
            c1.record()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - tw) / n_rep
            csec = c0.elapsed_time(c1) * 1e-3 / n_rep
            cbytes = pipe.algorithmic_bytes(np.full(args.batch, 100_000), fused=chain.fused) + args.batch * (480 * 640 + S * S) * 4.0
            result["loader_chain"] = {"value": args.batch / csec, "unit": "clips/s", "us_per_batch": csec * 1e6, "wall_us_per_batch": wall * 1e6,
                                      "bound": "hbm", "achieved": cbytes / csec / 1e9, "peak": HBM_PEAK_GBS, "frac": cbytes / csec / 1e9 / HBM_PEAK_GBS,
                                      "algorithmic_bytes_per_batch": cbytes, "host_prepare_ms_per_batch": host_ms, "host_ms_per_batch_in_this_loop": 0.0, "decision_stream": pipe.stream + " (plan on the device)",
                                      "note": "frac prices the bytes this form moves (window rows read once + augmented views written once + frames): the merged clip "
                                              "and the raw grids of the round-3 form (898 MB per batch, 655 us) are no longer written or read, so bytes fell faster than time",
                                      "includes": "get_random_index (100k of 150k events) -> events_augment -> events_reshape -> "
                                                  "events_to_voxel_grid -> evg_augment + frame_augment, 640x480 sensor clips resident in HBM; "
                                                  "the voxel grids binned straight from window rows + erase list + added rows (the merged clip is never written); "
                                                  "per batch ONE HIP-graph replay of 9 launches and nothing else on the host: the batch plan (windows, counts, "
                                                  "offsets, crop boxes), the erase / add rows and the noise are all drawn on the device from the counter stream; "
                                                  "host_prepare_ms_per_batch = what the host-planned form (prepare / run) would spend per batch on a worker thread"}
            result["end_to_end"] = {"value": args.batch * world / (ms * 1e-3 + csec), "unit": "samples/s",
                                    "includes": "the loader chain of the batch on the GPU (%.0f us) + optimiser step, serial, per GPU" % (csec * 1e6)}
            if headline and not multi and tuple(chain.out.shape) == tuple(vox.shape) and tuple(chain.tgt.shape) == tuple(tgt.shape):
                try:
                    # ... and MEASURED: the loop an epoch from raw events is -- one replay of the chain, which writes its grids and frame
                    # targets straight into the step's static inputs, then one replay of the step -- timed as a whole (events resident in
                    # HBM; per batch the host queues two graph replays)
                    chain2 = pipe.capture(evs2, args.batch, frames=frames2, clip_offsets=off2, out=executor.inputs[0], tgt_out=executor.inputs[1])
                    for _ in range(3):
                        chain2.run_next()
                        step()
                    torch.cuda.synchronize()
                    n_e2e = 30
                    t_e = time.perf_counter()
                    for _ in range(n_e2e):
                        chain2.run_next()
                        step()
                    torch.cuda.synchronize()
                    dt_e = (time.perf_counter() - t_e) / n_e2e
                    result["end_to_end"].update({"value": args.batch / dt_e, "ms_per_batch": dt_e * 1e3, "serial_sum_estimate": args.batch / (ms * 1e-3 + csec),
                                                 "includes": "MEASURED loop, %d batches: loader chain replay (events -> augmented grids + frame targets, %.0f us alone), "
                                                             "written straight into the step's static inputs -> optimiser step replay; per GPU" % (n_e2e, csec * 1e6)})
                except Exception as e2:      # keep the computed figure and say why the measured one is missing
                    result["end_to_end"]["measured_error"] = repr(e2)
        except Exception as e:      # a reported figure; never lose the bench line over it
            result["loader_chain"] = {"value": None, "unit": "clips/s", "error": repr(e)}
    if multi:
        dist.barrier()

    if rank == 0 and world == 1 and not args.no_cpu_baseline and headline and args.model == "base":
        n_thr = os.cpu_count() or 1
        try:
            result["cpu_baseline"] = cpu_baseline(fcfg, n_thr, 16, 6)          # all host cores
            result["cpu_baseline_1thread"] = cpu_baseline(fcfg, 1, 2, 3)         # the reference's shipped torch.set_num_threads(1)
            result["cpu_baseline_voxel"] = cpu_voxel_baseline()
        except Exception as e:  # the baseline is a reported figure; never lose the GPU line over it
            result["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": n_thr, "kind": "port", "sample": "failed: %r" % (e,)}
    if rank == 0:
        print(json.dumps(result))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
