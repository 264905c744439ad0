"""Round-4 additions on the GPU: the data-parallel default loop with its tensors on the device (two ranks on one card over gloo),
fresh dropout masks under HIP-graph replay."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(script, extra, n=2, timeout=600):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", script)] + extra
    env = dict(os.environ, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-6000:])
    return r


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_two_rank_epoch_loop_on_the_device(tmp_path, dtype):
    """the default epoch loop hands its executor the scaler's reducer without ever calling the scaler, so the
    1 / world gradient mean has to be set by the executor itself -- and the multi-GPU form (split backward, early decoder chunk,
    chunked weight gradients on two streams, AdamW in parts) had never run with a CUDA device and more than one rank. Two ranks on
    this one card (gloo transport), different data and mask noise per rank, three optimizer steps through pr_rec_one_epoch; against
    ONE process that averages the two ranks' losses per step by hand (eager, no reducer): logged loss, LR and every parameter's
    weighted checksum agree, and the two ranks end with identical weights."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dp_cuda_worker as w
    from eventpretrain_amd.utils.lr_sched import adjust_learning_rate
    from helpers import checksums
    out = tmp_path / "dp.json"
    steps, lr = 3, 1e-3
    _run_ranks("dp_cuda_worker.py", ["--out", str(out), "--dtype", dtype, "--steps", str(steps), "--lr", str(lr)])
    got = json.load(open(out))
    assert got["ranks_equal"] and abs(got["grad_scale"] - 0.5) < 1e-12
    assert got["note"].startswith("hip-graph"), got["note"]
    if dtype == "bf16":
        assert got["parts"] and got["split"], got        # the full multi-GPU form: split backward + AdamW in parts
    # the same three steps in one process: mean of the two ranks' losses, one backward, one AdamW step
    a, m, opt = w.build(dtype, lr)
    losses = []
    for s in range(steps):
        adjust_learning_rate(opt, s / steps, a)
        for r in range(2):
            x, y, noise = w.batch_of(r, s)
            o = m(x.cuda(), y.cuda(), is_rec=True, noise=noise.cuda())
            losses.append(o[0].item())
            (o[0] / 2).backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
    torch.cuda.synchronize()
    tol = 1e-5 if dtype == "f32" else 2e-3
    assert got["stats"]["reconstruct_loss"] == pytest.approx(sum(losses) / len(losses), rel=tol)
    worst = 0.0
    for k, p in m.named_parameters():
        ref = checksums(p)[2]
        scale = max(p.detach().abs().sum().item(), 1e-6)
        worst = max(worst, abs(got["wsums"][k] - ref) / scale)
    print(f"[dp on device, {dtype}] {got['note'][:60]}...; worst weighted-checksum distance / sum|w| = {worst:.2e}")
    assert worst <= (2e-6 if dtype == "f32" else 2e-4), worst


@pytest.mark.parametrize("kind", ["con_queue", "con_inbatch", "con_bcast", "swin"])
def test_two_rank_captured_steps_with_forward_collectives(tmp_path, kind):
    """VERDICT r3 missing 2 / 3: the data-parallel contrastive stage (a collective INSIDE its forward) and the data-parallel Swin step
    (per-rank window plans) used to step eagerly. Now: the key all-gather leaves the captured graphs (engine.ForwardCollectives --
    between two graphs for the in-batch InfoNCE, issued after the forward graph and consumed by the enqueue after the backward for the
    queue, the reference-faithful buffer broadcast in front of the step), and the Swin ranks agree per step whether every pattern fits
    the captured shape (one host-side MIN), falling back TOGETHER to a data-parallel eager step otherwise. Two ranks on this card
    (gloo), captured against eager from the same start: same losses, weights, queue contents and pointer; identical across ranks."""
    out = tmp_path / "dp.json"
    steps = 10 if kind == "swin" else 4
    _run_ranks("dp_cuda_worker.py", ["--scenario", kind, "--out", str(out), "--dtype", "f32", "--steps", str(steps)], timeout=900)
    got = json.load(open(out))
    e, g = got["eager"], got["graph"]
    assert g["note"].startswith("hip-graph"), g["note"]
    assert g["ranks_equal"] and e["ranks_equal"]
    assert g["split"], g                                      # [forward] / [backward] (or decoder / encoder backward) as two graphs
    if kind == "con_queue":
        assert g["n_post"] == 1 and g["n_graphs"] == 1 and g["fallbacks"] == 0
    if kind == "con_inbatch":
        assert g["n_graphs"] == 2 and g["n_post"] == 0        # the forward itself is split at the gather
    if kind == "con_bcast":
        assert g["n_pre"] == 1
    if kind == "swin":
        assert 0 < g["fallbacks"] < steps, g["fallbacks"]     # some steps replayed, some fell back -- on both ranks alike
    assert g["losses"] == pytest.approx(e["losses"], rel=2e-5), (g["losses"], e["losses"])
    worst = max(abs(g["wsums"][k] - e["wsums"][k]) / max(e["scale"][k], 1e-6) for k in e["wsums"])
    assert worst <= 5e-6, worst
    for k, v in e["bufs"].items():
        assert g["bufs"][k] == pytest.approx(v, rel=1e-5, abs=1e-6), k
    print(f"[dp captured, {kind}] {g['note'][:90]}...; fall-backs {g['fallbacks']}; worst weight distance {worst:.1e}")


def test_graph_replay_draws_fresh_dropout_masks():
    """the element-dropout key used to be a host integer frozen into the captured kernel arguments, so every
    replay of a captured step reused ONE mask per layer. The key is a device scalar drawn from torch's generator now
    (ops.draw_drop_seed), which advances under replay: two replays of a captured dropout give different masks, both with the
    asked-for rate, and the captured ViT block with drop > 0 gives different outputs per replay for the same input."""
    from eventpretrain_amd import ops
    from eventpretrain_amd.model.sub_module.vit_block import ViTBlock
    from eventpretrain_amd.testing import det_fill_module_
    ops.set_compute_dtype(torch.float32)
    x = torch.ones(1 << 16, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.DropoutFn.apply(x, 0.25, ops.draw_drop_seed(x.device))          # warm-up on the capture stream
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        y = ops.DropoutFn.apply(x, 0.25, ops.draw_drop_seed(x.device))
    outs = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        outs.append(y.clone())
    for o in outs:
        keep = (o != 0).float().mean().item()
        assert abs(keep - 0.75) < 0.01, keep
        assert torch.all((o == 0) | ((o - 1 / 0.75).abs() < 1e-6))
    assert not torch.equal(outs[0], outs[1]) and not torch.equal(outs[1], outs[2])
    # a whole block
    blk = ViTBlock(dim=64, num_heads=4, mlp_ratio=4., qkv_bias=True, drop=0.2, drop_path=0.0)
    det_fill_module_(blk)
    blk = blk.cuda().train()
    xb = torch.randn(4, 24, 64, device="cuda")
    with torch.cuda.stream(side):
        with torch.no_grad():
            blk(xb)
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=side):
        with torch.no_grad():
            yb = blk(xb)
    g2.replay()
    torch.cuda.synchronize()
    a = yb.clone()
    g2.replay()
    torch.cuda.synchronize()
    assert torch.isfinite(a).all() and not torch.equal(a, yb)
