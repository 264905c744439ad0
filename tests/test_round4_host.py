"""Round-4 host-side tests (no GPU): bench.py's self-launcher for --gpus N > 1."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launch_builds_the_drivers_command(monkeypatch):
    """`python bench.py --gpus N ...` outside torch.distributed.run starts the ranks as a CHILD process (never exec) with the same
    arguments and returns the child's exit code; fewer devices than ranks is a clear refusal, not a hang."""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setenv("EVP_BENCH_SHARE_DEVICE", "1")
    rc = bench.self_launch(4, ["--gpus", "4", "--steps", "7", "--warmup", "3"])
    assert rc == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # without the rehearsal switch the parent counts devices first (0 in the CPU container) and refuses with rc 2
    monkeypatch.delenv("EVP_BENCH_SHARE_DEVICE")
    seen.clear()
    import torch
    if torch.cuda.device_count() < 4:
        assert bench.self_launch(4, ["--gpus", "4"]) == 2 and not seen


def test_bench_gpus2_reaches_the_ranks_without_touching_the_gpu_in_the_parent():
    """The real thing, end to end, in a container without a GPU: the launcher starts two ranks, each rank stops at "no HIP device"
    (the library's loud refusal), the parent relays a non-zero exit code. Under torchrun already (WORLD_SIZE set) nothing is
    re-launched."""
    import torch
    if torch.cuda.device_count() > 0:
        import pytest
        pytest.skip("a GPU is visible: the ranks would run the benchmark")
    env = dict(os.environ, EVP_BENCH_SHARE_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "launching 2 ranks" in r.stderr and "no HIP device visible" in r.stderr, r.stderr[-2000:]
    assert r.stdout.strip() == ""        # no JSON line from a failed run
