"""bench.py's launch logic without a GPU: `python bench.py --gpus N` (the form the driver uses for N = 1) has to start its N ranks
itself -- as a child running torch.distributed.run, before anything touches HIP -- and fail, on a machine without GPUs, for the
lack of a device and not with a usage message."""

import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_with_two_gpus_requested_launches_its_own_ranks():
    import torch

    if torch.cuda.device_count() >= 2:
        import pytest

        pytest.skip("this machine has the GPUs: tests/test_gpu_multi.py runs the real thing")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert res.returncode != 0
    assert "must be launched with" not in res.stderr
    # both ranks got as far as looking for their device
    assert "needs GPU 1" in res.stderr and "--gpus 2 needs 2" in res.stderr, res.stderr[-2000:]


def test_self_launch_command_line(monkeypatch):
    import bench

    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, **kw):
        seen["cmd"] = cmd
        return Done()

    monkeypatch.setattr("subprocess.run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    assert bench.self_launch(4) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3"]
