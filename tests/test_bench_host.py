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


def test_the_command_lines_of_the_first_multi_gpu_script_parse():
    """tools/first_multi_gpu.sh is written for a machine nobody has had yet (more than one MI355X): every bench.py command line in
    it must at least be one bench.py's argument parser accepts (bench.py --parse-only: no torch, no GPU), for every N it loops
    over, and the script must be valid bash."""
    import json
    import re
    import shutil

    script = os.path.join(ROOT, "tools", "first_multi_gpu.sh")
    text = open(script).read()
    bash = shutil.which("bash")
    if bash:
        assert subprocess.run([bash, "-n", script], capture_output=True, text=True).returncode == 0
    lines = [ln.strip() for ln in text.splitlines() if re.match(r"\s*python bench\.py ", ln)]
    assert len(lines) == 2
    seen = set()
    for ln in lines:
        cmd = ln.split(" 2>")[0].split()[2:]  # the arguments, without the redirections
        for n in (1, 2, 4, 8):
            argv = [a.replace("$N", str(n)) for a in cmd] + ["--parse-only"]
            res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=60, cwd=ROOT)
            assert res.returncode == 0, (argv, res.stderr[-500:])
            parsed = json.loads(res.stdout)
            assert parsed["gpus"] == n and parsed["config"] == "cfg4_100mp" and parsed["backend"] == "nccl" and parsed["no_cpu_baseline"]
            seen.add((n, parsed["checksum"]))
    assert seen == {(n, c) for n in (1, 2, 4, 8) for c in (True, False)}
    assert "tests/test_gpu_multi.py" in text and "measured_ms_max_over_ranks" in text and "rccl_ranks" in text
