"""bench.py's own launcher (`python bench.py --gpus N` with no torch.distributed.run around it), CPU side: the parent starts
the rank processes as children, touches no GPU itself, and leaves with their exit code.  Without a GPU every rank stops at
"bench.py needs a GPU" -- which is the loud failure this checks; the working N = 2 run is tests/test_gpu_rccl.py."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def test_launch_ranks_command_line(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    assert bench.launch_ranks(4, ["--gpus", "4", "--steps", "3"]) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-5] == os.path.join(ROOT, "bench.py") and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_gpus_2_without_a_gpu_fails_loudly_through_the_launcher():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the working run is tests/test_gpu_rccl.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--vertices", "1000", "--edges", "5000",
                        "--no-index", "--no-cpu-baseline", "--no-config5"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0
    assert "bench.py needs a GPU" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
