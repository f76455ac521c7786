"""bench.py --gpus N: the argument plumbing of the rank launcher (no GPU needed).

The driver calls `python bench.py --gpus N` (N = 1) or torch.distributed.run around it (N > 1); a bare
`python bench.py --gpus N` with N > 1 must start the N ranks itself and must never print an N-GPU line from
fewer devices."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launch_command_is_the_drivers_own_form():
    argv = ["--gpus", "4", "--steps", "5", "--warmup", "1"]
    cmd = bench.launch_command(4, argv, 29517)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29517"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == argv  # every rank sees this invocation's own arguments, --gpus included


def test_free_port_is_bindable():
    import socket
    p = bench.free_port()
    with socket.socket() as s:
        s.bind(("127.0.0.1", p))


def _run(*argv, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=300)


def _gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("mode", ["ranks", "group"])
def test_more_gpus_than_the_box_has_fails_loudly(mode):
    n = _gpus() + 1
    p = _run("--gpus", str(max(n, 2)), "--mode", mode)
    assert p.returncode != 0
    assert "visible GPUs" in p.stderr
    assert p.stdout.strip() == ""  # no JSON line from fewer devices than asked for


def test_flag_and_world_size_must_agree():
    p = _run("--gpus", "1", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "disagree" in p.stderr and p.stdout.strip() == ""
