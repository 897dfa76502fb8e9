"""`bench.py --gpus N` with no launcher around it must become N ranks by itself ("What's missing" 1): the bare command
is what a driver types.  --launch-check runs the launch, the rendezvous and the spanning-bus exchange's collectives on host tensors
under gloo -- no engine, no GPU (the engine has no CPU render path) -- and prints the line's n_gpus."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=240)


@pytest.mark.parametrize("world", [2, 8])
def test_the_bare_command_starts_its_own_ranks(world):
    p = _run(["--gpus", str(world), "--launch-check"])
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                                   # stdout carries exactly one line
    line = json.loads(lines[0])
    assert line["n_gpus"] == world and line["ranks_counted"] == world and line["exchange_ok"] is True
    assert line["launch_check"] is True and line["value"] is None   # not a measurement


def test_under_a_launcher_the_command_does_not_launch_again():
    """The driver's N > 1 form: `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`: WORLD_SIZE is set, the
    process is a rank."""
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert b"without a launcher" not in p.stderr
    line = json.loads([ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2


def test_one_gpu_never_launches():
    """N = 1 stays one process: without a GPU the command fails loudly (no CPU render path), it does not spawn anything."""
    p = _run(["--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    assert p.returncode != 0 and b"needs an MI355X" in p.stderr and b"without a launcher" not in p.stderr


def test_failing_ranks_fail_the_command():
    """The launcher forwards the ranks' verdict: without a GPU every rank stops ("needs an MI355X": no CPU render path), the command
    prints no result line and exits non-zero -- it never invents a line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert b"needs an MI355X" in p.stderr
