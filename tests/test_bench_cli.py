"""bench.py's launcher logic on a box without a GPU: `--gpus N` with no launcher around starts the N ranks itself
(torch.distributed.run on 127.0.0.1) before anything touches the GPU, and leaves with their exit code; a rank that
finds no GPU refuses (there is no CPU fallback for the hot path)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_gpu():
    try:
        import torch
        return not torch.cuda.is_available()
    except Exception:   # noqa: BLE001
        return True


@pytest.mark.skipif(not _no_gpu(), reason="launcher test for the GPU-less build container")
def test_gpus_2_spawns_two_ranks_and_reports_their_failure():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode != 0
    # two ranks ran bench.py under the launcher and the job refused for want of a GPU: the rank that notices first says so,
    # the launcher then ends the other one -- which may or may not have come to say it too (a race under load) -- and its
    # report names both
    assert out.stderr.count("bench.py needs an MI355X") >= 1, out.stderr[-2000:]
    assert "local_rank: 0" in out.stderr and "local_rank: 1" in out.stderr, out.stderr[-2000:]
    assert out.stdout.strip() == ""          # no JSON line from a failed job


@pytest.mark.skipif(not _no_gpu(), reason="launcher test for the GPU-less build container")
def test_single_rank_without_gpu_refuses():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], capture_output=True, text=True,
                         timeout=300, env=env, cwd=ROOT)
    assert out.returncode != 0 and "no CPU fallback" in out.stderr
