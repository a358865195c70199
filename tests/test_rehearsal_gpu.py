"""The N > 1 job exactly as the driver starts it -- `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`:
one PROCESS per rank, mi355_group_adopt_rank with an id made by rank 0 and handed around by torch.distributed, the timed
steps, mi355_group_gather behind them (and, with --gather last, inside them), the gather-every-batch leg -- on the ONE GPU of a test box:
`--rehearse-on-one-gpu` puts every rank on device 0, torch.distributed on gloo and the exchange below the C-ABI on the
tests' inter-process stand-in for RCCL (tests/mock_rccl/mock_rccl_ipc.cpp; real RCCL refuses two ranks on one device).
What it cannot show is RCCL's own transport; everything else of bench.py's N > 1 path runs.  The line's `gather_verified`
says that the root holds, for every rank, the entries that rank produced."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOCK = os.path.join(ROOT, "tests", "mock_rccl", "librccl_mock_ipc.so")
SMALL = ["--width", "640", "--height", "360", "--batch", "8", "--steps", "3", "--warmup", "2", "--gather-every-steps", "2",
         "--no-cpu", "--no-pair", "--no-filters", "--no-host-path", "--preheat-s", "0.05", "--steady-steps", "20",
         # BASELINE configs[4] across the ranks of the job, at a size the oracle checks in a moment
         "--config5-size", "320", "180", "4", "--config5-steps", "2"]


def check_config5(line, nranks, rccl):
    """The config5 object of an N-rank line: the 4K-shaped sequence dealt round-robin over the ranks that are there, the gather
    to rank 0, rank 0's copy of every rank's first and last frame against the oracle."""
    c5 = line["config5"]
    assert c5["parity"] is True and c5["gather_verified"] is True and c5["ranks_seen"] == nranks, c5
    assert len(c5["frac_per_rank"]) == nranks and all(f >= 0 for f in c5["frac_per_rank"]) and c5["frac"] >= 0   # (tiny frames: the fractions round to 0.000x)
    assert c5["value"] > 0 and c5["final_gather_ms"] > 0 and c5["gather_bytes"] > 4 * 5 * nranks and c5["gather_gbps"] > 0
    assert 0 < c5["value_with_final_gather"] < c5["value"]
    assert ("RCCL, csrc/group.hip" in c5["gather_impl"]) if rccl else ("REHEARSAL" in c5["gather_impl"])
    assert line["parity"]["config5"] is True

pytestmark = pytest.mark.gpu


def run_bench(extra, env=None, timeout=420):
    e = dict(os.environ, MOCK_RCCL_TIMEOUT_S="60")
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, cwd=ROOT, env=e, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout)
    lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


@pytest.mark.skipif(not os.path.exists(MOCK), reason="tests/mock_rccl/librccl_mock_ipc.so not built (__graft_entry__.build())")
def test_two_processes_run_the_bench_sequence():
    r, line = run_bench(["--gpus", "2", "--rehearse-on-one-gpu"] + SMALL)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    assert line is not None and line["n_gpus"] == 2 and line["ranks_seen"] == 2
    assert "rehearsal" in line and "REHEARSAL" in line["config"]["gather_impl"]
    assert line["gather_verified"] is True
    assert line["gather_ms"] > 0 and line["gather_bytes"] > 4 * 9 * 2
    # the default: the exchange behind the timed steps, with its own time and the throughput that includes it
    assert line["config"]["gather"].startswith("after") and line["final_gather_ms"] > 0
    assert 0 < line["value_with_final_gather"] < line["value"]
    ge = line["gather_every"]
    assert ge["steps"] == 2 and ge["gather_bytes"] > 0 and ge["value"] > 0
    assert ge["gather_verified"] is True
    assert line["value"] > 0 and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["gather_ran"] is True and line["cold_start_window"]["value"] > 0 and line["preheat"]["steps"] > 0
    check_config5(line, 2, rccl=False)


@pytest.mark.skipif(not os.path.exists(MOCK), reason="tests/mock_rccl/librccl_mock_ipc.so not built")
def test_three_processes_and_a_rank_that_cannot_form_the_group():
    """Three ranks sharing out ONE sequence round-robin (BASELINE config 5's shape); then a job whose ranks cannot load the
    library behind MI355_RCCL_LIB: every rank says which step failed and the job ends with a non-zero exit status instead
    of hanging or measuring something else."""
    r, line = run_bench(["--gpus", "3", "--rehearse-on-one-gpu", "--shard", "roundrobin", "--gather", "last"] + SMALL)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    assert line["ranks_seen"] == 3 and line["gather_verified"] is True and "config5" not in line   # (--shard roundrobin IS that shape)
    assert line["config"]["gather"].startswith("last") and "value_with_final_gather" not in line   # the exchange inside the timed region
    # ... and the default job with three ranks: the config5 object with a shard per rank
    r, line = run_bench(["--gpus", "3", "--rehearse-on-one-gpu"] + SMALL)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    assert line["ranks_seen"] == 3 and line["gather_verified"] is True
    check_config5(line, 3, rccl=False)
    assert line["config"]["gather"].startswith("after")
    r, line = run_bench(["--gpus", "2", "--rehearse-on-one-gpu"] + SMALL, env={"MI355_RCCL_LIB": "/nonexistent/librccl.so"})
    err = r.stderr.decode(errors="replace")
    assert r.returncode != 0 and line is None
    assert "forming the group failed in mi355_group_unique_id" in err and "MI355_RCCL_LIB" in err
    assert "rank 1 of 2" in err or "rank 0 of 2" in err


def test_one_rank_under_the_launcher_reports_the_gather():
    """N = 1 the way the driver could start it (torch.distributed.run, real RCCL, one rank): the line carries ranks_seen,
    gather_ms and gather_bytes of the same code path the N > 1 runs take."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL
    r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, r.stderr.decode(errors="replace")[-3000:]
    line = json.loads(lines[-1])
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and line["gather_verified"] is True
    assert line["gather_ms"] is not None and line["gather_bytes"] > 0 and line["gather_ran"] is True
    assert "RCCL" in line["config"]["gather_impl"] and "rehearsal" not in line
    check_config5(line, 1, rccl=True)


def test_one_rank_whose_group_cannot_be_formed_still_measures_its_steps():
    """The default job (--gather after: the exchange behind the timed steps) under the launcher with real torch.distributed
    (backend nccl = RCCL), when the library behind the C-ABI cannot load its RCCL: the steps are measured all the same, the
    exchange runs in its torch.distributed form over device tensors, and the line says which form with the library's error.
    The same failure with the exchange INSIDE the timed steps ends the job instead."""
    import socket

    def launch(extra):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1"] + SMALL + extra
        r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=420,
                           env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MI355_RCCL_LIB="/nonexistent/librccl.so"))
        lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
        return r, (json.loads(lines[-1]) if lines else None)

    r, line = launch([])
    err = r.stderr.decode(errors="replace")
    assert r.returncode == 0 and line is not None, err[-3000:]
    assert "forming the group failed in mi355_group_unique_id" in err
    assert line["config"]["gather_impl"].startswith("torch.distributed (mi355_group unavailable")
    assert line["value"] > 0 and line["ranks_seen"] == 1
    # the exchange DID run (its torch.distributed form, behind the steps, timed the same way), and the steps ran as they do
    # with a group: on the core's own stream, pipelined
    assert line["gather_ran"] is True and line["gather_ms"] > 0 and line["final_gather_ms"] > 0 and line["gather_verified"] is True
    assert "kernel_ms_basis" in line["roofline"]
    c5 = line["config5"]
    assert c5["gather_impl"].startswith("torch.distributed (mi355_group unavailable") and c5["parity"] is True and c5["gather_verified"] is True
    r, line = launch(["--gather", "last"])
    assert r.returncode != 0 and line is None
