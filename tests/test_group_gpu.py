"""The multi-GPU entry points of the C-ABI (mi355_group_*, csrc/group.hip) on the GPUs this box has: a group of
one (or more) devices in one process, and the one-member-per-process form with an RCCL id.  The exchange logic
for several ranks is the same code path with more peers; its ordering rules are also covered by the gloo tests
of cudavideostream_amd/gather.py (tests/test_gather_gloo.py)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from cudavideostream_amd import lib, synth

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from cudavideostream_amd.group import CUDAGroup, unique_id  # noqa: E402
from gpu_util import DEV, CUDACore, to_dev  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_stream(po, base, frames):
    return po.diff_stream(frames, base)


def test_adopted_rank_gathers_its_own_stream(po):
    """One member per process (what bench.py does under torch.distributed.run), here a group of one rank."""
    w, h, T = 96, 54, 7
    n = 3 * w * h
    base, frames = synth.webcam_stream(T, w, h, seed=6)
    eo, exs, edf, est = oracle_stream(po, base, frames)
    with CUDACore(w, h, max_batch=T, sample_mat_data=base) as core:
        with CUDAGroup.adopt(core, 1, 0, unique_id()) as grp:
            assert grp.nranks == 1 and grp.local_members == 1 and grp.rank_of(0) == 0
            d_fr = to_dev(frames)
            d_off = torch.zeros(T + 1, dtype=torch.int32, device=DEV)
            d_xs = torch.full((T * n,), -3, dtype=torch.int32, device=DEV)
            d_df = torch.zeros(T * n, dtype=torch.uint8, device=DEV)
            r_off = torch.full((1, T + 1), -1, dtype=torch.int32, device=DEV)
            r_xs = torch.full((T * n,), -5, dtype=torch.int32, device=DEV)
            r_df = torch.full((T * n,), 0x77, dtype=torch.uint8, device=DEV)
            torch.cuda.synchronize()
            grp.diff_stream_batch([d_fr], T, [d_off], [d_xs], [d_df], T * n, stride=n)
            counts = grp.gather(0, T, [d_off], [d_xs], [d_df], T * n, r_off, r_xs, r_df, T * n)
            grp.synchronize()
            tot = int(eo[-1])
            assert counts.tolist() == [tot]
            assert np.array_equal(r_off.cpu().numpy().view(np.uint32)[0], eo)
            assert np.array_equal(r_xs[:tot].cpu().numpy(), exs) and np.array_equal(r_df[:tot].cpu().numpy(), edf)
            assert (r_xs[tot:tot + 8].cpu().numpy() == -5).all()
            assert np.array_equal(core.get_state(), est)
            # capacity below the gathered total is an error, not a truncation
            with pytest.raises(lib.Mi355Error):
                grp.gather(0, T, [d_off], [d_xs], [d_df], T * n, r_off, r_xs, r_df, tot - 1)


def test_group_in_one_process_over_the_visible_devices(po):
    """mi355_group_create over every GPU of the box (1 on the test box): independent streams, gather to rank 0."""
    ndev = torch.cuda.device_count()
    w, h, T = 80, 45, 5
    n = 3 * w * h
    L = lib.load()
    with CUDAGroup.create(w, h, ndev, max_batch=T) as grp:
        assert grp.nranks == ndev == grp.local_members
        bufs, want = [], []
        for r in range(ndev):
            base, frames = synth.webcam_stream(T, w, h, seed=40 + r)
            want.append(oracle_stream(po, base, frames))
            lib.check(L.mi355_set_state(grp.core_handle(r), np.ascontiguousarray(base).ctypes.data))
            dev = f"cuda:{r}"
            bufs.append(dict(fr=torch.from_numpy(frames).to(dev), off=torch.zeros(T + 1, dtype=torch.int32, device=dev),
                             xs=torch.zeros(T * n, dtype=torch.int32, device=dev),
                             df=torch.zeros(T * n, dtype=torch.uint8, device=dev)))
        r_off = torch.zeros((ndev, T + 1), dtype=torch.int32, device="cuda:0")
        r_xs = torch.zeros(ndev * T * n, dtype=torch.int32, device="cuda:0")
        r_df = torch.zeros(ndev * T * n, dtype=torch.uint8, device="cuda:0")
        torch.cuda.synchronize()
        grp.diff_stream_batch([b["fr"] for b in bufs], T, [b["off"] for b in bufs], [b["xs"] for b in bufs],
                              [b["df"] for b in bufs], T * n, stride=n)
        counts = grp.gather(0, T, [b["off"] for b in bufs], [b["xs"] for b in bufs], [b["df"] for b in bufs], T * n,
                            r_off, r_xs, r_df, ndev * T * n)
        grp.synchronize()
        at = 0
        for r in range(ndev):
            eo, exs, edf, _ = want[r]
            tot = int(eo[-1])
            assert int(counts[r]) == tot
            assert np.array_equal(r_off[r].cpu().numpy().view(np.uint32), eo)
            assert np.array_equal(r_xs[at:at + tot].cpu().numpy(), exs)
            assert np.array_equal(r_df[at:at + tot].cpu().numpy(), edf)
            at += tot


GD = os.path.join(ROOT, "tools", "group_demo")


@pytest.mark.skipif(not os.path.exists(GD), reason="tools/group_demo not built")
def test_cpp_group_demo_over_the_c_abi():
    """tools/group_demo: a g++-only program (no HIP header) drives mi355_group_* and checks the root's copy."""
    ndev = torch.cuda.device_count()
    out = subprocess.run([GD, "--ndev", str(ndev), "--width", "320", "--height", "180", "--frames", "6"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["ok"] and r["ndev"] == ndev and r["entries_at_root"] > 0


# ---- several ranks on the one GPU of the test box -------------------------------------------------------------------
# Real RCCL refuses two ranks on one device.  tests/mock_rccl is a single-process stand-in for the ten RCCL entry
# points group.hip binds (loaded through MI355_RCCL_LIB, in a child process: the binding is per process): with it
# the gather-v logic -- counts, rank-ordered places, matching of sends and receives, a root other than 0 -- runs
# with 2..4 ranks here.  What it cannot show is RCCL's own transport; that is the driver's multi-GPU run.
MOCK_SRC = os.path.join(ROOT, "tests", "mock_rccl", "mock_rccl.cpp")
MOCK = os.path.join(ROOT, "tests", "mock_rccl", "librccl_mock.so")


@pytest.fixture(scope="module")
def mock_rccl():
    if not os.path.exists(MOCK) or os.path.getmtime(MOCK) < os.path.getmtime(MOCK_SRC):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", MOCK, MOCK_SRC], check=True)
    return dict(os.environ, MI355_RCCL_LIB=MOCK)


@pytest.mark.skipif(not os.path.exists(GD), reason="tools/group_demo not built")
@pytest.mark.parametrize("ndev,root", [(2, 0), (2, 1), (3, 1), (4, 3)])
def test_cpp_group_demo_several_ranks(mock_rccl, ndev, root):
    out = subprocess.run([GD, "--ndev", str(ndev), "--root", str(root), "--same-device", "1", "--width", "160",
                          "--height", "90", "--frames", "5"], capture_output=True, text=True, timeout=120, env=mock_rccl)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["ok"] and r["ndev"] == ndev and r["root"] == root and r["entries_at_root"] > 0


CHILD = r"""
import sys
import numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
from cudavideostream_amd import lib, synth
from cudavideostream_amd.group import CUDAGroup
from oracle import pyoracle as po
ndev, root, w, h, T = 3, 2, 96, 54, 6
n = 3 * w * h
L = lib.load()
with CUDAGroup.create(w, h, ndev, devices=[0] * ndev, max_batch=T) as grp:
    bufs, want = [], []
    for r in range(ndev):
        base, frames = synth.webcam_stream(T, w, h, seed=70 + r)
        if r == 1:
            frames = np.repeat(base[None, :], T, axis=0)          # a rank with nothing to send
        want.append(po.diff_stream(frames, base))
        lib.check(L.mi355_set_state(grp.core_handle(r), np.ascontiguousarray(base).ctypes.data))
        bufs.append(dict(fr=torch.from_numpy(np.ascontiguousarray(frames)).cuda(), off=torch.zeros(T + 1, dtype=torch.int32, device="cuda"),
                         xs=torch.zeros(T * n, dtype=torch.int32, device="cuda"), df=torch.zeros(T * n, dtype=torch.uint8, device="cuda")))
    r_off = torch.zeros((ndev, T + 1), dtype=torch.int32, device="cuda")
    r_xs = torch.full((ndev * T * n,), -9, dtype=torch.int32, device="cuda")
    r_df = torch.zeros(ndev * T * n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    grp.diff_stream_batch([b["fr"] for b in bufs], T, [b["off"] for b in bufs], [b["xs"] for b in bufs], [b["df"] for b in bufs], T * n, stride=n)
    counts = grp.gather(root, T, [b["off"] for b in bufs], [b["xs"] for b in bufs], [b["df"] for b in bufs], T * n, r_off, r_xs, r_df, ndev * T * n)
    grp.synchronize()
    at = 0
    for r in range(ndev):
        eo, exs, edf, _ = want[r]
        tot = int(eo[-1])
        assert int(counts[r]) == tot, (r, counts, tot)
        assert np.array_equal(r_off[r].cpu().numpy().view(np.uint32), eo)
        assert np.array_equal(r_xs[at:at + tot].cpu().numpy(), exs) and np.array_equal(r_df[at:at + tot].cpu().numpy(), edf)
        at += tot
    assert int(counts[1]) == 0 and (r_xs[at:at + 4].cpu().numpy() == -9).all()
print("GROUP-OK")
"""


def test_three_ranks_against_the_oracle(mock_rccl):
    """Three ranks (one of them with an empty stream), root 2: what arrives at the root, rank by rank, is the oracle's."""
    code = CHILD % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=mock_rccl, cwd=ROOT)
    assert out.returncode == 0 and "GROUP-OK" in out.stdout, out.stderr[-2000:]


# ---- one member per process, with threads standing in for the processes ------------------------------------------------
# bench.py under torch.distributed.run has ONE member per process (mi355_group_adopt_rank).  What can go wrong there and
# cannot in the one-process form: a rank that returns early while its peers go on to send.  The mock's threaded shape
# (ncclCommInitRank with nranks > 1: a rank per thread, a group's end blocks until the peers have posted their side, a
# partner that never comes is an error after MOCK_RCCL_TIMEOUT_S) runs exactly that protocol on the one GPU.
CHILD_THREADS = r"""
import sys, threading
import numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
from cudavideostream_amd import CUDACore, lib, synth
from cudavideostream_amd.group import CUDAGroup, unique_id
from oracle import pyoracle as po
R, root, w, h, T = 3, 1, 96, 54, 6
n = 3 * w * h
ident = unique_id()
want, outcome, roots = {}, {}, {}
barrier = threading.Barrier(R)

def rank_main(r):
    try:
        base, frames = synth.webcam_stream(T, w, h, seed=90 + r)
        want[r] = po.diff_stream(frames, base)
        tot = int(want[r][0][-1])
        with CUDACore(w, h, max_batch=T, sample_mat_data=base) as core:
            with CUDAGroup.adopt(core, R, r, ident) as grp:
                d_fr = torch.from_numpy(np.ascontiguousarray(frames)).cuda()
                d_off = torch.zeros(T + 1, dtype=torch.int32, device="cuda")
                d_xs = torch.zeros(T * n, dtype=torch.int32, device="cuda")
                d_df = torch.zeros(T * n, dtype=torch.uint8, device="cuda")
                r_off = torch.zeros((R, T + 1), dtype=torch.int32, device="cuda") if r == root else None
                r_xs = torch.full((R * T * n,), -9, dtype=torch.int32, device="cuda") if r == root else None
                r_df = torch.zeros(R * T * n, dtype=torch.uint8, device="cuda") if r == root else None
                torch.cuda.synchronize()
                grp.diff_stream_batch([d_fr], T, [d_off], [d_xs], [d_df], T * n, stride=n)
                res = []
                # (1) the root's capacity is one entry short: EVERY rank must get the error, nobody may wait
                barrier.wait()
                try:
                    grp.gather(root, T, [d_off], [d_xs], [d_df], T * n, r_off, r_xs, r_df, (sum_tot[0] - 1) if r == root else 0)
                    res.append("no error")
                except lib.Mi355Error as e:
                    res.append("capacity" if "root capacity" in str(e) else "other: " + str(e))
                # (2) rank 2 claims buffers smaller than its batch: every rank fails alike
                barrier.wait()
                try:
                    grp.gather(root, T, [d_off], [d_xs], [d_df], (tot - 1) if r == 2 else T * n, r_off, r_xs, r_df, R * T * n if r == root else 0)
                    res.append("no error")
                except lib.Mi355Error as e:
                    res.append("member" if "member_capacity" in str(e) else "other: " + str(e))
                # (2b) rank 0 calls with a negative nframes, (2c) rank 2 with another nframes than its peers: programmer
                # errors of ONE rank, reported on every rank -- nobody is left inside the all-gather
                barrier.wait()
                try:
                    grp.gather(root, -1 if r == 0 else T, [d_off], [d_xs], [d_df], T * n, r_off, r_xs, r_df, R * T * n if r == root else 0)
                    res.append("no error")
                except lib.Mi355Error as e:
                    res.append("call" if ("invalid arguments" in str(e) or "nframes outside" in str(e)) and e.code == lib.ERR_INVALID else "other: " + str(e))
                barrier.wait()
                try:
                    grp.gather(root, T - 1 if r == 2 else T, [d_off], [d_xs], [d_df], T * n, r_off, r_xs, r_df, R * T * n if r == root else 0)
                    res.append("no error")
                except lib.Mi355Error as e:
                    res.append("frames" if "disagree about nframes" in str(e) else "other: " + str(e))
                # (3) and the group is still usable: the real gather
                barrier.wait()
                counts = grp.gather(root, T, [d_off], [d_xs], [d_df], T * n, r_off, r_xs, r_df, R * T * n if r == root else 0)
                grp.synchronize()
                res.append([int(c) for c in counts])
                if r == root:
                    roots["off"], roots["xs"], roots["df"] = r_off.cpu().numpy().view(np.uint32), r_xs.cpu().numpy(), r_df.cpu().numpy()
                barrier.wait()   # nobody tears its communicator down while a peer is still inside the gather
                outcome[r] = res
    except Exception as e:   # noqa: BLE001
        outcome[r] = ["exception: " + repr(e)]
        barrier.abort()

# the oracle's totals are needed by the root before the threads start
sum_tot = [0]
for r in range(R):
    base, frames = synth.webcam_stream(T, w, h, seed=90 + r)
    sum_tot[0] += int(po.diff_stream(frames, base)[0][-1])
threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(R)]
for t in threads: t.start()
for t in threads: t.join(timeout=200)
assert not any(t.is_alive() for t in threads), "a rank hangs"
tots = [int(want[r][0][-1]) for r in range(R)]
for r in range(R):
    assert outcome[r] == ["capacity", "member", "call", "frames", tots], (r, outcome[r])
at = 0
for r in range(R):
    eo, exs, edf, _ = want[r]
    assert np.array_equal(roots["off"][r], eo)
    assert np.array_equal(roots["xs"][at:at + tots[r]], exs) and np.array_equal(roots["df"][at:at + tots[r]], edf)
    at += tots[r]
print("THREADS-OK")
"""


def test_one_member_per_process_protocol_with_threads(mock_rccl):
    """Three ranks, one per thread (the one-member-per-process form): a root capacity that is too small, a member
    whose batch overflowed, one rank's invalid nframes and ranks that disagree about nframes are reported on EVERY rank,
    before anything is sent; the gather that follows delivers the oracle's streams at root 1."""
    code = CHILD_THREADS % {"root": ROOT, "tests": os.path.join(ROOT, "tests")}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=400,
                         env=dict(mock_rccl, MOCK_RCCL_TIMEOUT_S="30"), cwd=ROOT)
    assert out.returncode == 0 and "THREADS-OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
