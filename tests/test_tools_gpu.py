"""tools/diffbench (the C++ harness over the C-ABI): its device-side synthetic generator must produce
the same bytes as cudavideostream_amd/synth.py, and its stream run must report the same changed-byte
count as the oracle on that stream."""
import json
import os
import subprocess

import numpy as np
import pytest

from cudavideostream_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tools", "diffbench")


def run(*args):
    out = subprocess.run([EXE, *map(str, args)], capture_output=True, text=True, check=True, timeout=300).stdout
    return json.loads(out.strip().splitlines()[-1])


@pytest.mark.skipif(not os.path.exists(EXE), reason="tools/diffbench not built")
@pytest.mark.parametrize("t", [-1, 0, 7])
def test_generator_matches_synth_py(t):
    w, h, seed = 320, 180, 21
    r = run("--width", w, "--height", h, "--seed", seed, "--checksum", t)
    f = synth.webcam_frame(t, w, h, seed=seed).astype(np.uint64)
    i = np.arange(f.size, dtype=np.uint64)
    assert r["sum"] == int(f.sum())
    assert r["wsum"] == int((f * (i % np.uint64(65521) + np.uint64(1))).sum())


@pytest.mark.skipif(not os.path.exists(EXE), reason="tools/diffbench not built")
def test_stream_counts_match_oracle(po):
    w, h, B = 320, 180, 12
    r = run("--width", w, "--height", h, "--batch", B, "--steps", 1, "--warmup", 0)
    base, frames = synth.webcam_stream(B, w, h, seed=21)
    off, _, _, _ = po.diff_stream(frames, base)
    assert abs(r["changed_bytes_per_frame"] * B - int(off[-1])) < 0.5 * B


RT = os.path.join(ROOT, "tools", "roundtrip")


@pytest.mark.skipif(not os.path.exists(RT), reason="tools/roundtrip not built")
@pytest.mark.parametrize("w,h,T,B", [(320, 180, 24, 8), (97, 13, 10, 4), (1920, 1080, 6, 3)])
def test_cpp_server_to_client_round_trip_over_the_c_abi(w, h, T, B):
    """tools/roundtrip: a g++-only program (no HIP header) drives a server core and a client core through
    include/mi355diff.h, moving the sender's byte stream through a pipe; it exits non-zero on any mismatch."""
    out = subprocess.run([RT, "--width", str(w), "--height", str(h), "--frames", str(T), "--batch", str(B)],
                         capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["roundtrip"] == "ok" and r["max_abs_error"] <= 20
    assert r["wire_bytes"] == 4 * T + 5 * r["changed_bytes"]
    assert r["wire_bytes"] < r["raw_bytes"]


CP = os.path.join(ROOT, "tools", "compat_pipe")


@pytest.mark.skipif(not os.path.exists(CP), reason="tools/compat_pipe not built")
@pytest.mark.parametrize("w,h,T", [(640, 360, 12), (97, 13, 7), (1920, 1080, 5)])
def test_cpp_drop_in_class_pipelined_equals_blocking(w, h, T):
    """tools/compat_pipe: diff::cuda::CUDACore (compat/) -- exec_submit/exec_wait with 3 frames in flight give
    the same h_pos / h_xs / diff bytes per frame as the reference's blocking exec_core call."""
    out = subprocess.run([CP, str(w), str(h), str(T)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert r["compat_pipe"] == "ok" and r["changed_bytes"] > 0
