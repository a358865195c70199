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
