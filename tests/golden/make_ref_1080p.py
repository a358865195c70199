#!/usr/bin/env python3
"""Generates tests/golden/ref_server_cpu_1080p.npz: a run of the REFERENCE ITSELF at BASELINE size.

The reference's server/src/server.cpp CPU branch (gray-avg -> histogram -> two-max -> binarize, server.cpp:96-135,
compiled unmodified into oracle/_ref/server_cpu; stand-in ThreadsCore and -D'd common.h, see oracle/Makefile) is
driven with 1920x1080 frames of the seeded S1 generator (cudavideostream_amd/synth.py: any frame can be regenerated
from (t, width, height, seed), so the 6 MB inputs are NOT stored).  Stored: per frame the SHA-256 of the 6 220 800
output bytes, the number of white bytes, and the threshold the output pins (the largest gray value that stayed black,
when the next value up turned white -- the gray-avg of the input is integer arithmetic; -1 where the frame is all
black or all white).  Data only.
Run from the repo root in the build container (needs /root/reference for oracle/_ref/server_cpu)."""
import hashlib
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cudavideostream_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

W, H, T, SEED = 1920, 1080, 4, 77


def frames_1080p():
    """The inputs: S1 frames 0..T-1; frame 1 darkened (threshold clamp 50), frame 2 brightened (clamp 200)."""
    base = synth.webcam_frame(-1, W, H, seed=SEED)
    fr = np.stack([synth.webcam_frame(t, W, H, seed=SEED) for t in range(T)])
    fr[1] = (fr[1] // 6).astype(np.uint8)
    fr[2] = (200 + fr[2] // 5).astype(np.uint8)
    return base, fr


def main():
    po.build()
    assert po.ref_server_cpu_path(), "oracle/_ref/server_cpu is not built (needs /root/reference)"
    base, fr = frames_1080p()
    with tempfile.TemporaryDirectory() as d:
        out = po.run_ref_server_cpu(base, fr, W, H, d)
    sha = np.stack([np.frombuffer(hashlib.sha256(out[t].tobytes()).digest(), np.uint8) for t in range(T)])
    white = np.array([int((out[t] == 255).sum()) for t in range(T)], np.int64)
    thr = []
    for t in range(T):
        px = fr[t].reshape(-1, 3).astype(np.int32)
        g = (px[:, 0] + px[:, 1] + px[:, 2]) // 3
        w = out[t].reshape(-1, 3)[:, 0] == 255
        thr.append(int(g[w].min()) - 1 if 0 < w.sum() < w.size and int(g[w].min()) - 1 == int(g[~w].max()) else -1)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_server_cpu_1080p.npz")
    np.savez_compressed(path, width=W, height=H, nframes=T, seed=SEED, sha256=sha, white=white, thr=np.array(thr, np.int32))
    print(path, os.path.getsize(path), "bytes; thresholds", thr, "white", white.tolist())


if __name__ == "__main__":
    main()
