#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ (run from the repo root).

Two kinds of fixture:
  * ref_server_cpu_64x48.npz  -- outputs of the REFERENCE ITSELF: its server/src/server.cpp CPU
    branch (gray-avg -> histogram -> two-max -> binarize, server.cpp:96-135) compiled unmodified
    into oracle/_ref/server_cpu and driven with the seeded frames stored in the same file.
    Needs /root/reference (only available in the build container).
  * oracle_*.npz -- outputs of the CPU restatement oracle/cpu_ref.c on seeded inputs, for the rows
    the reference pins no numbers for.  Inputs are stored with the outputs.
The fixtures are data only (inputs + expected outputs).
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cudavideostream_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
W, H = 64, 48


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path)} bytes")


def main():
    po.build()
    rng = np.random.default_rng(20261003)

    # -- the reference's own CPU branch --------------------------------------------------------
    if po.ref_server_cpu_path():
        base, frames = synth.webcam_stream(6, W, H, seed=5)
        frames = frames.copy()
        frames[1] = rng.integers(0, 256, frames[1].shape, dtype=np.uint8)       # flat histogram
        frames[2] = (rng.integers(0, 40, frames[2].shape) + 10).astype(np.uint8)  # dark: clamp 50
        frames[3] = (rng.integers(0, 30, frames[3].shape) + 220).astype(np.uint8)  # bright: clamp 200
        frames[4][::2] = 17                                                       # two peaks
        with tempfile.TemporaryDirectory() as d:
            out = po.run_ref_server_cpu(base, frames, W, H, d)
        save("ref_server_cpu_64x48.npz", width=W, height=H, base=base, frames=frames, out=out)
    else:
        print("oracle/_ref/server_cpu not built (no /root/reference): keeping the committed fixture")

    # -- diff/threshold/pack: S1 stream, S4 edge strip, ragged size ------------------------------
    base, frames = synth.webcam_stream(5, W, H, seed=21)
    offsets, xs, df, st = po.diff_stream(frames, base)
    save("oracle_diff_stream_64x48.npz", width=W, height=H, base=base, frames=frames,
         offsets=offsets, xs=xs, diff=df, state=st)

    cur, prev = synth.edge_strip()
    c, xs, df, st = po.diff_pack(cur, prev)
    save("oracle_diff_edge_strip.npz", count=c, xs=xs, diff=df, state=st)

    n = 3 * 37 * 11  # 1221 bytes: not a multiple of 16 nor of 1024
    a = rng.integers(0, 256, n, dtype=np.uint8)
    b = np.clip(a.astype(np.int32) + rng.integers(-30, 31, n), 0, 255).astype(np.uint8)
    c, xs, df, st = po.diff_pack(a, b)
    save("oracle_diff_ragged_37x11.npz", cur=a, prev=b, count=c, xs=xs, diff=df, state=st)

    # -- filters ----------------------------------------------------------------------------------
    save("oracle_heat_lut.npz", lut=po.heat_lut())
    save("oracle_gaussian_k3.npz", k=po.gaussian_kernel(3, 1.5))
    img = synth.webcam_frame(3, W, H, seed=9)
    prv = synth.webcam_frame(2, W, H, seed=9)
    k = po.gaussian_kernel(3, 1.5)
    gw = po.gray_weighted(img)
    save("oracle_filters_64x48.npz", width=W, height=H, img=img, prev=prv, k=k,
         gray_avg=po.gray_avg(img), gray_weighted=gw, hist=po.histogram(gw),
         thr=po.two_max_threshold(po.histogram(gw)),
         binarized=po.binarize(gw, po.two_max_threshold(po.histogram(gw))),
         heat=po.heat_map(img, prv), red=po.red_dense(img, prv), conv=po.conv3x3(img, W, H, k))
    # weighted gray: a 4096-triple sample of the exhaustive 2^24 table + its checksum
    bgr = rng.integers(0, 256, (4096, 3), dtype=np.uint8)
    g = po.gray_weighted(bgr.reshape(-1)).reshape(-1, 3)[:, 0]
    save("oracle_gray_weighted_sample.npz", bgr=bgr, gray=g)


if __name__ == "__main__":
    main()
