#!/usr/bin/env python3
"""Secondary benchmark: the filter kernels (K2..K5) on batches of 1080p frames resident in HBM.

Prints one JSON line per filter: microseconds per frame, algorithmic bytes per frame (DESIGN.md
section 4) and the achieved GB/s against the 8 TB/s HBM peak.  Not the headline metric (bench.py)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cudavideostream_amd import CUDACore, lib, synth  # noqa: E402


def gaussian3(sigma):
    """A normalised 3x3 Gaussian (symmetric, like the server's; the exact values do not matter for timing)."""
    import numpy as np
    g = np.exp(-(np.arange(-1, 2)[:, None] ** 2 + np.arange(-1, 2)[None, :] ** 2) / (2.0 * sigma * sigma))
    return (g / g.sum()).astype(np.float32).reshape(-1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--batch", type=int, default=96)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    W, H, B = a.width, a.height, a.batch
    n = 3 * W * H
    _, frames = synth.webcam_stream(B + 1, W, H, device=dev)
    cur, prev = frames[1:], frames[:-1]
    out = torch.empty((B, n), dtype=torch.uint8, device=dev)
    core = CUDACore(W, H, k=gaussian3(1.5), max_batch=B)
    core.use_torch_stream()
    N = float(n)
    ops = [
        ("gray_avg", lib.OP_GRAY_AVG, None, 2 * N),
        ("gray_weighted", lib.OP_GRAY_WEIGHTED, None, 2 * N),
        ("binarize (gray3 in)", lib.OP_BINARIZE, None, N / 3 + 2 * N),
        ("gray_weighted+binarize fused (config 3)", lib.OP_GRAY_WEIGHTED_BINARIZE, None, 3 * N),
        ("heat_map", lib.OP_HEAT_MAP, prev, 3 * N),
        ("red_dense", lib.OP_RED_DENSE, prev, 3 * N),
        ("conv3x3", lib.OP_CONV3X3, None, 2 * N),
        ("median5x5", lib.OP_MEDIAN5X5, None, 2 * N),
    ]
    for name, op, second, alg in ops:
        for _ in range(3):
            core.filter_batch(op, cur, out, B, d_in2=second)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(a.reps):
            core.filter_batch(op, cur, out, B, d_in2=second)
        ev1.record()
        torch.cuda.synchronize()
        us = ev0.elapsed_time(ev1) * 1e3 / (a.reps * B)
        gbps = alg / (us * 1e-6) / 1e9
        print(json.dumps({"filter": name, "us_per_frame": round(us, 3), "frames_per_s": round(1e6 / us, 1),
                          "algorithmic_bytes_per_frame": int(alg), "achieved_gbps": round(gbps, 1),
                          "frac_of_8TBps": round(gbps / 8000.0, 4), "batch": B}), flush=True)
    # BASELINE configs 3 and 4 end to end (what the server does with a frame when the filter is enabled,
    # kernels.cu:457-520), on the same resident batch: the visualiser's frame AND the packed diff stream
    cap = B * n // 4
    d_off = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    d_xs = torch.empty(cap, dtype=torch.int32, device=dev)
    d_df = torch.empty(cap, dtype=torch.uint8, device=dev)
    filt = torch.empty((B, n), dtype=torch.uint8, device=dev)

    def config3():   # weighted gray + histogram + two-max + binarize (visualiser 5), then diff+threshold+pack
        core.filter_batch(lib.OP_GRAY_WEIGHTED_BINARIZE, cur, out, B)
        core.diff_stream_batch(cur, B, d_off, d_xs, d_df, cap)

    def config4():   # 3x3 noise filter, diff+threshold+pack of the filtered frames, red motion map
        core.filter_batch(lib.OP_CONV3X3, cur, filt, B)
        core.diff_stream_batch(filt, B, d_off, d_xs, d_df, cap)
        core.red_stream_batch(d_off, d_xs, B, out)   # from the packed indices, as kernels.cu:513 does

    for name, fn in (("config 3: gray-weighted + binarize + diff/threshold/pack", config3),
                     ("config 4: noise filter + diff/threshold/pack + red motion map", config4)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(a.reps):
            fn()
        ev1.record()
        torch.cuda.synchronize()
        us = ev0.elapsed_time(ev1) * 1e3 / (a.reps * B)
        print(json.dumps({"chain": name, "us_per_frame": round(us, 3), "frames_per_s": round(1e6 / us, 1),
                          "batch": B, "changed_bytes_per_frame": round((int(d_off[B].item()) & 0xFFFFFFFF) / B, 1)}),
              flush=True)
    core.close()


if __name__ == "__main__":
    main()
