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
from oracle import pyoracle as po  # noqa: E402  (only for the Gaussian kernel values)


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
    core = CUDACore(W, H, k=po.gaussian_kernel(3, 1.5), max_batch=B)
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
    cpu_reference_line(W, H)


def cpu_reference_line(W, H, T=16, seconds=8.0):
    """The reference's own CPU branch for the gray-avg -> histogram -> two-max -> binarize chain
    (server/src/server.cpp:96-135, compiled unmodified into oracle/_ref/server_cpu), timed by the
    reference's own per-frame `FOR:` counter (server.cpp:77,144,164) on this host, next to the GPU lines."""
    import re
    import subprocess
    import tempfile

    import numpy as np
    exe = po.ref_server_cpu_path()
    if exe is None:
        print(json.dumps({"filter": "reference server.cpp CPU branch", "skipped": "oracle/_ref/server_cpu not built"}))
        return
    base, frames = synth.webcam_stream(T, W, H)
    with tempfile.TemporaryDirectory() as tmp:
        fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
        with open(fin, "wb") as f:
            f.write(np.array([W, H, T], np.int32).tobytes())
            f.write(base.tobytes())
            f.write(frames.tobytes())
        # the reference prints its counters once per second (server.cpp:151): loop the sequence long enough
        # for several prints (about 12 ms per 1080p frame sizes the repeat count)
        repeat = max(2, int(seconds / (T * 0.012 * (W * H) / (1920 * 1080))))
        r = subprocess.run([exe], env=dict(os.environ, REF_IN=fin, REF_OUT=fout, REF_REPEAT=str(repeat)),
                           check=True, stdout=subprocess.PIPE, timeout=600)
    ms = [float(m) for m in re.findall(r"FOR:\s*([0-9.]+) ms", r.stdout.decode(errors="replace"))]
    if not ms:
        print(json.dumps({"filter": "reference server.cpp CPU branch", "skipped": "no FOR: lines in its output"}))
        return
    ms.sort()
    med = ms[len(ms) // 2]
    print(json.dumps({"filter": "reference server.cpp CPU branch (gray avg + histogram + two-max + binarize)",
                      "kind": "reference", "cores": 1, "ms_per_frame_median": round(med, 3),
                      "frames_per_s": round(1e3 / med, 1), "samples": len(ms), "frames_processed": T * repeat,
                      "timer": "the reference's own FOR: counter, server.cpp:164"}), flush=True)


if __name__ == "__main__":
    main()
