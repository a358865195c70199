#!/usr/bin/env python3
"""Secondary benchmark: the diff/threshold/pack pipeline across input regimes at 1080p,
device-resident batches: S2 static (P = 0), S1 webcam (P ~ 2.3 %), pairs of S0 refrand (P ~ 84.6 %),
S3 flip (P = N).  One JSON line per regime: frames/s of the whole pipeline and kernel milliseconds."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cudavideostream_amd import CUDACore, synth  # noqa: E402


def run(name, core, fn, B, n, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    core.set_timing(True)
    core.reset_timing()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms_pack, ms_total, launches = core.get_timing()
    core.set_timing(False)
    return {"regime": name, "frames_per_s": round(B * reps / dt, 1), "pack_ms": round(ms_pack / launches, 4),
            "all_kernels_ms": round(ms_total / launches, 4), "batch": B}


def main():
    dev = torch.device("cuda", 0)
    W, H, B = 1920, 1080, 32
    n = 3 * W * H
    cap = B * n
    d_off = torch.zeros(B + 1, dtype=torch.int32, device=dev)
    d_xs = torch.empty(cap, dtype=torch.int32, device=dev)
    d_df = torch.empty(cap, dtype=torch.uint8, device=dev)
    core = CUDACore(W, H, max_batch=B)
    core.use_torch_stream()
    base, frames = synth.webcam_stream(B, W, H, device=dev)
    rnd = torch.stack([synth.refrand_frame(n, 100 + t, device=dev) for t in range(B + 1)])
    flip = rnd[:B] ^ 0x80
    regimes = [
        ("S2 static pairs (P=0)", lambda: core.diff_pairs_batch(frames, frames, B, d_off, d_xs, d_df, cap)),
        ("S1 webcam stream", lambda: core.diff_stream_batch(frames, B, d_off, d_xs, d_df, cap)),
        ("S0 refrand pairs", lambda: core.diff_pairs_batch(rnd[1:], rnd[:B], B, d_off, d_xs, d_df, cap)),
        ("S3 flip pairs (P=N)", lambda: core.diff_pairs_batch(flip, rnd[:B], B, d_off, d_xs, d_df, cap)),
    ]
    for name, fn in regimes:
        r = run(name, core, fn, B, n)
        r["changed_bytes_per_frame"] = round((int(d_off[B].item()) & 0xFFFFFFFF) / B, 1)
        print(json.dumps(r), flush=True)
    core.close()


if __name__ == "__main__":
    main()
