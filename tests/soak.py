#!/usr/bin/env python3
"""Soak run (not part of the test suite): thousands of small batches of random geometry and content through
one long-lived core per geometry, every result compared with the oracle.  Meant to shake out rare ordering
problems (the scan kernel's ticket, stream reuse) that a single pass of the parity tests would not meet.
    python tests/soak.py [iterations]      exits non-zero on the first mismatch"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cudavideostream_amd import CUDACore  # noqa: E402
from oracle import pyoracle as po  # noqa: E402  (checker)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    rng = np.random.default_rng(7)
    dev = "cuda:0"
    geoms = [(64, 48), (97, 13), (320, 180), (1, 1), (683, 3), (640, 360)]
    cores = {}
    for w, h in geoms:
        n = 3 * w * h
        state = rng.integers(0, 256, n, dtype=np.uint8)
        cores[(w, h)] = [CUDACore(w, h, sample_mat_data=state, max_batch=9), state]
    done = 0
    for it in range(iters):
        w, h = geoms[int(rng.integers(0, len(geoms)))]
        core, state = cores[(w, h)]
        n = 3 * w * h
        T = int(rng.integers(1, 10))
        dens = float(rng.choice([0.0, 0.003, 0.03, 0.3, 1.0]))
        frames = np.empty((T, n), np.uint8)
        prev = state
        for t in range(T):
            f = prev.astype(np.int16) + rng.integers(-5, 6, n)
            hit = rng.random(n) < dens
            f[hit] = rng.integers(0, 256, int(hit.sum()))
            frames[t] = f.clip(0, 255).astype(np.uint8)
            prev = frames[t]
        off, xs, df, st = po.diff_stream(frames, state)
        cap = int(off[-1]) + 1
        d_frames = torch.from_numpy(frames).to(dev)
        d_off = torch.full((T + 1,), -1, dtype=torch.int32, device=dev)
        d_xs = torch.empty(cap, dtype=torch.int32, device=dev)
        d_df = torch.empty(cap, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        core.diff_stream_batch(d_frames, T, d_off, d_xs, d_df, cap)
        core.synchronize()
        tot = int(off[-1])
        ok = (np.array_equal(d_off.cpu().numpy().view(np.uint32), off) and
              np.array_equal(d_xs.cpu().numpy()[:tot], xs) and np.array_equal(d_df.cpu().numpy()[:tot], df))
        if ok and it % 50 == 0:
            ok = np.array_equal(core.get_state(), st)
        if not ok:
            print(f"MISMATCH at iteration {it}: {w}x{h} T={T} density={dens}")
            sys.exit(1)
        cores[(w, h)][1] = st
        done += 1
        if it % 500 == 0:
            print(f"iteration {it} ok", flush=True)
    print(f"soak ok: {done} batches")


if __name__ == "__main__":
    main()
