#!/usr/bin/env python3
"""Chain soak (not part of the test suite; tests/test_diff_pack_gpu.py runs a short one): random sequences of entry points on
ONE long-lived core, queued on the core's own stream WITHOUT host synchronisation between them, through a few scratch
buffers that are reused all the time -- frame filter into a scratch, stream batch out of that scratch, pair batches of
both operand forms, dense batches (after which the library stops overlapping batches until the input is sparse again),
the red map of a batch's packed stream, switches between the core's stream and a caller's.  A round
is synchronised once at its end and every output of the round compared with the oracle.  What it is after: ordering
between the streams a pipelined batch uses inside the library (core.hip, run_batch / use_device), not arithmetic.
    python tests/soak_chain.py [rounds] [seed]      exits non-zero on the first mismatch"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cudavideostream_amd import CUDACore, lib, synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402  (checker)

DEV = "cuda:0"


def oracle_pairs(cur, prev):
    offs, xs, df = [0], [], []
    for t in range(cur.shape[0]):
        c, x, d, _ = po.diff_pack(cur[t], prev[t], 20)
        offs.append(offs[-1] + c); xs.append(x); df.append(d)
    return (np.array(offs, np.uint32), np.concatenate(xs) if xs else np.empty(0, np.int32),
            np.concatenate(df) if df else np.empty(0, np.uint8))


def run(rounds, seed, w=320, h=180, T=5, verbose=True, flags=0):
    rng = np.random.default_rng(seed)
    n = 3 * w * h
    k9 = po.gaussian_kernel(3, 1.5)
    base, pool = synth.webcam_stream(64, w, h, seed=seed + 1)
    pool = np.ascontiguousarray(pool)
    d_pool = torch.from_numpy(pool).to(DEV)
    noise = rng.integers(0, 256, (2 * T, n), dtype=np.uint8)   # dense input: nearly every byte changes (the library then stops
    d_noise = torch.from_numpy(noise).to(DEV)                  # overlapping batches once such a batch's total has arrived)
    scratch = torch.empty((T, n), dtype=torch.uint8, device=DEV)      # filter output = batch input, rewritten every time
    vis = torch.empty((T, n), dtype=torch.uint8, device=DEV)
    nout = 6
    outs = [(torch.zeros(T + 1, dtype=torch.int32, device=DEV), torch.empty(T * n, dtype=torch.int32, device=DEV),
             torch.empty(T * n, dtype=torch.uint8, device=DEV)) for _ in range(nout)]
    core = CUDACore(w, h, k=k9, max_batch=T, sample_mat_data=base, flags=flags)
    state = base.copy()
    own = True
    for rnd in range(rounds):
        checks = []   # (what, output slot, expectation)
        red_expect = None
        nops = int(rng.integers(2, nout + 1))
        torch.cuda.synchronize()
        for i in range(nops):
            op = int(rng.integers(0, 7))
            if int(rng.integers(0, 24)) == 0:
                op = 7       # (rarely: the oracle sorts 25 bytes per output byte)
            if int(rng.integers(0, 16)) == 0:   # (round 6) mi355_prepare in the middle of queued work: blocking, changes nothing
                mask = int(rng.integers(1, 32))
                if os.environ.get("SOAK_PREPARE", "1") != "0":
                    core.prepare(mask)
            if int(rng.integers(0, 40)) == 0:   # (round 6) the index kernel's launch tag put in front of its wrap: the totals are cleared
                left = int(rng.integers(1, 4))
                if os.environ.get("SOAK_EPOCH", "1") != "0":
                    core.set_option(lib.OPT_SCAN_EPOCH_LEFT, left)   # (set_option completes what is queued)
            f0 = int(rng.integers(0, 64 - 2 * T))
            o = outs[i]
            if op == 0:      # stream batch straight from the pool
                core.diff_stream_batch(d_pool[f0:f0 + T], T, *o, T * n)
                eo, exs, edf, state = po.diff_stream(pool[f0:f0 + T], state)
                checks.append(("stream", o, (eo, exs, edf)))
            elif op == 1:    # noise filter into the scratch, stream batch out of it
                core.filter_batch(lib.OP_CONV3X3, d_pool[f0:f0 + T], scratch, T)
                core.diff_stream_batch(scratch, T, *o, T * n)
                filt = np.stack([po.conv3x3(f, w, h, k9) for f in pool[f0:f0 + T]])
                eo, exs, edf, state = po.diff_stream(filt, state)
                checks.append(("conv+stream", o, (eo, exs, edf)))
            elif op == 7:    # 5x5 median (column-strip kernel: 960-byte rows) into the scratch, stream batch out of it
                core.filter_batch(lib.OP_MEDIAN5X5, d_pool[f0:f0 + T], scratch, T)
                core.diff_stream_batch(scratch, T, *o, T * n)
                filt = np.stack([po.median5x5(f, w, h) for f in pool[f0:f0 + T]])
                eo, exs, edf, state = po.diff_stream(filt, state)
                checks.append(("median+stream", o, (eo, exs, edf)))
            elif op == 2:    # pairs of consecutive frames
                core.diff_pairs_batch(d_pool[f0 + 1:f0 + T + 1], d_pool[f0:f0 + T], T, *o, T * n)
                checks.append(("pairs", o, oracle_pairs(pool[f0 + 1:f0 + T + 1], pool[f0:f0 + T])))
            elif op == 3:    # pairs that share no frame
                blk, hb = d_pool[f0:f0 + 2 * T], pool[f0:f0 + 2 * T]
                core.diff_pairs_batch(blk[1::2], blk[0::2], T, *o, T * n, stride=2 * n)
                checks.append(("pairs apart", o, oracle_pairs(hb[1::2], hb[0::2])))
            elif op == 4 and checks and checks[-1][0] in ("stream", "conv+stream", "median+stream"):   # red map of the batch before
                prev_o = checks[-1][1]
                core.red_stream_batch(prev_o[0], prev_o[1], T, vis, True)
                eo, exs, _ = checks[-1][2]
                red_expect = np.zeros((T, n), np.uint8)
                for t in range(T):
                    red_expect[t] = po.red_overlap(red_expect[t], exs[eo[t]:eo[t + 1]])
            elif op == 6:    # a dense stream batch
                g0 = int(rng.integers(0, T + 1))
                core.diff_stream_batch(d_noise[g0:g0 + T], T, *o, T * n)
                eo, exs, edf, state = po.diff_stream(noise[g0:g0 + T], state)
                checks.append(("dense stream", o, (eo, exs, edf)))
            else:            # switch streams (a synchronising call by contract)
                own = not own
                (core.use_own_stream if own else core.use_torch_stream)()
        core.synchronize()
        torch.cuda.synchronize()
        for what, o, exp in checks:
            eo, exs, edf = exp
            tot = int(eo[-1])
            ok = (np.array_equal(o[0].cpu().numpy().view(np.uint32), eo) and np.array_equal(o[1][:tot].cpu().numpy(), exs)
                  and np.array_equal(o[2][:tot].cpu().numpy(), edf))
            if not ok:
                print(f"MISMATCH in round {rnd} ({what}); seed {seed}")
                return False
        if red_expect is not None and not np.array_equal(vis.cpu().numpy(), red_expect):   # the round's last red map
            print(f"MISMATCH in round {rnd} (red map); seed {seed}")
            return False
        if not np.array_equal(core.get_state(), state):
            print(f"MISMATCH in round {rnd} (state); seed {seed}")
            return False
        if verbose and rnd % 25 == 0:
            print(f"round {rnd} ok", flush=True)
    core.close()
    return True


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 11
    ok = (run(rounds, seed) and run(max(rounds // 4, 1), seed + 100, w=640, h=360, T=4)
          and run(max(rounds // 4, 1), seed + 200, flags=lib.FLAG_OWN_QUEUES))   # the core's streams in their own priority class
    print("chain soak ok" if ok else "chain soak FAILED")
    sys.exit(0 if ok else 1)
