"""Helpers for the `-m gpu` parity tests: every call goes through the C-ABI (libmi355diff.so)."""
import numpy as np
import torch

from cudavideostream_amd import CUDACore as _CUDACore

DEV = "cuda:0"

_DEVICE_CALLS = {"diff_stream_batch", "diff_pairs_batch", "diff_stream_wire_batch", "apply_batch",
                 "apply_wire_batch", "merge_parts", "int_diff", "gray_avg", "gray_weighted", "binarize_chain",
                 "heat_map", "red_dense", "red_overlap", "red_stream_batch", "conv3x3", "median5x5", "filter_batch"}


class CUDACore(_CUDACore):
    """The product class with one test-side addition: a core runs on a non-blocking stream of its own, which is NOT
    ordered against torch's streams (include/mi355diff.h, "Streams"), so buffers the tests fill with torch (uploads,
    torch.full on torch's stream) must be complete before a device-resident entry point is enqueued.  Every such entry
    point is wrapped: torch's streams are drained when the method is CALLED -- not when it is looked up (round 3's
    harness did that, and a method bound before the last torch.full raced with the fill: 1 failure in 12 runs)."""


def _synced(name):
    inner = getattr(_CUDACore, name)

    def call(self, *args, **kwargs):
        torch.cuda.synchronize()
        # (temporaries handed in -- core.conv3x3(to_dev(a), out) -- are kept alive by the product class itself until
        # synchronize(): cudavideostream_amd/core.py, _hold)
        return inner(self, *args, **kwargs)

    call.__name__ = name
    call.__doc__ = inner.__doc__
    return call


for _name in _DEVICE_CALLS:
    setattr(CUDACore, _name, _synced(_name))


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def run_stream(core, frames, capacity=None, stride=None, pair_prev=None):
    """frames: (T, N) uint8 numpy or cuda tensor.  Returns (offsets, xs, diff) as numpy, with xs/diff
    cut to min(total, capacity)."""
    d_frames = frames if torch.is_tensor(frames) else to_dev(frames)
    T = d_frames.shape[0]
    n = core.total
    cap = T * n if capacity is None else capacity
    d_off = torch.full((T + 1,), 0xFFFFFFFF, dtype=torch.int64, device=DEV).to(torch.int32)
    d_xs = torch.full((max(cap, 1),), -7, dtype=torch.int32, device=DEV)
    d_df = torch.full((max(cap, 1),), 0xA5, dtype=torch.uint8, device=DEV)
    if pair_prev is None:
        core.diff_stream_batch(d_frames, T, d_off, d_xs, d_df, cap, stride=stride)
    else:
        d_prev = pair_prev if torch.is_tensor(pair_prev) else to_dev(pair_prev)
        core.diff_pairs_batch(d_frames, d_prev, T, d_off, d_xs, d_df, cap, stride=stride)
    core.synchronize()
    off = d_off.cpu().numpy().view(np.uint32)
    tot = min(int(off[-1]), cap)
    return off, d_xs[:tot].cpu().numpy(), d_df[:tot].cpu().numpy(), (d_xs, d_df)


def oracle_pairs(po, cur, prev, thr=20):
    offs, xs, df = [0], [], []
    for t in range(cur.shape[0]):
        c, x, d, _ = po.diff_pack(cur[t], prev[t], thr)
        offs.append(offs[-1] + c); xs.append(x); df.append(d)
    return (np.array(offs, np.uint32), np.concatenate(xs) if xs else np.empty(0, np.int32),
            np.concatenate(df) if df else np.empty(0, np.uint8))
