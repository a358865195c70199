"""world_size-2 (and 3) CPU rehearsal of the multi-GPU exchange step (cudavideostream_amd/gather.py)
over the gloo backend: every rank owns an independent stream, the per-frame index and the
changed-pixel payload are gathered to rank 0 and must equal the ranks' oracle outputs concatenated
in rank order -- the same code path bench.py runs over RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cudavideostream_amd import gather as gx
from cudavideostream_amd import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank_stream(rank, T, w, h):
    from oracle import pyoracle as po
    base, frames = synth.webcam_stream(T, w, h, seed=21 + rank)
    if rank == 1:
        frames = frames.copy()
        frames[1] = base          # a frame with (almost) nothing to send
    return po.diff_stream(frames, base)


def _worker(rank, world, port, T, w, h, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        off, xs, df, _ = _rank_stream(rank, T, w, h)
        cap = xs.size + 100
        t_off = torch.from_numpy(off.view(np.int32).copy())
        t_xs = torch.zeros(cap, dtype=torch.int32); t_xs[:xs.size] = torch.from_numpy(xs)
        t_df = torch.zeros(cap, dtype=torch.uint8); t_df[:df.size] = torch.from_numpy(df)
        idx = gx.gather_index(t_off, dst=0)
        totals, xs_all, df_all, index = gx.gather_payload(t_off, t_xs, t_df, dst=0)
        if rank == 0:
            q.put((idx.numpy().copy(), totals, xs_all.numpy().copy(), df_all.numpy().copy(),
                   index.numpy().copy()))
        else:
            assert idx is None and xs_all is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_payload_and_index(world):
    T, w, h = 4, 48, 32
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, T, w, h, q)) for r in range(world)]
    for p in procs:
        p.start()
    idx, totals, xs_all, df_all, index = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    exp = [_rank_stream(r, T, w, h) for r in range(world)]
    assert totals == [int(e[0][-1]) for e in exp]
    assert np.array_equal(idx.view(np.uint32), np.stack([e[0] for e in exp]))
    assert np.array_equal(index, idx)
    assert np.array_equal(xs_all, np.concatenate([e[1] for e in exp]))
    assert np.array_equal(df_all, np.concatenate([e[2] for e in exp]))
    # rank r's frame t is recoverable from the gathered buffers alone
    at = 0
    for r in range(world):
        off = exp[r][0].astype(np.int64)
        for t in range(T):
            seg = slice(at + off[t], at + off[t + 1])
            assert np.array_equal(xs_all[seg], exp[r][1][off[t]:off[t + 1]])
        at += totals[r]


def _band_stream(rank, world, T, w, h):
    from oracle import pyoracle as po
    base, frames = synth.webcam_stream(T, w, h, seed=33)
    r0, r1 = gx.band_rows(rank, world, h)
    b0, b1 = 3 * w * r0, 3 * w * r1
    return b0, po.diff_stream(np.ascontiguousarray(frames[:, b0:b1]), base[b0:b1])


def _band_worker(rank, world, port, T, w, h, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        b0, (off, xs, df, _) = _band_stream(rank, world, T, w, h)
        t_off = torch.from_numpy(off.view(np.int32).copy())
        t_xs = torch.from_numpy(np.append(xs, np.int32(0)))
        t_df = torch.from_numpy(np.append(df, np.uint8(0)))
        out = gx.gather_bands(t_off, t_xs, t_df, b0, core=None, dst=0)
        if rank == 0:
            index, part_base, xs_bias, xs_all, df_all = out
            q.put((index.numpy().copy(), part_base, xs_bias, xs_all.numpy().copy(), df_all.numpy().copy()))
        else:
            assert out == (None, None, None)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_bands_of_one_stream_gather_to_the_whole_frame_stream(world):
    """E2 of SURVEY.md section 8e: ranks own row bands of the same frames; the gathered pieces, merged in
    band order with the bands' byte offsets (what mi355_merge_parts does on the device, checked against
    this same rule in tests/test_stream_ops_gpu.py), are the whole-frame stream of the oracle."""
    from oracle import pyoracle as po
    T, w, h = 5, 40, 21
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_band_worker, args=(r, world, port, T, w, h, q)) for r in range(world)]
    for p in procs:
        p.start()
    index, part_base, xs_bias, xs_all, df_all = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    base, frames = synth.webcam_stream(T, w, h, seed=33)
    off, xs, df, _ = po.diff_stream(frames, base)
    assert xs_bias == [3 * w * gx.band_rows(r, world, h)[0] for r in range(world)]
    idx = index.view(np.uint32).astype(np.int64)
    m_xs, m_df, m_off = [], [], [0]
    for t in range(T):
        for p in range(world):
            seg = slice(part_base[p] + idx[p, t], part_base[p] + idx[p, t + 1])
            m_xs.append(xs_all[seg] + xs_bias[p]); m_df.append(df_all[seg])
        m_off.append(m_off[-1] + int((idx[:, t + 1] - idx[:, t]).sum()))
    assert np.array_equal(np.array(m_off, np.uint32), off)
    assert np.array_equal(np.concatenate(m_xs), xs)
    assert np.array_equal(np.concatenate(m_df), df)


def _rr_local(rank, world, T, w, h):
    """Rank-local work of BASELINE config 5: frames t = rank (mod world) of one sequence against their raw
    predecessors (pair mode), padded with empty frames to the common local batch length."""
    from oracle import pyoracle as po
    base, frames = synth.webcam_stream(T, w, h, seed=55)
    seq = np.concatenate([base[None, :], frames])            # seq[t + 1] = frame t, seq[t] its predecessor
    mine = gx.roundrobin_frames(rank, world, T)
    B = (T + world - 1) // world
    offs, xs, df = [0], [], []
    for t in mine:
        c, x, d, _ = po.diff_pack(seq[t + 1], seq[t])
        offs.append(offs[-1] + c); xs.append(x); df.append(d)
    while len(offs) < B + 1:
        offs.append(offs[-1])
    return (np.array(offs, np.uint32), np.concatenate(xs) if xs else np.empty(0, np.int32),
            np.concatenate(df) if df else np.empty(0, np.uint8))


def _rr_worker(rank, world, port, T, w, h, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        off, xs, df = _rr_local(rank, world, T, w, h)
        t_off = torch.from_numpy(off.view(np.int32).copy())
        t_xs = torch.from_numpy(np.append(xs, np.int32(0)))
        t_df = torch.from_numpy(np.append(df, np.uint8(0)))
        totals, xs_all, df_all, index = gx.gather_payload(t_off, t_xs, t_df, dst=0)
        if rank == 0:
            q.put((gx.roundrobin_order(index, T), xs_all.numpy().copy(), df_all.numpy().copy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,T", [(2, 7), (3, 7), (3, 2)])
def test_round_robin_frames_of_one_sequence(world, T):
    """BASELINE config 5 (frames dealt round-robin to the GPUs, gather to one rank): the root finds every
    frame's segment in frame order, equal to the single-process oracle on the whole sequence."""
    from oracle import pyoracle as po
    w, h = 40, 12
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rr_worker, args=(r, world, port, T, w, h, q)) for r in range(world)]
    for p in procs:
        p.start()
    order, xs_all, df_all = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    base, frames = synth.webcam_stream(T, w, h, seed=55)
    seq = np.concatenate([base[None, :], frames])
    assert len(order) == T
    for t, (a, b) in enumerate(order):
        c, x, d, _ = po.diff_pack(seq[t + 1], seq[t])
        assert b - a == c
        assert np.array_equal(xs_all[a:b], x) and np.array_equal(df_all[a:b], d)


def test_single_process_passthrough():
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        off = torch.tensor([0, 2, 5], dtype=torch.int32)
        xs = torch.arange(8, dtype=torch.int32)
        df = torch.arange(8, dtype=torch.uint8)
        totals, a, b, index = gx.gather_payload(off, xs, df)
        assert totals == [5] and a.tolist() == [0, 1, 2, 3, 4] and b.tolist() == [0, 1, 2, 3, 4]
        assert gx.gather_index(off).shape == (1, 3)
    finally:
        dist.destroy_process_group()


# ---- bench.py's own exchange object and config 5's round-robin bookkeeping, on the CPU ---------------------------------------
def _config5_worker(rank, world, port, B, w, h, q):
    """What bench.py::config5_across_ranks does with the GPU taken out: ONE sequence dealt round-robin over the ranks, every
    rank's shard diffed against its raw predecessors (here by the oracle), bench.Exchange to rank 0 (its torch.distributed
    form: no RCCL group), Exchange.verify, and rank 0's parity check of ITS copy through gather.roundrobin_order."""
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        from oracle import pyoracle as po
        n = 3 * w * h
        mine = gx.roundrobin_frames(rank, world, B * world)
        offs, xs_l, df_l = [0], [], []
        for t in mine:
            c, xs, df, _ = po.diff_pack(synth.webcam_frame(t, w, h, seed=31), synth.webcam_frame(t - 1, w, h, seed=31))
            offs.append(offs[-1] + c); xs_l.append(xs); df_l.append(df)
        total = offs[-1]
        cap = total + 64
        d_off = torch.tensor(offs, dtype=torch.int64).to(torch.int32)
        d_xs = torch.zeros(cap, dtype=torch.int32); d_xs[:total] = torch.from_numpy(np.concatenate(xs_l))
        d_df = torch.zeros(cap, dtype=torch.uint8); d_df[:total] = torch.from_numpy(np.concatenate(df_l))
        cpu = torch.device("cpu")
        xch = bench.Exchange(None, dist, world, rank, B, cap, cpu, cpu)
        ms = xch.run(d_off, d_xs, d_df)
        ok = xch.verify(d_off, d_xs, d_df)
        assert ms >= 0 and xch.calls == 1 and xch.ranks_seen == world
        if rank == 0:
            assert ok is True
            index, xs_all, df_all = xch.root
            seg = gx.roundrobin_order(index, B * world)
            par = True
            for t in range(B * world):      # every frame of the sequence, not only the shards' first and last
                c, xs, df, _ = po.diff_pack(synth.webcam_frame(t, w, h, seed=31), synth.webcam_frame(t - 1, w, h, seed=31))
                a, b = seg[t]
                par = par and b - a == c and np.array_equal(xs_all[a:b].numpy(), xs) and np.array_equal(df_all[a:b].numpy(), df)
            # a corrupted copy must be noticed by verify()
            xs_all[0] += 1
            q.put((par, xch.bytes, xch.verify(d_off, d_xs, d_df)))
        else:
            assert ok is None
            xch.verify(d_off, d_xs, d_df)      # (collective: rank 0 calls it a second time)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_bench_exchange_and_config5_bookkeeping_on_cpu(world):
    B, w, h = 3, 48, 32
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_config5_worker, args=(r, world, port, B, w, h, q)) for r in range(world)]
    for p in procs:
        p.start()
    par, nbytes, corrupted = q.get(timeout=180)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert par is True and nbytes > 4 * (B + 1) * world and corrupted is False
