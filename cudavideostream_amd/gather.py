"""Multi-GPU exchange step: gather-v of the changed-pixel streams to one rank.

The reference has no multi-GPU code; its only "exchange" is the sender thread
that writes {u32 n, i32 xs[n], u8 diff[n]} per frame to one socket (server/src/threads.cpp:223-233).
With one process per GPU, every rank runs an independent stream (or an independent set of frame
pairs) and the diff/threshold/pack path needs NO collective; the single exchange is this gather of
the per-frame index (offsets) and, where a single consumer wants it, of the payload.

Collectives are torch.distributed calls: backend "nccl" is RCCL over xGMI on the GPU box, "gloo" in
the CPU tests.  The payload moves as grouped point-to-point sends/receives (one per peer, all
concurrent: 7 direct xGMI links into the root), not as a ring collective.
"""
import torch
import torch.distributed as dist


def gather_index(offsets, dst=0):
    """offsets: uint32-as-int32 [B+1] exclusive scan of one rank's per-frame counts.
    Returns on dst an int32 tensor [world, B+1]; None elsewhere."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    if world == 1:
        return offsets.unsqueeze(0)
    if rank == dst:
        out = torch.empty((world,) + tuple(offsets.shape), dtype=offsets.dtype, device=offsets.device)
        parts = list(out.unbind(0))
        dist.gather(offsets, parts, dst=dst)
        return out
    dist.gather(offsets, None, dst=dst)
    return None


def gather_payload(offsets, xs, diff, dst=0):
    """Gather-v of (xs[:total], diff[:total]) of every rank to dst, concatenated in rank order.

    Returns on dst (totals[world] (python ints), xs_all, diff_all, index[world, B+1]); on other ranks
    (totals, None, None, None).  Synchronises the host once to learn the local total (the reference
    also reads its count back before the copies, kernels.cu:507-508)."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    total = int(offsets[-1].item()) & 0xFFFFFFFF
    if world == 1:
        return [total], xs[:total], diff[:total], offsets.unsqueeze(0)
    t_local = torch.tensor([total], dtype=torch.int64, device=offsets.device)
    t_all = [torch.zeros_like(t_local) for _ in range(world)]
    dist.all_gather(t_all, t_local)
    totals = [int(t.item()) for t in t_all]
    index = gather_index(offsets, dst)
    if rank == dst:
        n_all = sum(totals)
        xs_all = torch.empty(max(n_all, 1), dtype=xs.dtype, device=xs.device)
        df_all = torch.empty(max(n_all, 1), dtype=diff.dtype, device=diff.device)
        ops, at = [], 0
        for r in range(world):
            c = totals[r]
            if r == dst:
                xs_all[at:at + c] = xs[:c]
                df_all[at:at + c] = diff[:c]
            elif c:
                ops.append(dist.P2POp(dist.irecv, xs_all[at:at + c], r))
                ops.append(dist.P2POp(dist.irecv, df_all[at:at + c], r))
            at += c
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return totals, xs_all[:n_all], df_all[:n_all], index
    if total:
        ops = [dist.P2POp(dist.isend, xs[:total], dst), dist.P2POp(dist.isend, diff[:total], dst)]
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return totals, None, None, None


# ---- E2: one stream split into row bands ------------------------------------------
def band_rows(rank, world, height):
    """Rows [first, last) of the band rank `rank` owns; bands are contiguous byte ranges of the frame."""
    return height * rank // world, height * (rank + 1) // world


def gather_bands(offsets, xs, diff, band_first_byte, core=None, dst=0):
    """Every rank packed its row band of the SAME frames (a core of the band's height fed with
    frames + band_first_byte, stride = whole-frame bytes).  Gathers the bands to dst and, when `core`
    is given there, merges them on the device (mi355_merge_parts) into the stream of the whole frame.

    Returns on dst (offsets[T+1], xs, diff) of the merged stream when core is given, else the raw pieces
    (index[world, T+1], part_base, xs_bias, xs_all, diff_all); (None,)*3 on the other ranks."""
    world = dist.get_world_size()
    rank = dist.get_rank()
    b_local = torch.tensor([int(band_first_byte)], dtype=torch.int64, device=offsets.device)
    if world > 1:
        b_all = [torch.zeros_like(b_local) for _ in range(world)]
        dist.all_gather(b_all, b_local)
        xs_bias = [int(b.item()) for b in b_all]
    else:
        xs_bias = [int(band_first_byte)]
    totals, xs_all, df_all, index = gather_payload(offsets, xs, diff, dst)
    if rank != dst:
        return None, None, None
    part_base = [sum(totals[:p]) for p in range(world)]
    if core is None:
        return index, part_base, xs_bias, xs_all, df_all
    nframes = offsets.numel() - 1
    total = sum(totals)
    out_off = torch.empty_like(offsets)
    out_xs = torch.empty(max(total, 1), dtype=xs.dtype, device=xs.device)
    out_df = torch.empty(max(total, 1), dtype=diff.dtype, device=diff.device)
    core.merge_parts(index.contiguous(), part_base, xs_bias, xs_all, df_all, nframes, out_off, out_xs, out_df,
                     total)
    return out_off, out_xs[:total], out_df[:total]


# ---- E1, round-robin: frames of ONE sequence dealt to the ranks (BASELINE config 5) ----------------------
def roundrobin_frames(rank, world, nframes):
    """Global frame numbers rank `rank` processes: t = rank, rank + world, ... (stateless pair mode: every
    frame travels with its own predecessor, so the ranks do not depend on each other)."""
    return list(range(rank, nframes, world))


def roundrobin_order(index, nframes):
    """index[world, B+1] (gather_index / gather_payload on the root, B = ceil(nframes / world) local frames,
    ranks with fewer frames pad with empty ones) -> for every global frame t the pair
    (first entry, last entry) of its segment inside the rank-major gathered arrays, in frame order."""
    world = index.shape[0]
    idx = index.to(torch.int64) & 0xFFFFFFFF
    base = torch.cumsum(idx[:, -1], 0) - idx[:, -1]          # rank r's entries start here
    out = []
    for t in range(nframes):
        r, k = t % world, t // world
        out.append((int(base[r] + idx[r, k]), int(base[r] + idx[r, k + 1])))
    return out
