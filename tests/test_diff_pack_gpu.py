"""GPU parity of K1 (diff + threshold + negative feedback + ordered pack) against the CPU oracle
(tests/cuda_streaming/test.cu:560-576 restated), bit-exact, through the C-ABI."""
import os

import numpy as np
import pytest

from conftest import golden
from cudavideostream_amd import lib, synth

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from gpu_util import DEV, CUDACore, oracle_pairs, run_stream, to_dev  # noqa: E402
from gpu_util import _CUDACore as RawCore  # noqa: E402  (the product class without the harness's call-time synchronisation)


def check_stream(po, core, base, frames, thr=20, **kw):
    core.set_state(base)
    off, xs, df, _ = run_stream(core, frames, **kw)
    eo, exs, edf, est = po.diff_stream(np.asarray(frames), base, thr)
    assert np.array_equal(off, eo), (off, eo)
    assert np.array_equal(xs, exs)
    assert np.array_equal(df, edf)
    assert np.array_equal(core.get_state(), est)
    return off


def test_golden_stream_64x48(po):
    g = golden("oracle_diff_stream_64x48.npz")
    with CUDACore(64, 48, max_batch=8) as core:
        core.set_state(g["base"])
        off, xs, df, _ = run_stream(core, g["frames"])
        assert np.array_equal(off, g["offsets"]) and np.array_equal(xs, g["xs"])
        assert np.array_equal(df, g["diff"]) and np.array_equal(core.get_state(), g["state"])


def test_edge_strip_every_byte_pair(po):
    """All 65536 (prev, cur) pairs, three times (N = 3*256*256): |df| = 20 / 21, wrap-around."""
    cur, prev = synth.edge_strip(3)
    with CUDACore(256, 256, max_batch=1) as core:
        off = check_stream(po, core, prev, cur[None, :])
        assert off[1] == 3 * int(golden("oracle_diff_edge_strip.npz")["count"])


@pytest.mark.parametrize("T", [1, 2, 3, 4, 5, 7, 8, 9, 13])
def test_batch_lengths(po, T):
    """Frame groups of 4 are double buffered: every tail length must work."""
    base, frames = synth.webcam_stream(T, 80, 60, seed=T)
    with CUDACore(80, 60, max_batch=13) as core:
        check_stream(po, core, base, frames)


def test_consecutive_batches_continue_the_stream(po):
    base, frames = synth.webcam_stream(11, 96, 54, seed=3)
    eo, exs, edf, est = po.diff_stream(frames, base)
    with CUDACore(96, 54, max_batch=6) as core:
        core.set_state(base)
        o1, x1, d1, _ = run_stream(core, frames[:6])
        o2, x2, d2, _ = run_stream(core, frames[6:])
        assert np.array_equal(np.concatenate([x1, x2]), exs)
        assert np.array_equal(np.concatenate([d1, d2]), edf)
        assert np.array_equal(np.concatenate([o1, o2[1:] + o1[-1]]), eo)
        assert np.array_equal(core.get_state(), est)


@pytest.mark.parametrize("w,h", [(37, 11), (1, 1), (5, 1), (341, 1), (342, 1), (21, 17), (683, 3)])
def test_ragged_sizes(po, w, h):
    """N not a multiple of 16 / 1024: byte-wise tail path, partial last tile."""
    rng = np.random.default_rng(w * 131 + h)
    n = 3 * w * h
    base = rng.integers(0, 256, n, dtype=np.uint8)
    frames = np.clip(base.astype(int) + rng.integers(-40, 41, (3, n)), 0, 255).astype(np.uint8)
    with CUDACore(w, h, max_batch=3) as core:
        check_stream(po, core, base, frames)
    if (w, h) == (37, 11):
        g = golden("oracle_diff_ragged_37x11.npz")
        with CUDACore(w, h, max_batch=1) as core:
            off, xs, df, _ = run_stream(core, g["cur"][None, :], pair_prev=g["prev"][None, :])
            assert off[1] == int(g["count"]) and np.array_equal(xs, g["xs"]) and np.array_equal(df, g["diff"])


def test_static_flip_and_dense(po):
    n = 3 * 128 * 64
    with CUDACore(128, 64, max_batch=2) as core:
        cur, prev = synth.static_pair(n)
        assert check_stream(po, core, prev, cur[None, :])[1] == 0          # P = 0
        cur, prev = synth.flip_pair(n)
        assert check_stream(po, core, prev, cur[None, :])[1] == n          # P = N
        a, b = synth.refrand_frame(n, 1), synth.refrand_frame(n, 2)        # S0, ~84.6 % flagged
        off = check_stream(po, core, b, np.stack([a, b]))
        assert 0.8 < off[1] / n < 0.9


@pytest.mark.parametrize("thr", [0, 1, 20, 21, 127, 128, 129, 200, 254, 255])
def test_thresholds(po, thr):
    """Every byte pair at thresholds either side of 128 (the reference's LR_THRESHOLDS is an unconstrained int,
    common.h:14): 128 and above take the HIGH form of the 4-bytes-per-instruction compare; 255 flags nothing."""
    cur, prev = synth.edge_strip(3)
    with CUDACore(256, 256, threshold=thr, max_batch=1) as core:
        off = check_stream(po, core, prev, cur[None, :], thr=thr)
        if thr == 255:
            assert off[-1] == 0


def test_threshold_range_is_0_to_255():
    for bad in (-1, 256, 1000):
        with pytest.raises(lib.Mi355Error):
            CUDACore(16, 16, threshold=bad)


def test_high_threshold_stream_with_pairs_and_ragged_tiles(po):
    """Threshold 150 on a stateful stream and on stateless pairs of a geometry with a partial last tile."""
    w, h, T = 97, 13, 5
    rng = np.random.default_rng(150)
    base = rng.integers(0, 256, 3 * w * h, dtype=np.uint8)
    frames = rng.integers(0, 256, (T, 3 * w * h), dtype=np.uint8)
    with CUDACore(w, h, threshold=150, max_batch=T) as core:
        check_stream(po, core, base, frames, thr=150)
        off, xs, df, _ = run_stream(core, frames, pair_prev=np.roll(frames, 1, axis=0))
        eo, exs, edf = oracle_pairs(po, frames, np.roll(frames, 1, axis=0), thr=150)
        assert np.array_equal(off, eo) and np.array_equal(xs, exs) and np.array_equal(df, edf)


def test_pair_mode_refrand(po):
    """tests/algorithms_benchmarks.cu style independent pairs; the core's state is untouched."""
    n = 3 * 100 * 50
    cur = np.stack([synth.refrand_frame(n, 10 + t) for t in range(5)])
    prev = np.stack([synth.refrand_frame(n, 20 + t) for t in range(5)])
    marker = synth.refrand_frame(n, 99)
    with CUDACore(100, 50, max_batch=5) as core:
        core.set_state(marker)
        off, xs, df, _ = run_stream(core, cur, pair_prev=prev)
        eo, exs, edf = oracle_pairs(po, cur, prev)
        assert np.array_equal(off, eo) and np.array_equal(xs, exs) and np.array_equal(df, edf)
        assert np.array_equal(core.get_state(), marker)


def test_pairs_of_consecutive_frames_and_of_disjoint_frames(po):
    """Pair mode picks its load policy from the operands (core.hip, run_batch): views of ONE buffer as pairs of consecutive
    frames (cur of a pair is prev of the next: the plain loads) and as pairs (0,1), (2,3), ... with a stride of two frames
    (a round-robin shard: no frame twice, the non-temporal loads), 16-byte aligned and not; both against the oracle."""
    for w, h, T in ((320, 180, 6), (97, 13, 5)):
        n = 3 * w * h
        _, frames = synth.webcam_stream(2 * T, w, h, seed=77)
        frames = np.ascontiguousarray(frames)
        buf = to_dev(frames)
        with CUDACore(w, h, max_batch=T) as core:
            off, xs, df, _ = run_stream(core, buf[1:T + 1], pair_prev=buf[0:T])
            eo, exs, edf = oracle_pairs(po, frames[1:T + 1], frames[0:T])
            assert np.array_equal(off, eo) and np.array_equal(xs, exs) and np.array_equal(df, edf)
            off, xs, df, _ = run_stream(core, buf[1::2], pair_prev=buf[0::2], stride=2 * n)
            eo, exs, edf = oracle_pairs(po, frames[1::2], frames[0::2])
            assert np.array_equal(off, eo) and np.array_equal(xs, exs) and np.array_equal(df, edf)


def test_capacity_truncation_keeps_offsets_exact(po):
    base, frames = synth.webcam_stream(4, 64, 48, seed=8)
    eo, exs, edf, _ = po.diff_stream(frames, base)
    cap = int(eo[2]) + 5  # cuts inside frame 2
    with CUDACore(64, 48, max_batch=4) as core:
        core.set_state(base)
        off, xs, df, (d_xs, d_df) = run_stream(core, frames, capacity=cap)
        assert np.array_equal(off, eo)
        assert np.array_equal(xs, exs[:cap]) and np.array_equal(df, edf[:cap])


def test_unaligned_stride_and_pointer(po):
    """stride/pointer not multiples of 16 take the byte-load path and give the same answer."""
    w, h = 64, 48
    n = 3 * w * h
    base, frames = synth.webcam_stream(3, w, h, seed=4)
    stride = n + 5
    buf = np.zeros(3 * stride + 3, np.uint8)
    for t in range(3):
        buf[3 + t * stride: 3 + t * stride + n] = frames[t]
    d_buf = to_dev(buf)
    eo, exs, edf, est = po.diff_stream(frames, base)
    with CUDACore(w, h, max_batch=3) as core:
        core.set_state(base)
        d_off = torch.zeros(4, dtype=torch.int32, device=DEV)
        d_xs = torch.zeros(3 * n, dtype=torch.int32, device=DEV)
        d_df = torch.zeros(3 * n, dtype=torch.uint8, device=DEV)
        core.diff_stream_batch(d_buf.data_ptr() + 3, 3, d_off, d_xs, d_df, 3 * n, stride=stride)
        core.synchronize()
        off = d_off.cpu().numpy().view(np.uint32)
        assert np.array_equal(off, eo)
        assert np.array_equal(d_xs[:off[-1]].cpu().numpy(), exs)
        assert np.array_equal(d_df[:off[-1]].cpu().numpy(), edf)
        assert np.array_equal(core.get_state(), est)


def test_empty_frame_and_empty_batch():
    with CUDACore(0, 0, max_batch=2) as core:
        d_off = torch.full((3,), 9, dtype=torch.int32, device=DEV)
        core.diff_stream_batch(None, 2, d_off, None, None, 0)
        core.synchronize()
        assert d_off.cpu().tolist() == [0, 0, 0]
    with CUDACore(16, 16, max_batch=2) as core:
        d_off = torch.full((1,), 9, dtype=torch.int32, device=DEV)
        core.diff_stream_batch(None, 0, d_off, None, None, 0)
        core.synchronize()
        assert d_off.cpu().tolist() == [0]
        with pytest.raises(lib.Mi355Error):
            core.diff_stream_batch(d_off, 3, d_off, None, None, 0)   # nframes > max_batch


def test_int_diff_benchmark_identity(po):
    """tests/algorithms_benchmarks.cu: GPU diff checked by the reference's own checkDifference."""
    h, w = 1920, 1080
    n = h * w * 3
    L = po.lib()
    a = np.empty(n, np.int32); b = np.empty(n, np.int32)
    L.ora_generate_image(a, h, w, 1)
    L.ora_generate_image(b, h, w, 2)
    with CUDACore(8, 8) as core:
        d_a, d_b = to_dev(a), to_dev(b)
        d_o = torch.empty_like(d_a)
        core.int_diff(d_a, d_b, d_o, n)
        core.synchronize()
        out = d_o.cpu().numpy()
        assert L.ora_check_difference(a, b, out, h, w) == 0
        assert np.array_equal(out, a - b)
        core.int_diff(d_a.data_ptr() + 4, d_b.data_ptr() + 4, d_o.data_ptr() + 4, 1001)  # unaligned, ragged
        core.synchronize()
        assert np.array_equal(d_o.cpu().numpy()[1:1002], (a - b)[1:1002])


# ---- BASELINE.json sizes ------------------------------------------------------------------------------

def test_1080p_stream_vs_oracle(po):
    """config 2 at full size on a short batch the oracle finishes in seconds."""
    W, H, T = 1920, 1080, 6
    base, frames = synth.webcam_stream(T, W, H, device=DEV)
    with CUDACore(W, H, max_batch=T) as core:
        core.set_state(base.cpu().numpy())
        off, xs, df, _ = run_stream(core, frames)
        eo, exs, edf, est = po.diff_stream(frames.cpu().numpy(), base.cpu().numpy())
        assert np.array_equal(off, eo) and np.array_equal(xs, exs) and np.array_equal(df, edf)
        assert np.array_equal(core.get_state(), est)
        p = np.diff(off.astype(np.int64))[1:] / (3 * W * H)
        assert (0.01 < p).all() and (p < 0.06).all()


def test_1080p_long_batch_properties():
    """Full-size, 64-frame batch: size-independent properties instead of the (slow) oracle --
    the client reconstruction (client/opencv.cpp:64-66) applied to the packed output rebuilds the
    server state exactly, indices ascend strictly inside every frame, and the un-sent residual is
    within the threshold of the true frame."""
    W, H, T = 1920, 1080, 64
    n = 3 * W * H
    base, frames = synth.webcam_stream(T, W, H, device=DEV)
    with CUDACore(W, H, max_batch=T) as core:
        core.set_state(base.cpu().numpy())
        cap = T * n // 8
        d_off = torch.zeros(T + 1, dtype=torch.int32, device=DEV)
        d_xs = torch.empty(cap, dtype=torch.int32, device=DEV)
        d_df = torch.empty(cap, dtype=torch.uint8, device=DEV)
        core.diff_stream_batch(frames, T, d_off, d_xs, d_df, cap)
        core.synchronize()
        off = d_off.cpu().numpy().view(np.uint32).astype(np.int64)
        assert off[0] == 0 and (np.diff(off) > 0).all() and off[-1] <= cap
        client = base.clone()
        for t in range(T):
            xs = d_xs[off[t]:off[t + 1]].long()
            assert bool((xs[1:] > xs[:-1]).all()) and int(xs[0]) >= 0 and int(xs[-1]) < n
            client[xs] += d_df[off[t]:off[t + 1]]          # uint8 wrap == client/opencv.cpp:65
            resid = (client.to(torch.int16) - frames[t].to(torch.int16)).abs().max()
            assert int(resid) <= 20
        state = torch.from_numpy(core.get_state()).to(DEV)
        assert bool((client == state).all())


def test_4k_pair_vs_oracle(po):
    """config 5 frame size (3840x2160), one S0-style dense pair and one sparse pair."""
    W, H = 3840, 2160
    n = 3 * W * H
    cur = torch.stack([synth.refrand_frame(n, 1, device=DEV), synth.webcam_frame(1, W, H, device=DEV)])
    prev = torch.stack([synth.refrand_frame(n, 2, device=DEV), synth.webcam_frame(0, W, H, device=DEV)])
    with CUDACore(W, H, max_batch=2) as core:
        off, xs, df, _ = run_stream(core, cur, pair_prev=prev)
        eo, exs, edf = oracle_pairs(po, cur.cpu().numpy(), prev.cpu().numpy())
        assert np.array_equal(off, eo) and np.array_equal(xs, exs) and np.array_equal(df, edf)


def test_4k_stream_vs_oracle(po):
    """config 5 frame size as a stateful stream (3840x2160, 4 frames)."""
    W, H, T = 3840, 2160, 4
    base, frames = synth.webcam_stream(T, W, H, seed=31, device=DEV)
    with CUDACore(W, H, max_batch=T) as core:
        core.set_state(base.cpu().numpy())
        off, xs, df, _ = run_stream(core, frames, capacity=T * 3 * W * H // 4)
        eo, exs, edf, est = po.diff_stream(frames.cpu().numpy(), base.cpu().numpy())
        assert np.array_equal(off, eo) and np.array_equal(xs, exs) and np.array_equal(df, edf)
        assert np.array_equal(core.get_state(), est)


def test_long_batch_crosses_expand_table_blocks(po):
    """More than 256 frames in one batch: k_expand's per-tile table is rebuilt per 256-frame block and
    the frame cut by a chunk boundary carries across blocks."""
    w, h, T = 64, 36, 600
    base, frames = synth.webcam_stream(T, w, h, seed=13)
    frames = frames.copy()
    frames[255:258] = frames[0]      # quiet frames around the first block boundary
    with CUDACore(w, h, max_batch=T) as core:
        check_stream(po, core, base, frames)


def _density_frames(rng, base, T, plan):
    """Frames whose flagged-byte density varies by region: plan = [(first_byte, last_byte, p)], every
    selected byte moves by more than the threshold, the rest by at most 3."""
    n = base.size
    frames = np.empty((T, n), np.uint8)
    for t in range(T):
        f = (base.astype(np.int16) + rng.integers(-3, 4, n)).clip(0, 255)
        for a, b, p in plan:
            hit = np.flatnonzero(rng.random(b - a) < p) + a
            f[hit] = (base[hit].astype(np.int16) + np.where(base[hit] < 128, 60, -60) + 5 * (t % 3))
        frames[t] = f.astype(np.uint8)
    return frames


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_mixed_density_regions_cover_every_expander_path(po, seed):
    """k_expand stages a workgroup's entries in LDS when they fit (<= 3072 per 64 tiles), stores directly
    otherwise, walks light records lane by lane and emits records with more than 4 flagged bytes
    cooperatively unless a wave holds more than 12 of them: regions of 64 KiB with densities from one byte
    per few records up to every byte put each combination, and the boundaries between them, in one batch."""
    rng = np.random.default_rng(100 + seed)
    w, h, T = 1024, 160, 6                     # 491 520 bytes = 480 tiles = 7.5 workgroups of 64 tiles
    n = 3 * w * h
    base = rng.integers(0, 256, n).astype(np.uint8)
    grp = 64 * 1024
    dens = [0.0005, 0.004, 0.02, 0.045, 0.05, 0.3, 1.0, 0.01]     # 0.045..0.05: around 3072 entries / group
    plan = [(g * grp, min((g + 1) * grp, n), dens[(g + seed) % len(dens)]) for g in range((n + grp - 1) // grp)]
    # a few fully flagged 16-byte lanes (heavy records) sprinkled into the light regions: 1..20 per wave
    for g in range(len(plan)):
        a = g * grp
        for k in range(1 + 3 * g):
            s = a + 16 * int(rng.integers(0, grp // 16 - 1))
            if s + 16 <= n:
                plan.append((s, s + 16, 1.0))
    frames = _density_frames(rng, base, T, plan)
    off, xs, df, st = po.diff_stream(frames, base)
    with CUDACore(w, h, sample_mat_data=base, max_batch=T) as core:
        g_off, g_xs, g_df, _ = run_stream(core, frames)
        assert np.array_equal(g_off, off)
        assert np.array_equal(g_xs, xs)
        assert np.array_equal(g_df, df)
        assert np.array_equal(core.get_state(), st)
        # the same batch as the sender's byte stream (k_expand<WIRE>)
        core.set_state(base)
        want = po.wire_pack(off, xs, df)
        d_off = torch.zeros(T + 1, dtype=torch.int32, device=DEV)
        d_wire = torch.full((want.size + 32,), 0x5C, dtype=torch.uint8, device=DEV)
        core.diff_stream_wire_batch(to_dev(frames), T, d_off, d_wire, want.size)
        core.synchronize()
        wire = d_wire.cpu().numpy()
        assert np.array_equal(wire[:want.size], want) and (wire[want.size:] == 0x5C).all()


@pytest.mark.parametrize("per_group", [3071, 3072, 3073, 4097])
def test_staging_boundary_exact_entry_counts(po, per_group):
    """Exactly 3071 / 3072 / 3073 / 4097 flagged bytes inside one 64-tile workgroup (the LDS staging holds
    3072 entries), next to an empty and a full group."""
    rng = np.random.default_rng(per_group)
    w, h = 1024, 64                            # 196 608 bytes = 3 groups of 64 tiles
    n = 3 * w * h
    base = rng.integers(0, 200, n).astype(np.uint8)
    frame = base.copy()
    hit = np.sort(rng.choice(64 * 1024, per_group, replace=False)) + 64 * 1024      # group 1
    frame[hit] += 50
    frame[2 * 64 * 1024:] += 50                                                       # group 2: every byte
    c, xs, df, st = po.diff_pack(frame, base)
    assert c == per_group + 64 * 1024
    with CUDACore(w, h, sample_mat_data=base, max_batch=1) as core:
        g_off, g_xs, g_df, _ = run_stream(core, frame[None, :])
        assert g_off.tolist() == [0, c]
        assert np.array_equal(g_xs, xs) and np.array_equal(g_df, df)
        assert np.array_equal(core.get_state(), st)


@pytest.mark.parametrize("heavy", [0, 1, 11, 12, 13, 14, 40, 64])
def test_heavy_record_count_per_wave(po, heavy):
    """`heavy` fully flagged lanes among the first 64 records of a workgroup, isolated bytes elsewhere: the
    cooperative emission takes over up to 12 heavy records per wave and hands back above."""
    rng = np.random.default_rng(heavy)
    w, h = 1024, 22                            # 67 584 bytes: one full group of 64 tiles + 2 tiles
    n = 3 * w * h
    base = rng.integers(0, 200, n).astype(np.uint8)
    frame = base.copy()
    lanes = rng.permutation(64)                # records 0..63 of the group = the 64 lanes of tile 0, all candidates
    for i, l in enumerate(lanes):
        if i < heavy:
            frame[16 * l:16 * l + 16] += 40
        else:
            frame[16 * l + int(rng.integers(0, 16))] += 40
    frame[5000::97] += 30
    c, xs, df, st = po.diff_pack(frame, base)
    with CUDACore(w, h, sample_mat_data=base, max_batch=1) as core:
        g_off, g_xs, g_df, _ = run_stream(core, frame[None, :])
        assert g_off.tolist() == [0, c]
        assert np.array_equal(g_xs, xs) and np.array_equal(g_df, df)
        assert np.array_equal(core.get_state(), st)


# ---- pipelined batches: expansion of batch k beside the pack kernel of batch k + 1 (own stream) ---------------------------
def _check_batches_against_oracle(outs, eo, exs, edf, T):
    per_frame = np.diff(eo.astype(np.int64))
    at = 0
    for k, (o, x, d) in enumerate(outs):
        cnt = per_frame[k * T:(k + 1) * T]
        off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint32)
        tot = int(off[-1])
        assert np.array_equal(o.cpu().numpy().view(np.uint32), off), k
        assert np.array_equal(x[:tot].cpu().numpy(), exs[at:at + tot]), k
        assert np.array_equal(d[:tot].cpu().numpy(), edf[at:at + tot]), k
        at += tot


@pytest.mark.parametrize("own_stream", [True, False])
def test_scan_epoch_wrap_clears_the_totals(po, own_stream):
    """The index kernel's frame totals travel as {total: 31 bits, launch tag: 33 bits}; when the tag wraps (2^33 launches) the
    host clears every total behind a synchronisation before a tag is used again (core.hip, next_scan_epoch).  Forced here:
    the tag is put 3 launches before its wrap (MI355_OPT_SCAN_EPOCH_LEFT), nine batches run across it back to back -- on
    the core's own stream (pipelined, two sets of totals) and on a caller's -- and every batch must equal the oracle."""
    from cudavideostream_amd import lib as L
    w, h, T, K = 320, 180, 4, 9
    n = 3 * w * h
    base, frames = synth.webcam_stream(T * K, w, h, seed=57)
    eo, exs, edf, est = po.diff_stream(frames, base)
    d_fr = to_dev(frames)
    outs = [(torch.zeros(T + 1, dtype=torch.int32, device=DEV), torch.full((T * n,), -7, dtype=torch.int32, device=DEV),
             torch.zeros(T * n, dtype=torch.uint8, device=DEV)) for _ in range(K)]
    with CUDACore(w, h, max_batch=T, sample_mat_data=base) as core:
        if not own_stream:
            core.use_torch_stream()
        torch.cuda.synchronize()
        RawCore.diff_stream_batch(core, d_fr[:T], T, *outs[0], T * n)     # the pipelined mode is set up, both sets of totals exist
        core.set_option(L.OPT_SCAN_EPOCH_LEFT, 3)
        assert core.get_option(L.OPT_SCAN_EPOCH_LEFT) == 3
        for k in range(1, K):
            if k == 2:   # the tag set BACK onto values the slots already carry (the option clears the totals with the jump)
                core.set_option(L.OPT_SCAN_EPOCH_LEFT, 3)
            RawCore.diff_stream_batch(core, d_fr[k * T:(k + 1) * T], T, *outs[k], T * n)
        core.synchronize()
        torch.cuda.synchronize()
        assert core.get_option(L.OPT_SCAN_EPOCH_LEFT) == (1 << 30)      # wrapped: the tag restarted at 1
        assert np.array_equal(core.get_state(), est)
    _check_batches_against_oracle(outs, eo, exs, edf, T)


def test_streams_in_their_own_priority_class_give_the_same_stream(po):
    """MI355_FLAG_OWN_QUEUES: the core's three streams in the least stream-priority class (hardware queues of their own in a
    process that also holds a framework's stream pools); pipelined batches back to back, every batch against the oracle."""
    w, h, T, K = 320, 180, 6, 5
    n = 3 * w * h
    base, frames = synth.webcam_stream(T * K, w, h, seed=91)
    eo, exs, edf, est = po.diff_stream(frames, base)
    d_fr = to_dev(frames)
    outs = [(torch.zeros(T + 1, dtype=torch.int32, device=DEV), torch.full((T * n,), -7, dtype=torch.int32, device=DEV),
             torch.zeros(T * n, dtype=torch.uint8, device=DEV)) for _ in range(K)]
    with CUDACore(w, h, max_batch=T, sample_mat_data=base, flags=lib.FLAG_OWN_QUEUES) as core:
        torch.cuda.synchronize()
        for k in range(K):
            RawCore.diff_stream_batch(core, d_fr[k * T:(k + 1) * T], T, *outs[k], T * n)
        core.synchronize()
        assert np.array_equal(core.get_state(), est)
    _check_batches_against_oracle(outs, eo, exs, edf, T)


@pytest.mark.parametrize("w,h,T", [(320, 180, 6), (1920, 1080, 8)])
def test_alloc_outputs_gives_a_usable_pair(po, w, h, T):
    """mi355_alloc_outputs: the index and the value array of the batch entry points as a pair placed for the dense expansion.
    Small frames take plain allocations (1 draw); at 1080p x 8 frames the library probes candidate value arrays with its own
    kernels on noise frames (the core's state and later results must not notice).  A dense and a sparse pair batch into the
    pair, read back through mi355_download, against the oracle."""
    import ctypes as C
    n = 3 * w * h
    cur = np.stack([synth.refrand_frame(n, 700 + t) for t in range(T)])
    prev = np.stack([synth.refrand_frame(n, 800 + t) for t in range(T)])
    base, frames = synth.webcam_stream(T, w, h, seed=5)
    with CUDACore(w, h, max_batch=T, sample_mat_data=base) as core:
        cap = T * n
        d_xs, d_df, draws = core.alloc_outputs(cap)
        assert d_xs and d_df and 1 <= draws <= 32
        assert np.array_equal(core.get_state(), base)           # the probe batches are stateless pairs
        d_off = torch.zeros(T + 1, dtype=torch.int32, device=DEV)

        def fetch(total):
            xs, df = np.empty(total, np.int32), np.empty(total, np.uint8)
            L = lib.load()
            lib.check(L.mi355_download(core._h, xs.ctypes.data_as(C.c_void_p), C.c_void_p(d_xs), 4 * total))
            lib.check(L.mi355_download(core._h, df.ctypes.data_as(C.c_void_p), C.c_void_p(d_df), total))
            return xs, df

        core.diff_pairs_batch(to_dev(cur), to_dev(prev), T, d_off, d_xs, d_df, cap)
        core.synchronize()
        eo, exs, edf = oracle_pairs(po, cur, prev)
        assert np.array_equal(d_off.cpu().numpy().view(np.uint32), eo)
        xs, df = fetch(int(eo[-1]))
        assert np.array_equal(xs, exs) and np.array_equal(df, edf)
        core.diff_stream_batch(to_dev(frames), T, d_off, d_xs, d_df, cap)
        core.synchronize()
        eo, exs, edf, est = po.diff_stream(frames, base)
        assert np.array_equal(d_off.cpu().numpy().view(np.uint32), eo)
        xs, df = fetch(int(eo[-1]))
        assert np.array_equal(xs, exs) and np.array_equal(df, edf) and np.array_equal(core.get_state(), est)
        core.dev_free(d_xs)
        core.dev_free(d_df)


def test_core_created_while_the_null_stream_is_busy(po):
    """The core's own buffers are cleared through the core's own stream and waited for (round 6: the launch tags' wrap path
    cleared with plain hipMemset calls and the chain soak hung there).  This is the creation case: the null stream -- torch's
    default stream -- is busy for about a second while a core is created, its state set and two pipelined batches run; the
    state and both sets of totals must be what the batches expect whenever the clears ran."""
    w, h, T = 320, 180, 4
    n = 3 * w * h
    base, frames = synth.webcam_stream(2 * T, w, h, seed=64)
    eo, exs, edf, est = po.diff_stream(frames, base)
    d_fr = to_dev(frames)
    outs = [(torch.zeros(T + 1, dtype=torch.int32, device=DEV), torch.full((T * n,), -7, dtype=torch.int32, device=DEV),
             torch.zeros(T * n, dtype=torch.uint8, device=DEV)) for _ in range(2)]
    torch.cuda.synchronize()
    torch.cuda._sleep(2_000_000_000)          # ~1 s of the null stream (cycles of the device clock)
    with RawCore(w, h, max_batch=T, sample_mat_data=base) as core:
        RawCore.diff_stream_batch(core, d_fr[:T], T, *outs[0], T * n)        # sets the pipelined mode up (second log set)
        core.synchronize()
        torch.cuda.synchronize()              # the null stream's work is over: anything that waited behind it has run now
        RawCore.diff_stream_batch(core, d_fr[T:], T, *outs[1], T * n)
        core.synchronize()
        assert np.array_equal(core.get_state(), est)
    _check_batches_against_oracle(outs, eo, exs, edf, T)


def test_prepare_leaves_nothing_to_allocate(po):
    """mi355_prepare(MI355_PREPARE_ALL): the second set of logs, the side streams and events of pipelined batches, the gray
    bytes of the fused binarize chain, the cleared red map's slice bounds and the K x K taps are made NOW; the entry points
    that would have made them on first use then leave mi355_workspace_bytes where it is (and give the oracle's results)."""
    from cudavideostream_amd import lib as L
    w, h, T = 320, 180, 6
    n = 3 * w * h
    base, frames = synth.webcam_stream(2 * T, w, h, seed=12)
    eo, exs, edf, est = po.diff_stream(frames, base)
    d_fr = to_dev(frames)
    outs = [(torch.zeros(T + 1, dtype=torch.int32, device=DEV), torch.full((T * n,), -7, dtype=torch.int32, device=DEV),
             torch.zeros(T * n, dtype=torch.uint8, device=DEV)) for _ in range(2)]
    vis = torch.empty((T, n), dtype=torch.uint8, device=DEV)
    with CUDACore(w, h, max_batch=T, sample_mat_data=base) as core:
        ws0 = core.workspace_bytes
        core.prepare()
        ws1 = core.workspace_bytes
        assert ws1 > ws0 + T * n          # the second record log alone is max_batch frames
        core.prepare(L.PREPARE_BATCHES | L.PREPARE_GRAY_CHAIN)   # idempotent
        assert core.workspace_bytes == ws1
        torch.cuda.synchronize()
        for k in range(2):                # own stream: pipelined
            RawCore.diff_stream_batch(core, d_fr[k * T:(k + 1) * T], T, *outs[k], T * n)
        core.filter_batch(L.OP_GRAY_WEIGHTED_BINARIZE, d_fr[:T], vis, T)
        core.red_stream_batch(outs[1][0], outs[1][1], T, vis)
        k5 = np.full(25, 1 / 25, np.float32)
        one = torch.empty(n, dtype=torch.uint8, device=DEV)
        core.conv_kxk(d_fr[0], one, k5)
        core.synchronize()
        assert core.workspace_bytes == ws1
        assert np.array_equal(core.get_state(), est)
    _check_batches_against_oracle(outs, eo, exs, edf, T)
    with pytest.raises(Exception):
        with CUDACore(w, h) as core:
            core.prepare(1 << 9)          # an unknown bit is refused


def test_back_to_back_batches_without_synchronisation(po):
    """Seven batches of one stream queued back to back on the core's own stream (the index and the expansion of a
    batch then run on the side stream beside the next batch's pack kernel, two sets of logs in turn), every batch
    into its own output buffers; one synchronisation at the end; every batch against the oracle, state carried."""
    w, h, T, K = 320, 180, 6, 7
    n = 3 * w * h
    base, frames = synth.webcam_stream(T * K, w, h, seed=33)
    eo, exs, edf, est = po.diff_stream(frames, base)
    d_fr = to_dev(frames)
    outs = [(torch.zeros(T + 1, dtype=torch.int32, device=DEV), torch.full((T * n,), -7, dtype=torch.int32, device=DEV),
             torch.zeros(T * n, dtype=torch.uint8, device=DEV)) for _ in range(K)]
    with CUDACore(w, h, max_batch=T, sample_mat_data=base) as core:
        torch.cuda.synchronize()
        for k in range(K):   # the product class's method, not the harness's wrapper: no host synchronisation between the calls
            RawCore.diff_stream_batch(core, d_fr[k * T:(k + 1) * T], T, *outs[k], T * n)
        core.synchronize()
        assert np.array_equal(core.get_state(), est)
    # oracle: per-frame entries of the whole sequence, cut into the batches
    per_frame = np.diff(eo.astype(np.int64))
    at = 0
    for k in range(K):
        cnt = per_frame[k * T:(k + 1) * T]
        off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint32)
        tot = int(off[-1])
        assert np.array_equal(outs[k][0].cpu().numpy().view(np.uint32), off), k
        assert np.array_equal(outs[k][1][:tot].cpu().numpy(), exs[at:at + tot]), k
        assert np.array_equal(outs[k][2][:tot].cpu().numpy(), edf[at:at + tot]), k
        at += tot


def test_back_to_back_pair_batches_without_synchronisation(po):
    """Pair mode on the core's own stream: five batches queued back to back (two pack launches per batch on two streams, the
    expansion beside the next batch's pack kernels), alternately pairs of consecutive frames (plain loads) and pairs that
    share no frame (non-temporal loads); one synchronisation at the end; every batch against the oracle, state untouched."""
    w, h, T, K = 320, 180, 5, 5
    n = 3 * w * h
    _, frames = synth.webcam_stream(2 * T * K, w, h, seed=91)
    frames = np.ascontiguousarray(frames)
    d_fr = to_dev(frames)
    marker = synth.refrand_frame(n, 5)
    outs = [(torch.zeros(T + 1, dtype=torch.int32, device=DEV), torch.full((T * n,), -7, dtype=torch.int32, device=DEV),
             torch.zeros(T * n, dtype=torch.uint8, device=DEV)) for _ in range(K)]
    want = []
    with CUDACore(w, h, max_batch=T, sample_mat_data=marker) as core:
        torch.cuda.synchronize()
        for k in range(K):
            blk = d_fr[2 * T * k:2 * T * (k + 1)]
            host = frames[2 * T * k:2 * T * (k + 1)]
            if k % 2 == 0:   # (0,1), (1,2), ...: cur of a pair is prev of the next
                RawCore.diff_pairs_batch(core, blk[1:T + 1], blk[0:T], T, *outs[k], T * n)
                want.append(oracle_pairs(po, host[1:T + 1], host[0:T]))
            else:            # (0,1), (2,3), ...: a stride of two frames, no frame twice
                RawCore.diff_pairs_batch(core, blk[1::2], blk[0::2], T, *outs[k], T * n, stride=2 * n)
                want.append(oracle_pairs(po, host[1::2], host[0::2]))
        core.synchronize()
        assert np.array_equal(core.get_state(), marker)
    for k in range(K):
        eo, exs, edf = want[k]
        tot = int(eo[-1])
        assert np.array_equal(outs[k][0].cpu().numpy().view(np.uint32), eo), k
        assert np.array_equal(outs[k][1][:tot].cpu().numpy(), exs), k
        assert np.array_equal(outs[k][2][:tot].cpu().numpy(), edf), k


def test_filter_then_diff_on_own_stream_orders_every_part(po):
    """Config 4's chain on the core's own stream with ONE scratch buffer and no synchronisation in between: the noise filter
    writes `filt`, the batch's pack kernels -- two launches on two streams of the core -- read it, the next round's filter
    rewrites it.  Every pack launch has to come behind the filter that made its input (the parts on other streams wait for
    an event of the core's stream) and the next filter behind every part of the batch before.  Against the oracle, state
    carried through the filtered frames."""
    w, h, T, K = 640, 360, 6, 4
    n = 3 * w * h
    k9 = po.gaussian_kernel(3, 1.5)
    base, frames = synth.webcam_stream(T * K, w, h, seed=83)
    frames = np.ascontiguousarray(frames)
    filtered = np.stack([po.conv3x3(f, w, h, k9) for f in frames])
    eo, exs, edf, est = po.diff_stream(filtered, base)
    d_fr = to_dev(frames)
    filt = torch.empty((T, n), dtype=torch.uint8, device=DEV)
    outs = [(torch.zeros(T + 1, dtype=torch.int32, device=DEV), torch.full((T * n,), -7, dtype=torch.int32, device=DEV),
             torch.zeros(T * n, dtype=torch.uint8, device=DEV)) for _ in range(K)]
    # (MI355_OPT_CHAIN_HINT 0: by default a batch that follows a frame filter is not overlapped at all -- this test is about
    # the overlapped batch's ordering)
    with CUDACore(w, h, k=k9, max_batch=T, sample_mat_data=base) as core:
        core.set_option(lib.OPT_CHAIN_HINT, 0)
        assert core.get_option(lib.OPT_CHAIN_HINT) == 0
        torch.cuda.synchronize()
        for k in range(K):
            RawCore.filter_batch(core, lib.OP_CONV3X3, d_fr[k * T:(k + 1) * T], filt, T)
            RawCore.diff_stream_batch(core, filt, T, *outs[k], T * n)
        core.synchronize()
        assert np.array_equal(core.get_state(), est)
    per_frame = np.diff(eo.astype(np.int64))
    at = 0
    for k in range(K):
        cnt = per_frame[k * T:(k + 1) * T]
        off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint32)
        tot = int(off[-1])
        assert np.array_equal(outs[k][0].cpu().numpy().view(np.uint32), off), k
        assert np.array_equal(outs[k][1][:tot].cpu().numpy(), exs[at:at + tot]), k
        assert np.array_equal(outs[k][2][:tot].cpu().numpy(), edf[at:at + tot]), k
        at += tot


def test_chain_soak_short(po):
    """A short run of tests/soak_chain.py: random sequences of filters, stream and pair batches, red maps and stream switches
    on one core without host synchronisation inside a round, scratch buffers reused throughout; every output and the state
    against the oracle."""
    import soak_chain
    assert soak_chain.run(12, 5, verbose=False)
    assert soak_chain.run(4, 6, w=640, h=360, T=4, verbose=False)


def test_pipelined_1080p_batches_equal_the_sequential_path(po):
    """Full-size overlap: five batches of 65 (odd) 1080p frames queued back to back on the core's own stream -- the
    expansion of batch k runs beside the pack kernel of batch k + 1, two sets of logs in turn, the pack kernel on its
    pipelined grid --, the last two in the wire form; against the same sequence on a caller's stream (never pipelined),
    byte for byte: offsets, indices, differences, wire bytes and the final state; batch 0 also against the oracle."""
    w, h, T, K = 1920, 1080, 65, 5
    n = 3 * w * h
    base, frames = synth.webcam_stream(T * K, w, h, seed=77, device=DEV)
    cap = T * n // 8
    wire_cap = 4 * T + 5 * cap

    def run(pipelined):
        outs = [(torch.zeros(T + 1, dtype=torch.int32, device=DEV), torch.full((cap,), -7, dtype=torch.int32, device=DEV),
                 torch.zeros(cap, dtype=torch.uint8, device=DEV)) for _ in range(K - 2)]
        wires = [(torch.zeros(T + 1, dtype=torch.int32, device=DEV), torch.zeros(wire_cap, dtype=torch.uint8, device=DEV)) for _ in range(2)]
        with CUDACore(w, h, max_batch=T, sample_mat_data=base.cpu().numpy()) as core:
            if not pipelined:
                core.use_torch_stream()
            torch.cuda.synchronize()
            for k in range(K - 2):
                RawCore.diff_stream_batch(core, frames[k * T:(k + 1) * T], T, *outs[k], cap)
            for j in range(2):
                k = K - 2 + j
                RawCore.diff_stream_wire_batch(core, frames[k * T:(k + 1) * T], T, wires[j][0], wires[j][1], wire_cap)
            core.synchronize()
            torch.cuda.synchronize()
            state = core.get_state()
        res = []
        for o, x, d in outs:
            off = o.cpu().numpy().view(np.uint32)
            assert int(off[-1]) <= cap
            res.append((off, x[:int(off[-1])].cpu().numpy(), d[:int(off[-1])].cpu().numpy()))
        for o, wbuf in wires:
            off = o.cpu().numpy().view(np.uint32)
            res.append((off, wbuf[:4 * T + 5 * int(off[-1])].cpu().numpy()))
        return res, state

    got, st_p = run(True)
    want, st_s = run(False)
    assert np.array_equal(st_p, st_s)
    for k in range(K):
        for a, b in zip(got[k], want[k]):
            assert np.array_equal(a, b), k
    eo, exs, edf, _ = po.diff_stream(frames[:T].cpu().numpy(), base.cpu().numpy())
    assert np.array_equal(got[0][0], eo) and np.array_equal(got[0][1], exs) and np.array_equal(got[0][2], edf)


def test_pipelined_batches_then_other_entry_points(po):
    """What follows a pipelined batch on the same core -- here the red motion map built from the batch's packed
    stream, queued at once -- sees the batch complete (every entry point joins the side stream first); the client's
    reconstruction on a second core after one synchronisation of the server."""
    w, h, T = 256, 144, 5
    n = 3 * w * h
    base, frames = synth.webcam_stream(3 * T, w, h, seed=34)
    eo, exs, edf, est = po.diff_stream(frames, base)
    per_frame = np.diff(eo.astype(np.int64))
    d_fr = to_dev(frames)
    with CUDACore(w, h, max_batch=T, sample_mat_data=base) as core, CUDACore(w, h, max_batch=T, sample_mat_data=base) as client:
        torch.cuda.synchronize()
        stream_batch = lambda *a, **kw: RawCore.diff_stream_batch(core, *a, **kw)   # noqa: E731  (no host synchronisation in between)
        red_batch = lambda *a, **kw: RawCore.red_stream_batch(core, *a, **kw)      # noqa: E731
        outs, maps = [], []
        for k in range(3):
            d_off = torch.zeros(T + 1, dtype=torch.int32, device=DEV)
            d_xs = torch.zeros(T * n, dtype=torch.int32, device=DEV)
            d_df = torch.zeros(T * n, dtype=torch.uint8, device=DEV)
            d_map = torch.full((T, n), 0x5a, dtype=torch.uint8, device=DEV)
            torch.cuda.synchronize()
            stream_batch(d_fr[k * T:(k + 1) * T], T, d_off, d_xs, d_df, T * n)
            red_batch(d_off, d_xs, T, d_map, True)             # reads the batch's output at once
            outs.append((d_off, d_xs, d_df)); maps.append(d_map)
        core.synchronize()
        at = 0
        for k in range(3):
            got = maps[k].cpu().numpy()
            for t in range(T):
                cnt = int(per_frame[k * T + t])
                want = np.zeros(n, np.uint8)
                x = exs[at:at + cnt].astype(np.int64)
                want[x + (2 - x % 3)] = 255                     # kernels.cu:273-281 on a cleared frame
                assert np.array_equal(got[t], want), (k, t)
                at += cnt
            client.apply_batch(*outs[k], T)
        client.synchronize()
        assert np.array_equal(client.get_state(), est)          # the client's frame == the server's state
        assert np.array_equal(core.get_state(), est)


def test_options_change_the_schedule_never_the_result(po):
    """mi355_set_option (include/mi355diff.h, "Options"): every combination of pipelining, split share, pack grid, dense
    threshold and chain hint gives the oracle's stream for batches queued back to back on the core's own stream; values
    outside an option's range, unknown options and non-zero cfg.flags are refused."""
    w, h, T, K = 640, 360, 5, 3
    n = 3 * w * h
    base, frames = synth.webcam_stream(T * K, w, h, seed=91)
    frames = np.ascontiguousarray(frames)
    eo, exs, edf, est = po.diff_stream(frames, base)
    d_fr = to_dev(frames)
    combos = [{}, {lib.OPT_PIPELINE: 0}, {lib.OPT_SPLIT_PCT: 0}, {lib.OPT_SPLIT_PCT: 30}, {lib.OPT_SPLIT_PCT: 95},
              {lib.OPT_PACK_BLOCKS: 0}, {lib.OPT_PACK_BLOCKS: 7}, {lib.OPT_PACK_BLOCKS: 4096, lib.OPT_SPLIT_PCT: 70},
              {lib.OPT_DENSE_PCT: 0}, {lib.OPT_DENSE_PCT: 1}, {lib.OPT_CHAIN_HINT: 0, lib.OPT_DENSE_PCT: 100}]
    per_frame = np.diff(eo.astype(np.int64))
    for opts in combos:
        with CUDACore(w, h, max_batch=T, sample_mat_data=base) as core:
            for k, v in opts.items():
                core.set_option(k, v)
                assert core.get_option(k) == v
            outs = [(torch.zeros(T + 1, dtype=torch.int32, device=DEV), torch.full((T * n,), -7, dtype=torch.int32, device=DEV),
                     torch.zeros(T * n, dtype=torch.uint8, device=DEV)) for _ in range(K)]
            torch.cuda.synchronize()
            for k in range(K):   # no synchronisation between the batches
                RawCore.diff_stream_batch(core, d_fr[k * T:(k + 1) * T], T, *outs[k], T * n)
            core.synchronize()
            assert np.array_equal(core.get_state(), est), opts
            at = 0
            for k in range(K):
                cnt = per_frame[k * T:(k + 1) * T]
                off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.uint32)
                tot = int(off[-1])
                assert np.array_equal(outs[k][0].cpu().numpy().view(np.uint32), off), (opts, k)
                assert np.array_equal(outs[k][1][:tot].cpu().numpy(), exs[at:at + tot]), (opts, k)
                assert np.array_equal(outs[k][2][:tot].cpu().numpy(), edf[at:at + tot]), (opts, k)
                at += tot
    with CUDACore(8, 8) as core:
        assert core.get_option(lib.OPT_PIPELINE) == 1 and core.get_option(lib.OPT_SPLIT_PCT) == 50
        assert core.get_option(lib.OPT_DENSE_PCT) == 40 and core.get_option(lib.OPT_CHAIN_HINT) == 1
        assert core.get_option(lib.OPT_PACK_BLOCKS) == -1
        for k, v in ((lib.OPT_PIPELINE, 2), (lib.OPT_SPLIT_PCT, 3), (lib.OPT_SPLIT_PCT, 96), (lib.OPT_DENSE_PCT, 101),
                     (lib.OPT_CHAIN_HINT, -1), (lib.OPT_PACK_BLOCKS, -2), (99, 0), (0, 1)):
            with pytest.raises(lib.Mi355Error):
                core.set_option(k, v)
    import ctypes as C
    h_ = C.c_void_p()
    cfg = lib.Config(8, 8, 20, 1, -1, 0, 0, 2)   # an unknown flag bit (the experiment flags of rounds 2-4 are gone; bit 0 is MI355_FLAG_OWN_QUEUES)
    assert lib.load().mi355_create(C.byref(cfg), C.byref(h_)) == lib.ERR_INVALID
