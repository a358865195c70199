"""-m gpu parity tests of the operators either side of the packed stream,
all through the C-ABI: the sender's wire bytes (server/src/threads.cpp:227-229), the client's
reconstruction (client/opencv.cpp:50-66) and the merge of row-band streams."""
import numpy as np
import pytest
import torch

from cudavideostream_amd import synth
from oracle import pyoracle as po
from gpu_util import DEV, CUDACore, run_stream, to_dev

pytestmark = pytest.mark.gpu

GUARD = 0x5C


def run_wire(core, frames, capacity=None):
    d_frames = frames if torch.is_tensor(frames) else to_dev(frames)
    T = d_frames.shape[0]
    cap = 4 * T + 5 * T * core.total if capacity is None else capacity
    d_off = torch.full((T + 1,), -1, dtype=torch.int32, device=DEV)
    d_wire = torch.full((cap + 64,), GUARD, dtype=torch.uint8, device=DEV)
    core.diff_stream_wire_batch(d_frames, T, d_off, d_wire, cap)
    core.synchronize()
    return d_off.cpu().numpy().view(np.uint32), d_wire.cpu().numpy(), d_wire


@pytest.mark.parametrize("w,h,T", [(64, 48, 5), (33, 7, 9), (1, 1, 3), (640, 360, 4), (211, 3, 17)])
def test_wire_matches_sender_bytes(w, h, T):
    base, frames = synth.webcam_stream(T, w, h, seed=5)
    frames = frames.copy()
    frames[T // 2] = frames[T // 2 - 1]            # a frame with nothing to send: header only
    off, xs, df, st = po.diff_stream(frames, base)
    want = po.wire_pack(off, xs, df)
    with CUDACore(w, h, sample_mat_data=base, max_batch=T) as core:
        g_off, wire, _ = run_wire(core, frames)
        assert np.array_equal(g_off, off)
        assert core.wire_bytes(T, int(off[-1])) == want.size
        assert np.array_equal(wire[:want.size], want)
        assert (wire[want.size:] == GUARD).all()
        assert np.array_equal(core.get_state(), st)


def test_wire_dense_frames_direct_store_path():
    # refrand frames: ~85 % of the bytes change, the expander takes its non-staged path
    w, h, T = 320, 64, 3
    n = 3 * w * h
    base = synth.refrand_frame(n, 100)
    frames = np.stack([synth.refrand_frame(n, 1 + t) for t in range(T)])
    off, xs, df, _ = po.diff_stream(frames, base)
    want = po.wire_pack(off, xs, df)
    with CUDACore(w, h, sample_mat_data=base, max_batch=T) as core:
        g_off, wire, _ = run_wire(core, frames)
        assert np.array_equal(g_off, off)
        assert np.array_equal(wire[:want.size], want)
        assert (wire[want.size:] == GUARD).all()


def test_wire_capacity_drops_whole_frames():
    w, h, T = 64, 48, 6
    base, frames = synth.webcam_stream(T, w, h, seed=9)
    off, xs, df, _ = po.diff_stream(frames, base)
    want = po.wire_pack(off, xs, df)
    starts = [4 * t + 5 * int(off[t]) for t in range(T + 1)]
    cap = starts[3] + 4 + 7            # frames 0..2 whole, header of frame 3, not its payload
    with CUDACore(w, h, sample_mat_data=base, max_batch=T) as core:
        g_off, wire, _ = run_wire(core, frames, capacity=cap)
        assert np.array_equal(g_off, off)                      # the index stays exact
        assert np.array_equal(wire[:starts[3] + 4], want[:starts[3] + 4])
        assert (wire[starts[3] + 4:] == GUARD).all()           # nothing past the header, nothing past cap


def test_wire_empty_frames_and_empty_batch():
    with CUDACore(0, 0, max_batch=4) as core:
        d_off = torch.full((4,), -1, dtype=torch.int32, device=DEV)
        d_wire = torch.full((64,), GUARD, dtype=torch.uint8, device=DEV)
        core.diff_stream_wire_batch(None, 3, d_off, d_wire, 64)
        core.synchronize()
        assert d_off.cpu().tolist() == [0, 0, 0, 0]
        assert d_wire.cpu().tolist() == [0] * 12 + [GUARD] * 52


@pytest.mark.parametrize("per_frame", [False, True])
@pytest.mark.parametrize("w,h,T", [(64, 48, 6), (33, 7, 5), (1, 1, 4), (640, 360, 3)])
def test_client_apply_matches_oracle(w, h, T, per_frame):
    base, frames = synth.webcam_stream(T, w, h, seed=13)
    off, xs, df, st = po.diff_stream(frames, base)
    shown, _ = po.wire_client(base, po.wire_pack(off, xs, df), T)
    n = 3 * w * h
    with CUDACore(w, h, sample_mat_data=base, max_batch=T) as client:
        d_off, d_xs, d_df = to_dev(off.view(np.int32)), to_dev(np.append(xs, np.int32(0))), to_dev(np.append(df, np.uint8(0)))
        d_out = torch.full((T, n), GUARD, dtype=torch.uint8, device=DEV) if per_frame else None
        client.apply_batch(d_off, d_xs, d_df, T, d_out)
        client.synchronize()
        assert np.array_equal(client.get_state(), st)          # client == server's reconstructed state
        assert np.array_equal(client.get_state(), shown[-1])
        if per_frame:
            assert np.array_equal(d_out.cpu().numpy(), shown)


def test_client_apply_repeated_indices_across_frames_commute():
    # the one-launch form relies on byte adds commuting: hammer the same few bytes from many frames
    n_w, n_h, T = 8, 2, 200
    n = 3 * n_w * n_h
    rng = np.random.default_rng(3)
    counts = rng.integers(0, n + 1, T)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
    xs = np.concatenate([np.sort(rng.choice(n, c, replace=False)) for c in counts]).astype(np.int32)
    df = rng.integers(1, 256, xs.size).astype(np.uint8)
    base = rng.integers(0, 256, n).astype(np.uint8)
    want = base.copy()
    for t in range(T):
        want = po.client_apply(want, xs[off[t]:off[t + 1]], df[off[t]:off[t + 1]])
    with CUDACore(n_w, n_h, sample_mat_data=base, max_batch=1) as client:
        client.apply_batch(to_dev(off.view(np.int32)), to_dev(xs), to_dev(df), T)
        client.synchronize()
        assert np.array_equal(client.get_state(), want)


def test_client_apply_ignores_out_of_range_indices():
    base = np.arange(48, dtype=np.uint8)
    off = np.array([0, 3], np.uint32)
    xs = np.array([5, 48, -1], np.int32)
    df = np.array([10, 99, 99], np.uint8)
    with CUDACore(4, 4, sample_mat_data=base, max_batch=1) as client:
        for out in (None, torch.zeros((1, 48), dtype=torch.uint8, device=DEV)):
            client.set_state(base)
            client.apply_batch(to_dev(off.view(np.int32)), to_dev(xs), to_dev(df), 1, out)
            client.synchronize()
            want = base.copy(); want[5] += 10
            assert np.array_equal(client.get_state(), want)


@pytest.mark.parametrize("per_frame", [False, True])
def test_client_apply_wire(per_frame):
    w, h, T = 96, 40, 7
    base, frames = synth.webcam_stream(T, w, h, seed=17)
    off, xs, df, st = po.diff_stream(frames, base)
    wire = po.wire_pack(off, xs, df)
    shown, counts = po.wire_client(base, wire, T)
    with CUDACore(w, h, sample_mat_data=base, max_batch=T) as client:
        d_out = torch.zeros((T, 3 * w * h), dtype=torch.uint8, device=DEV) if per_frame else None
        client.apply_wire_batch(to_dev(wire), counts, T, d_out)
        client.synchronize()
        assert np.array_equal(client.get_state(), st)
        if per_frame:
            assert np.array_equal(d_out.cpu().numpy(), shown)


def test_server_to_client_round_trip_1080p_on_device():
    # full size: server core packs to wire bytes, client core rebuilds from them; both states must agree
    # with each other and every reconstructed byte is within the threshold of the frame that was sent
    w, h, T = 1920, 1080, 48
    n = 3 * w * h
    base_d, frames_d = synth.webcam_stream(T, w, h, seed=21, device=DEV)
    base = base_d.cpu().numpy()
    with CUDACore(w, h, sample_mat_data=base, max_batch=T) as server, \
            CUDACore(w, h, sample_mat_data=base, max_batch=T) as client:
        cap = 4 * T + 5 * (T * n // 8)
        d_off = torch.zeros(T + 1, dtype=torch.int32, device=DEV)
        d_wire = torch.empty(cap, dtype=torch.uint8, device=DEV)
        server.diff_stream_wire_batch(frames_d, T, d_off, d_wire, cap)
        server.synchronize()
        off = d_off.cpu().numpy().view(np.uint32)
        assert server.wire_bytes(T, int(off[-1])) <= cap
        counts = np.diff(off.astype(np.int64)).astype(np.uint32)
        d_out = torch.empty((T, n), dtype=torch.uint8, device=DEV)
        client.apply_wire_batch(d_wire, counts, T, d_out)
        client.synchronize()
        s_state = torch.from_numpy(server.get_state()).to(DEV)
        assert torch.equal(s_state, torch.from_numpy(client.get_state()).to(DEV))
        assert torch.equal(d_out[-1], s_state)
        err = (d_out.to(torch.int16) - frames_d.to(torch.int16)).abs().max().item()
        assert err <= 20
        # and the array form gives the same bytes as the wire form
        server.set_state(base)
        off2, xs2, df2, _ = run_stream(server, frames_d, capacity=T * n // 8)
        assert np.array_equal(off2, off)
        assert np.array_equal(d_wire[:server.wire_bytes(T, int(off[-1]))].cpu().numpy(), po.wire_pack(off2, xs2, df2))


@pytest.mark.parametrize("w,h,T,parts", [(64, 48, 5, 2), (64, 50, 4, 3), (40, 9, 6, 8), (1920, 1080, 3, 8)])
def test_row_bands_merge_to_whole_frame_stream(w, h, T, parts):
    base, frames = synth.webcam_stream(T, w, h, seed=23)
    if (w, h) != (1920, 1080):
        off, xs, df, st = po.diff_stream(frames, base)
    n = 3 * w * h
    rows = [h * p // parts for p in range(parts + 1)]
    d_frames = to_dev(frames)
    p_off, p_xs, p_df, p_state = [], [], [], []
    for p in range(parts):
        b0, b1 = 3 * w * rows[p], 3 * w * rows[p + 1]
        with CUDACore(w, rows[p + 1] - rows[p], sample_mat_data=base[b0:b1], max_batch=T) as band:
            cap = T * (b1 - b0)
            d_o = torch.zeros(T + 1, dtype=torch.int32, device=DEV)
            d_x = torch.zeros(cap, dtype=torch.int32, device=DEV)
            d_d = torch.zeros(cap, dtype=torch.uint8, device=DEV)
            band.diff_stream_batch(d_frames.data_ptr() + b0, T, d_o, d_x, d_d, cap, stride=n)
            band.synchronize()
            o = d_o.cpu().numpy().view(np.uint32)
            x, d = d_x[:int(o[-1])].cpu().numpy(), d_d[:int(o[-1])].cpu().numpy()
            p_off.append(o); p_xs.append(x); p_df.append(d); p_state.append(band.get_state())
    part_base = np.concatenate([[0], np.cumsum([x.size for x in p_xs])])[:-1]
    bias = [3 * w * r for r in rows[:-1]]
    total = sum(x.size for x in p_xs)
    with CUDACore(w, h, max_batch=1) as core:
        d_off = torch.full((T + 1,), -1, dtype=torch.int32, device=DEV)
        d_xs = torch.full((total + 8,), -7, dtype=torch.int32, device=DEV)
        d_df = torch.full((total + 8,), GUARD, dtype=torch.uint8, device=DEV)
        core.merge_parts(to_dev(np.stack(p_off).view(np.int32)), part_base, bias, to_dev(np.concatenate(p_xs + [[0]]).astype(np.int32)),
                         to_dev(np.concatenate(p_df + [[0]]).astype(np.uint8)), T, d_off, d_xs, d_df, total)
        core.synchronize()
        g_off = d_off.cpu().numpy().view(np.uint32)
        g_xs, g_df = d_xs.cpu().numpy(), d_df.cpu().numpy()
    if (w, h) == (1920, 1080):
        with CUDACore(w, h, sample_mat_data=base, max_batch=T) as whole:
            off, xs, df, _ = run_stream(whole, d_frames, capacity=total + 8)
            st = whole.get_state()
    assert np.array_equal(g_off, off)
    assert np.array_equal(g_xs[:total], xs) and (g_xs[total:] == -7).all()
    assert np.array_equal(g_df[:total], df) and (g_df[total:] == GUARD).all()
    assert np.array_equal(np.concatenate(p_state), st)
