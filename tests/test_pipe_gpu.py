"""-m gpu parity tests of the pipelined per-frame path (mi355_pipe_*): the same
outputs as exec_core (server/src/kernels.cu:430-525) for every frame, with several frames in flight."""
import numpy as np
import pytest

from cudavideostream_amd import lib, synth

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from gpu_util import DEV, CUDACore  # noqa: E402
from test_filters_gpu import oracle_exec  # noqa: E402


class Ring:
    """depth sets of pinned buffers, as threads.cpp:95 allocates its six."""

    def __init__(self, depth, w, h):
        self.sets = [CUDACore.alloc_arrays(h, w) for _ in range(depth)]

    def __getitem__(self, i):
        return self.sets[i % len(self.sets)]

    def free(self):
        for s in self.sets:
            for a in s:
                a.free()


@pytest.mark.parametrize("vis", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("noise_filter,depth", [(False, 3), (True, 2)])
def test_pipe_matches_exec_core_semantics(po, vis, noise_filter, depth):
    w, h, T = 96, 54, 7
    base, frames = synth.webcam_stream(T, w, h, seed=60 + vis)
    k = po.gaussian_kernel(3, 1.5)
    n = 3 * w * h
    ring = Ring(depth, w, h)
    with CUDACore(w, h, k=k, sample_mat_data=base, visualizer=vis, noise_filter=noise_filter) as core:
        core.pipe_open(depth)
        want, state = [], base
        for t in range(T):
            c, xs, df, state, show = oracle_exec(po, frames[t], state, vis, k, noise_filter, w, h)
            want.append((c, xs, df, show))
        tickets = {}

        def finish(t):
            h_frame, n_frame, _, h_xs = ring[t]
            pos = core.exec_wait(tickets.pop(t))
            c, xs, df, show = want[t]
            assert pos == c
            assert np.array_equal(h_xs.array[:pos], xs)
            assert np.array_equal(h_frame.array[:pos], df)
            if show is not None:
                assert np.array_equal(n_frame.array[:n], show)

        for t in range(T):
            if t >= depth:
                finish(t - depth)          # the slot's buffers are about to be reused
            h_frame, n_frame, _, h_xs = ring[t]
            h_frame.array[:n] = frames[t]
            tickets[t] = core.exec_submit(h_frame.array, n_frame.array, "", h_xs.array)
        for t in sorted(tickets):
            finish(t)
        assert np.array_equal(core.get_state(), state)
        core.pipe_close()
    ring.free()


def test_pipe_1080p_stream_in_flight(po):
    w, h, T, depth = 1920, 1080, 12, 4
    base, frames = synth.webcam_stream(T, w, h, device=DEV)
    base, frames = base.cpu().numpy(), frames.cpu().numpy()
    n = 3 * w * h
    ring = Ring(depth, w, h)
    with CUDACore(w, h, sample_mat_data=base) as core:
        core.pipe_open(depth)
        off, xs, df, st = po.diff_stream(frames, base)
        tickets = []
        for t in range(T):
            if t >= depth:
                check_frame(core, ring, tickets, t - depth, off, xs, df)
            h_frame, _, _, h_xs = ring[t]
            h_frame.array[:n] = frames[t]
            tickets.append(core.exec_submit(h_frame.array, None, "", h_xs.array))
        for t in range(T - depth, T):
            check_frame(core, ring, tickets, t, off, xs, df)
        assert np.array_equal(core.get_state(), st)
    ring.free()


def check_frame(core, ring, tickets, t, off, xs, df):
    h_frame, _, _, h_xs = ring[t]
    pos = core.exec_wait(tickets[t])
    assert pos == off[t + 1] - off[t]
    assert np.array_equal(h_xs.array[:pos], xs[off[t]:off[t + 1]])
    assert np.array_equal(h_frame.array[:pos], df[off[t]:off[t + 1]])


def test_pipe_full_ring_completes_oldest_and_ticket_rules(po):
    w, h, T, depth = 64, 48, 5, 2
    base, frames = synth.webcam_stream(T, w, h, seed=71)
    n = 3 * w * h
    ring = Ring(T, w, h)                      # one buffer set per frame: nothing is overwritten
    with CUDACore(w, h, sample_mat_data=base) as core:
        with pytest.raises(lib.Mi355Error, match="pipe not open"):
            core.exec_submit(ring[0][0].array, None, "", ring[0][3].array)
        core.pipe_open(depth)
        with pytest.raises(lib.Mi355Error, match="already open"):
            core.pipe_open(depth)
        with pytest.raises(lib.Mi355Error, match="pipe open"):
            core.exec_core(ring[0][0].array, None, "", ring[0][3].array)
        with pytest.raises(lib.Mi355Error, match="pinned"):
            core.exec_submit(np.zeros(n + 32, np.uint8), None, "", ring[0][3].array)
        tickets = []
        for t in range(T):                    # never waits: the ring (depth 2) is overrun on purpose
            ring[t][0].array[:n] = frames[t]
            tickets.append(core.exec_submit(ring[t][0].array, None, "", ring[t][3].array))
        assert tickets == list(range(T))
        with pytest.raises(lib.Mi355Error, match="already waited for or overwritten"):
            core.exec_wait(tickets[0])
        with pytest.raises(lib.Mi355Error, match="unknown ticket"):
            core.exec_wait(T)
        pos = core.exec_wait(tickets[-1])
        with pytest.raises(lib.Mi355Error, match="already waited"):
            core.exec_wait(tickets[-1])
        core.exec_wait(tickets[-2])
        # frames whose tickets were overrun were still processed, in order, into their own buffers
        off, xs, df, st = po.diff_stream(frames, base)
        assert pos == off[T] - off[T - 1]
        for t in range(T):
            c = int(off[t + 1] - off[t])
            assert np.array_equal(ring[t][3].array[:c], xs[off[t]:off[t + 1]])
            assert np.array_equal(ring[t][0].array[:c], df[off[t]:off[t + 1]])
        assert np.array_equal(core.get_state(), st)
        core.pipe_close()
        # closed: the blocking entry point works again
        ring[0][0].array[:n] = frames[0]
        core.set_state(base)
        assert core.exec_core(ring[0][0].array, None, "", ring[0][3].array) == off[1]
    ring.free()


def test_pipe_text_overlay_and_unaligned_outputs(po):
    w, h = 96, 54
    gh, gw = 7, 5
    charset = "0123456789BFPSWbkps :/"
    rng = np.random.default_rng(1)
    atlas = rng.integers(0, 256, (len(charset), gh, gw * 3), dtype=np.uint8)
    base, frames = synth.webcam_stream(1, w, h, seed=40)
    text = "FPS: 25"
    exp = frames[0].reshape(h, w * 3).copy()
    for j, ch in enumerate(text):
        exp[:gh, j * gw * 3:(j + 1) * gw * 3] = atlas[charset.index(ch)]
    exp = exp.reshape(-1)
    n = 3 * w * h
    with CUDACore(w, h, sample_mat_data=base, chars_px=atlas, chars_sz=(gh, gw), charset=charset) as core:
        h_frame, n_frame, o_frame, h_xs = CUDACore.alloc_arrays(h, w)
        core.pipe_open(1)
        # outputs at odd offsets inside the pinned blocks: k_export's scalar path
        frame_view = n_frame.array[3:3 + n]
        frame_view[:] = frames[0]
        xs_view = h_xs.array[1:]
        pos = core.exec_wait(core.exec_submit(frame_view.ctypes.data, None, text, xs_view.ctypes.data))
        c, xs, df, st = po.diff_pack(exp, base)
        assert pos == c and np.array_equal(xs_view[:pos], xs) and np.array_equal(frame_view[:pos], df)
        assert np.array_equal(core.get_state(), st)
        for a in (h_frame, n_frame, o_frame, h_xs):
            a.free()
