"""Integration of the drop-in boundary: the reference's OWN server/src/server.cpp main loop (GPU
branch, compiled unmodified into oracle/_ref/server_hip in the build container) runs against the
C++ `diff::cuda::CUDACore` drop-in + libmi355diff.so on the MI355X, fed by the synthetic ThreadsCore,
and its socket byte stream (server/src/threads.cpp:220-233) is checked against the CPU oracle and
replayed through the client's reconstruction (client/opencv.cpp:38-66)."""
import os
import subprocess

import numpy as np
import pytest

from cudavideostream_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "server_hip")


def run_server(tmp_path, base, frames, w, h, **env):
    fin, fout, fvis = (str(tmp_path / n) for n in ("in.bin", "out.bin", "vis.bin"))
    with open(fin, "wb") as f:
        f.write(np.array([w, h, frames.shape[0]], np.int32).tobytes())
        f.write(base.tobytes())
        f.write(frames.tobytes())
    e = dict(os.environ, REF_IN=fin, REF_OUT=fout, REF_VIS=fvis)
    e.update({k: str(v) for k, v in env.items()})
    subprocess.run([EXE], env=e, check=True, stdout=subprocess.DEVNULL, timeout=120)
    return np.fromfile(fout, dtype=np.uint8), np.fromfile(fvis, dtype=np.uint8)


def parse_wire(buf, n, nframes):
    """base frame, then per frame {u32 count, i32 xs[count], u8 diff[count]}."""
    base, at, out = buf[:n], n, []
    for _ in range(nframes):
        c = int(buf[at:at + 4].view(np.uint32)[0]); at += 4
        xs = buf[at:at + 4 * c].view(np.int32).copy(); at += 4 * c
        df = buf[at:at + c].copy(); at += c
        out.append((c, xs, df))
    assert at == buf.size
    return base, out


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/server_hip not built")
@pytest.mark.parametrize("w,h,T", [(96, 54, 6), (1920, 1080, 3)])
def test_reference_main_loop_on_the_drop_in(po, tmp_path, w, h, T):
    n = 3 * w * h
    if w >= 1920:
        import torch
        base, frames = synth.webcam_stream(T, w, h, seed=77, device="cuda:0")
        base, frames = base.cpu().numpy(), frames.cpu().numpy()
    else:
        base, frames = synth.webcam_stream(T, w, h, seed=77)
    wire, _ = run_server(tmp_path, base, frames, w, h)
    sent_base, per_frame = parse_wire(wire, n, T)
    assert np.array_equal(sent_base, base)
    client = base.copy()          # client/opencv.cpp:44-47
    state = base
    for t, (c, xs, df) in enumerate(per_frame):
        ec, exs, edf, state = po.diff_pack(frames[t], state)
        assert c == ec and np.array_equal(xs, exs) and np.array_equal(df, edf), f"frame {t}"
        client = po.client_apply(client, xs, df)   # client/opencv.cpp:64-66
        assert np.array_equal(client, state)


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/server_hip not built")
@pytest.mark.parametrize("vis", [1, 2, 3, 4, 5])
def test_reference_main_loop_visualizers(po, tmp_path, vis):
    from test_filters_gpu import oracle_exec
    w, h, T = 80, 45, 4
    n = 3 * w * h
    base, frames = synth.webcam_stream(T, w, h, seed=78)
    # server.cpp:43 builds the Gaussian kernel itself (sigma = K*K/6.0)
    k = po.gaussian_kernel(3, 1.5)
    wire, visbuf = run_server(tmp_path, base, frames, w, h, MI355_VISUALIZER=vis, MI355_NOISE_FILTER=1)
    _, per_frame = parse_wire(wire, n, T)
    vis_frames = visbuf.reshape(T, n)
    state = base
    for t in range(T):
        c, xs, df, state, show = oracle_exec(po, frames[t], state, vis, k, True, w, h)
        assert per_frame[t][0] == c and np.array_equal(per_frame[t][1], xs) and np.array_equal(per_frame[t][2], df)
        assert np.array_equal(vis_frames[t], show)
