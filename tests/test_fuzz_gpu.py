"""-m gpu: seeded random geometries against the oracle, through the C-ABI -- sizes, batch lengths, thresholds,
frame strides, pointer alignments and change densities nobody picked by hand."""
import numpy as np
import pytest

from cudavideostream_amd import lib

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from gpu_util import DEV, CUDACore, to_dev  # noqa: E402


def _frames(rng, base, T, density, thr):
    n = base.size
    out = np.empty((T, n), np.uint8)
    prev = base
    for t in range(T):
        f = prev.astype(np.int16) + rng.integers(-min(thr, 6), min(thr, 6) + 1, n)
        hit = rng.random(n) < density
        f[hit] = rng.integers(0, 256, int(hit.sum()))
        out[t] = f.clip(0, 255).astype(np.uint8)
        prev = out[t]
    return out


@pytest.mark.parametrize("seed", range(24))
def test_random_streams(po, seed):
    rng = np.random.default_rng(1000 + seed)
    w, h = int(rng.integers(1, 260)), int(rng.integers(1, 48))
    if seed % 6 == 0:
        w, h = int(rng.integers(300, 700)), int(rng.integers(60, 200))     # several expander groups
    T = int(rng.integers(1, 10))
    thr = int(rng.choice([0, 1, 5, 20, 20, 20, 64, 127]))
    density = float(rng.choice([0.0, 0.002, 0.02, 0.1, 0.5, 1.0]))
    n = 3 * w * h
    pad = int(rng.choice([0, 0, 16, 5, 64]))                               # frame stride = n + pad
    skew = int(rng.choice([0, 0, 16, 1, 7]))                               # byte offset of the first frame
    base = rng.integers(0, 256, n, dtype=np.uint8)
    frames = _frames(rng, base, T, density, thr)
    off, xs, df, st = po.diff_stream(frames, base, thr)
    buf = np.zeros(skew + T * (n + pad) + 64, np.uint8)
    for t in range(T):
        buf[skew + t * (n + pad): skew + t * (n + pad) + n] = frames[t]
    d_buf = to_dev(buf)
    cap = int(off[-1]) + 3
    d_off = torch.full((T + 1,), -1, dtype=torch.int32, device=DEV)
    d_xs = torch.full((cap,), -7, dtype=torch.int32, device=DEV)
    d_df = torch.full((cap,), 0xA5, dtype=torch.uint8, device=DEV)
    with CUDACore(w, h, threshold=thr, sample_mat_data=base, max_batch=T) as core:
        core.diff_stream_batch(d_buf.data_ptr() + skew, T, d_off, d_xs, d_df, cap, stride=n + pad)
        core.synchronize()
        assert np.array_equal(d_off.cpu().numpy().view(np.uint32), off)
        tot = int(off[-1])
        assert np.array_equal(d_xs.cpu().numpy()[:tot], xs) and (d_xs.cpu().numpy()[tot:] == -7).all()
        assert np.array_equal(d_df.cpu().numpy()[:tot], df) and (d_df.cpu().numpy()[tot:] == 0xA5).all()
        assert np.array_equal(core.get_state(), st)
        # the same stream as wire bytes, then through the client
        core.set_state(base)
        want = po.wire_pack(off, xs, df)
        d_wire = torch.full((want.size + 8,), 0x5C, dtype=torch.uint8, device=DEV)
        core.diff_stream_wire_batch(d_buf.data_ptr() + skew, T, d_off, d_wire, want.size, stride=n + pad)
        core.synchronize()
        wire = d_wire.cpu().numpy()
        assert np.array_equal(wire[:want.size], want) and (wire[want.size:] == 0x5C).all()
    with CUDACore(w, h, sample_mat_data=base, max_batch=T) as client:
        client.apply_wire_batch(d_wire, np.diff(off.astype(np.int64)).astype(np.uint32), T)
        client.synchronize()
        assert np.array_equal(client.get_state(), st)


@pytest.mark.parametrize("seed", range(12))
def test_random_filters(po, seed):
    rng = np.random.default_rng(2000 + seed)
    w, h = int(rng.integers(1, 200)), int(rng.integers(1, 40))
    if seed % 4 == 0:
        w = 16 * int(rng.integers(1, 24))                                   # 16-byte rows: the vector paths
    elif seed % 4 == 2:
        w, h = 8 * int(rng.integers(1, 48)), int(rng.integers(1, 90))       # 8-byte rows: the median's column strips, several bands
    n = 3 * w * h
    cur = rng.integers(0, 256, n, dtype=np.uint8)
    prev = np.where(rng.random(n) < 0.7, cur, rng.integers(0, 256, n)).astype(np.uint8)
    k = rng.random(9).astype(np.float32)
    k = (k / k.sum()).astype(np.float32)
    if seed % 2:
        k = po.gaussian_kernel(3, float(rng.uniform(0.6, 2.5)))
    with CUDACore(w, h, k=k) as core:
        d_cur, d_prev = to_dev(cur), to_dev(prev)

        def run(fn, *args):
            d_o = torch.full((n + 16,), 0x3C, dtype=torch.uint8, device=DEV)
            torch.cuda.synchronize()    # the fill runs on torch's stream, the core on its own
            fn(*args, d_o)
            core.synchronize()
            got = d_o.cpu().numpy()
            assert (got[n:] == 0x3C).all()
            return got[:n]

        assert np.array_equal(run(core.gray_avg, d_cur), po.gray_avg(cur))
        gw = po.gray_weighted(cur)
        assert np.array_equal(run(core.gray_weighted, d_cur), gw)
        assert np.array_equal(run(core.heat_map, d_cur, d_prev), po.heat_map(cur, prev))
        assert np.array_equal(run(core.red_dense, d_cur, d_prev), po.red_dense(cur, prev))
        assert np.array_equal(run(core.conv3x3, d_cur), po.conv3x3(cur, w, h, k))
        assert np.array_equal(run(core.median5x5, d_cur), po.median5x5(cur, w, h))
        thr = po.two_max_threshold(po.histogram(gw))
        assert np.array_equal(run(core.binarize_chain, to_dev(gw)), po.binarize(gw, thr))
        d_o = torch.zeros((1, n), dtype=torch.uint8, device=DEV)
        core.filter_batch(lib.OP_GRAY_WEIGHTED_BINARIZE, d_cur, d_o, 1)
        core.synchronize()
        assert np.array_equal(d_o.cpu().numpy()[0], po.binarize(gw, thr))
