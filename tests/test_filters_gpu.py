"""GPU parity of the filter kernels (gray, binarize chain, heat map, red map, 3x3 noise filter) and of
the per-frame host entry point exec_core, bit-exact against the CPU oracle and the fixtures recorded
from the reference's own server.cpp CPU branch."""
import os

import numpy as np
import pytest

from conftest import golden
from cudavideostream_amd import lib, synth

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from gpu_util import DEV, CUDACore, to_dev  # noqa: E402


def dev_out(n):
    return torch.full((n,), 0x5A, dtype=torch.uint8, device=DEV)


SIZES = [(64, 48), (37, 11), (1, 1), (17, 3), (200, 9), (1920, 8)]


@pytest.mark.parametrize("w,h", SIZES)
def test_gray_heat_red(po, w, h):
    rng = np.random.default_rng(w + h)
    n = 3 * w * h
    img = rng.integers(0, 256, n, dtype=np.uint8)
    prv = np.clip(img.astype(int) + rng.integers(-60, 61, n), 0, 255).astype(np.uint8)
    d_img, d_prv = to_dev(img), to_dev(prv)
    with CUDACore(w, h) as core:
        # (the method is looked up AFTER the output buffer is made: the test wrapper drains torch's streams at the
        # look-up, and torch.full above runs on torch's stream, the kernel on the core's own -- a method bound before
        # dev_out() raced with the fill once in a dozen runs)
        for name, exp in (("gray_avg", po.gray_avg(img)), ("gray_weighted", po.gray_weighted(img))):
            d_o = dev_out(n)
            getattr(core, name)(d_img, d_o); core.synchronize()
            assert np.array_equal(d_o.cpu().numpy(), exp)
        d_o = dev_out(n)
        core.heat_map(d_img, d_prv, d_o); core.synchronize()
        assert np.array_equal(d_o.cpu().numpy(), po.heat_map(img, prv))
        d_o = dev_out(n)
        core.red_dense(d_img, d_prv, d_o); core.synchronize()
        assert np.array_equal(d_o.cpu().numpy(), po.red_dense(img, prv))
        # in place + unaligned pointers
        d_pad = torch.zeros(n + 1, dtype=torch.uint8, device=DEV)
        d_pad[1:] = d_img
        core.gray_avg(d_pad.data_ptr() + 1, d_pad.data_ptr() + 1); core.synchronize()
        assert np.array_equal(d_pad[1:].cpu().numpy(), po.gray_avg(img))


def test_heat_map_spot_values_of_the_reference_expression():
    """tests/heat_map_benchmark/cpu.cu:19-27 evaluated by hand, normaliser 510: the colour of
    d = |dB|+|dG|+|dR| is (r,g,b) = d 0 -> (0,0,255), 1 -> (0,1,254), 255 -> (0,255,0), 510 -> (255,0,0),
    600 -> (216,0,0), 765 -> (0,0,0) -- red falls off again above 510.  No oracle involved."""
    spots = {0: (0, 0, 255), 1: (0, 1, 254), 255: (0, 255, 0), 510: (255, 0, 0), 600: (216, 0, 0), 765: (0, 0, 0)}
    ds = sorted(spots)
    cur = np.zeros((len(ds), 3), np.uint8)
    for i, d in enumerate(ds):                  # spread d over the three channels of one pixel
        cur[i] = (min(d, 255), min(max(d - 255, 0), 255), min(max(d - 510, 0), 255))
    prv = np.zeros_like(cur)
    with CUDACore(len(ds), 1) as core:
        for a, b in ((cur, prv), (prv, cur)):   # |.| per channel: direction does not matter
            d_o = dev_out(cur.size)
            core.heat_map(to_dev(a.reshape(-1)), to_dev(b.reshape(-1)), d_o); core.synchronize()
            out = d_o.cpu().numpy().reshape(-1, 3)
            for i, d in enumerate(ds):
                r, g, bl = spots[d]
                assert tuple(out[i]) == (bl, g, r), (d, tuple(out[i]))     # stored B,G,R (cpu.cu:62-64)


def test_gray_weighted_exhaustive_2_24(po):
    """Every (B, G, R) triple against the double expression of tests/grayscale-weighted/cpu.cu:40."""
    w, h = 4096, 4096
    v = torch.arange(1 << 24, dtype=torch.int32, device=DEV)
    bgr = torch.stack([v & 255, (v >> 8) & 255, v >> 16], dim=1).to(torch.uint8).reshape(-1)
    with CUDACore(w, h) as core:
        d_o = dev_out(bgr.numel())
        core.gray_weighted(bgr, d_o); core.synchronize()
        got = d_o.reshape(-1, 3)
        assert bool((got[:, 0] == got[:, 1]).all()) and bool((got[:, 0] == got[:, 2]).all())
        got = got[:, 0].cpu().numpy()
    x = np.arange(1 << 24, dtype=np.int64)
    exp = (0.114 * (x & 255).astype(np.float64) + 0.587 * ((x >> 8) & 255) + 0.299 * (x >> 16)).astype(np.uint8)
    assert np.array_equal(got, exp)
    s = golden("oracle_gray_weighted_sample.npz")
    idx = s["bgr"][:, 0].astype(np.int64) | (s["bgr"][:, 1].astype(np.int64) << 8) | (s["bgr"][:, 2].astype(np.int64) << 16)
    assert np.array_equal(got[idx], s["gray"])


def test_binarize_chain_vs_reference_fixture(po):
    """gray-avg + binarize chain == outputs recorded from the reference's server.cpp CPU branch."""
    g = golden("ref_server_cpu_64x48.npz")
    w, h = int(g["width"]), int(g["height"])
    n = 3 * w * h
    with CUDACore(w, h) as core:
        d_hist = torch.zeros(256, dtype=torch.int32, device=DEV)
        d_thr = torch.zeros(1, dtype=torch.int32, device=DEV)
        for t in range(g["frames"].shape[0]):
            d_in = to_dev(g["frames"][t])
            d_gray, d_o = dev_out(n), dev_out(n)
            core.gray_avg(d_in, d_gray)
            core.binarize_chain(d_gray, d_o, d_hist, d_thr)
            core.synchronize()
            assert np.array_equal(d_o.cpu().numpy(), g["out"][t]), f"frame {t}"
            gray = po.gray_avg(g["frames"][t])
            assert np.array_equal(d_hist.cpu().numpy(), po.histogram(gray))
            assert int(d_thr.item()) == po.two_max_threshold(po.histogram(gray))


def test_two_max_threshold_shapes_of_histogram(po):
    """The threshold kernel finds the last two prefix-maximum records with a wave-wide max-scan; the coded loop of
    server.cpp:108-127 (the oracle, dead branch included) is sequential.  Histograms built to order: ties,
    plateaus, strictly rising / falling, a single bin, records at lane boundaries (bins 4k-1, 4k), the clamps."""
    rng = np.random.default_rng(8)
    shapes = []
    shapes.append(np.full(256, 5))                                   # all equal: every bin is a record -> (255+254)/2
    shapes.append(np.arange(256) + 1)                                # rising
    shapes.append(256 - np.arange(256))                              # falling: only bin 0 -> (0 + -1)/2 = 0 -> clamp 50
    one = np.zeros(256, int); one[137] = 9; shapes.append(one)       # bin 0 (count 0) then bin 137
    for k in (3, 4, 63, 64, 127, 128, 251, 252):
        v = np.zeros(256, int); v[0] = 3; v[k] = 3; v[min(k + 1, 255)] = 2; shapes.append(v)
        v = np.zeros(256, int); v[k] = 7; v[k - 1] = 7; shapes.append(v)
    for _ in range(40):
        v = rng.integers(0, 6, 256); shapes.append(v)
        v = np.sort(rng.integers(0, 50, 256)); v[rng.integers(0, 256, 20)] = 0; shapes.append(v)
    T = len(shapes)
    npix = max(int(np.max([v.sum() for v in shapes])), 1)
    w, h = npix, 1
    n = 3 * npix
    frames = np.zeros((T, n), np.uint8)
    for t, v in enumerate(shapes):
        vals = np.repeat(np.arange(256, dtype=np.uint8), v)
        pad = np.full(npix - vals.size, vals[0] if vals.size else 0, np.uint8)   # pad with the first value present
        frames[t] = np.repeat(np.concatenate([vals, pad]), 3)
    with CUDACore(w, h, max_batch=T) as core:
        d_out = torch.zeros((T, n), dtype=torch.uint8, device=DEV)
        core.filter_batch(lib.OP_BINARIZE, to_dev(frames), d_out, T)
        core.synchronize()
        got = d_out.cpu().numpy()
    seen = set()
    for t in range(T):
        thr = po.two_max_threshold(po.histogram(frames[t]))
        seen.add(thr)
        assert np.array_equal(got[t], po.binarize(frames[t], thr)), (t, thr)
    assert 50 in seen and 200 in seen and len(seen) >= 5


@pytest.mark.parametrize("w,h", [(64, 48), (37, 11), (1920, 1080)])
def test_binarize_chain_weighted(po, w, h):
    """config 3: weighted gray -> histogram -> two-max -> binarize."""
    img = synth.webcam_frame(2, w, h, seed=6)
    n = img.size
    with CUDACore(w, h) as core:
        d_gray, d_o = dev_out(n), dev_out(n)
        core.gray_weighted(to_dev(img), d_gray)
        core.binarize_chain(d_gray, d_o)
        core.synchronize()
        gw = po.gray_weighted(img)
        assert np.array_equal(d_o.cpu().numpy(), po.binarize(gw, po.two_max_threshold(po.histogram(gw))))


@pytest.mark.parametrize("w,h", [(64, 48), (3, 3), (1, 1), (65, 9), (130, 17), (1920, 1080)])
def test_conv3x3(po, w, h):
    rng = np.random.default_rng(w * 7 + h)
    img = rng.integers(0, 256, 3 * w * h, dtype=np.uint8)
    k = po.gaussian_kernel(3, 1.5)
    with CUDACore(w, h, k=k) as core:
        d_o = dev_out(img.size)
        core.conv3x3(to_dev(img), d_o); core.synchronize()
        assert np.array_equal(d_o.cpu().numpy(), po.conv3x3(img, w, h, k))
    kk = rng.random(9).astype(np.float32)
    kk /= kk.sum()   # mean-like kernel with awkward roundings
    with CUDACore(w, h, k=kk) as core:
        d_o = dev_out(img.size)
        core.conv3x3(to_dev(img), d_o); core.synchronize()
        assert np.array_equal(d_o.cpu().numpy(), po.conv3x3(img, w, h, kk))


@pytest.mark.parametrize("w,h", [(16, 1), (16, 2), (16, 30), (16, 31), (32, 61), (352, 95), (3840, 66)])
def test_conv3x3_column_strips(po, w, h):
    """Geometries of k_conv3x3_strip (row bytes a multiple of 16): strips of 30 rows with 1..30 rows in the
    last one, one to many 16-byte columns, with the symmetric (shared products) and the general form."""
    rng = np.random.default_rng(w + 13 * h)
    img = rng.integers(0, 256, 3 * w * h, dtype=np.uint8)
    img[:3 * w] = 255                      # saturated first row against the zero padding
    kernels = [po.gaussian_kernel(3, 1.5),
               np.array([0.05, 0.1, 0.05, 0.1, 0.4, 0.1, 0.05, 0.1, 0.05], np.float32),      # symmetric
               np.array([0.0625, 0.125, 0.0625, 0.125, 0.25, 0.125, 0.0625, 0.125, 0.0625], np.float32),
               (lambda k: (k / k.sum()).astype(np.float32))(rng.random(9))]                 # general
    for k in kernels:
        with CUDACore(w, h, k=k) as core:
            d_o = dev_out(img.size)
            core.conv3x3(to_dev(img), d_o); core.synchronize()
            assert np.array_equal(d_o.cpu().numpy(), po.conv3x3(img, w, h, k))


@pytest.mark.parametrize("w,h", [(64, 48), (1, 1), (2, 5), (5, 2), (65, 17), (131, 33), (640, 360)])
def test_median5x5(po, w, h):
    """tests/noise_filter_benchmark/v3.cu:32-90 restated (swap sort, middle element) against the selection
    network of the kernel: random bytes, a constant frame (borders see the zero padding) and salt noise."""
    rng = np.random.default_rng(5 * w + h)
    n = 3 * w * h
    salt = np.full(n, 90, np.uint8)
    salt[rng.random(n) < 0.1] = 255
    for img in (rng.integers(0, 256, n, dtype=np.uint8), np.full(n, 200, np.uint8), salt):
        with CUDACore(w, h) as core:
            d_o = dev_out(n)
            core.median5x5(to_dev(img), d_o); core.synchronize()
            assert np.array_equal(d_o.cpu().numpy(), po.median5x5(img, w, h))
    with CUDACore(w, h, max_batch=3) as core:       # batched, with a frame stride larger than the frame
        frames = rng.integers(0, 256, (3, n + 16), dtype=np.uint8)
        d_in, d_out = to_dev(frames), torch.zeros((3, n + 16), dtype=torch.uint8, device=DEV)
        core.filter_batch(lib.OP_MEDIAN5X5, d_in, d_out, 3, stride=n + 16)
        core.synchronize()
        got = d_out.cpu().numpy()
        for t in range(3):
            assert np.array_equal(got[t, :n], po.median5x5(frames[t, :n], w, h))
            assert (got[t, n:] == 0).all()


@pytest.mark.parametrize("w,h", [(8, 1), (8, 7), (16, 21), (24, 3), (168, 23), (336, 41), (64, 100), (176, 61)])
def test_median5x5_strip_kernel(po, w, h):
    """Rows that are a multiple of 8 bytes take the column-strip kernel (two row bands per register, shared column sorts,
    the selection program of csrc/median_net.h): one strip wave (w = 8: three strips beside the two lanes that only
    supply columns), a row that ends one strip into the second wave (168 pixels = 63 strips) and two into the third
    (336), heights below, at and beyond one and two bands for every band length the option allows -- the band pairing,
    the steps that round a band up to five rows and the rows below the image.  Saturated frames see the zero padding
    on every border; every byte value occurs."""
    rng = np.random.default_rng(31 * w + h)
    n = 3 * w * h
    ramp = (np.arange(n, dtype=np.int64) * 7 % 256).astype(np.uint8)
    for rows in (0, 5, 10, 20, 25, 40, 60):
        for img in (rng.integers(0, 256, n, dtype=np.uint8), np.full(n, 255, np.uint8), ramp):
            with CUDACore(w, h) as core:
                core.set_option(lib.OPT_MEDIAN_ROWS, rows)
                d_o = dev_out(n)
                core.median5x5(to_dev(img), d_o); core.synchronize()
                assert np.array_equal(d_o.cpu().numpy(), po.median5x5(img, w, h)), rows
    with CUDACore(w, h, max_batch=5) as core:       # batched, frame stride larger than the frame, guard bytes untouched
        frames = rng.integers(0, 256, (5, n + 24), dtype=np.uint8)
        d_in, d_out = to_dev(frames), torch.full((5, n + 24), 7, dtype=torch.uint8, device=DEV)
        core.filter_batch(lib.OP_MEDIAN5X5, d_in, d_out, 5, stride=n + 24)
        core.synchronize()
        got = d_out.cpu().numpy()
        for t in range(5):
            assert np.array_equal(got[t, :n], po.median5x5(frames[t, :n], w, h))
            assert (got[t, n:] == 7).all()


@pytest.mark.parametrize("w,h", [(16, 30), (65, 17), (352, 95), (7, 5)])
def test_conv3x3_sharpen_and_edge_kernels_saturate_the_same_way(po, w, h):
    """Negative taps and sums above 255: the column-strip kernel (row bytes a multiple of 16) and the general
    one use one conversion (truncate, saturate to [0, 255]), the oracle the same."""
    rng = np.random.default_rng(w * 3 + h)
    img = rng.integers(0, 256, 3 * w * h, dtype=np.uint8)
    kernels = [np.array([0, -1, 0, -1, 5, -1, 0, -1, 0], np.float32),                  # sharpen
               np.array([-1, -1, -1, -1, 8, -1, -1, -1, -1], np.float32),              # edge
               np.array([0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5, 0.5], np.float32),    # gain 4.5
               np.array([0.1, -0.3, 0.7, 1.1, -0.9, 0.2, 0.6, -0.2, 0.4], np.float32)]
    for k in kernels:
        want = po.conv3x3(img, w, h, k)
        assert (want == 0).any() or (want == 255).any()      # the kernel does leave [0, 255]
        with CUDACore(w, h, k=k) as core:
            d_o = dev_out(img.size)
            core.conv3x3(to_dev(img), d_o); core.synchronize()
            assert np.array_equal(d_o.cpu().numpy(), want)


def test_conv3x3_conversion_sweep(po):
    """float -> byte of the strip kernel (v_cvt_pk_u8_f32 under round-toward-zero, one instruction per byte) against the
    oracle's truncate-and-saturate for products all over the number line: a centre-only kernel g makes every output
    trunc_sat(fl(g * x)) for x = 0..255 -- gains that land on .5, just below an integer, at 254.99 / 255.0, negative, huge,
    infinite and NaN ones, and a few hundred random ones; then the same gains as ONE tap of an asymmetric kernel (the
    general form of the kernel) with the other taps zero."""
    w, h = 128, 8
    img = np.tile(np.arange(256, dtype=np.uint8), 3 * w * h // 256)
    rng = np.random.default_rng(77)
    gains = [0.5, 1.5, 0.25, 0.75, 1.0, 0.99999994, 1.0000001, 0.3333333, 0.6666667, 2.0, 127.99999 / 255, 254.99 / 255,
             254.999 / 255, 255.5 / 255, 1e-30, -1e-30, -0.5, -1.0, 3.0e38, -3.0e38, float("inf"), float("-inf"), float("nan"),
             1.0 / 3, 1.0 / 7, 255.0 / 256, 1.0039216]
    gains += list(rng.uniform(-2.0, 3.0, 200)) + list(np.exp(rng.uniform(-8, 8, 100)))
    with np.errstate(all="ignore"):
        for g in gains:
            for sym in (True, False):
                k = np.zeros(9, np.float32)
                k[4 if sym else 5] = np.float32(g)       # (0, 0, 0, 0, g, ...) is "symmetric" bit for bit; one off-centre tap is not
                want = po.conv3x3(img, w, h, k)
                with CUDACore(w, h, k=k) as core:
                    d_o = dev_out(img.size)
                    core.conv3x3(to_dev(img), d_o); core.synchronize()
                    got = d_o.cpu().numpy()
                assert np.array_equal(got, want), (g, sym, np.flatnonzero(got != want)[:5])


def test_conv_requires_kernel_and_out_of_place():
    with CUDACore(8, 8) as core:
        d = dev_out(192)
        with pytest.raises(lib.Mi355Error):
            core.conv3x3(d, dev_out(192))
    with CUDACore(8, 8, k=np.ones(9, np.float32) / 9) as core:
        d = dev_out(192)
        with pytest.raises(lib.Mi355Error):
            core.conv3x3(d, d)


def test_red_overlap(po):
    w, h = 64, 48
    base, frames = synth.webcam_stream(1, w, h, seed=12)
    c, xs, df, _ = po.diff_pack(frames[0], base)
    with CUDACore(w, h) as core:
        d_img = to_dev(base)
        core.red_overlap(d_img, to_dev(xs), None, c); core.synchronize()
        assert np.array_equal(d_img.cpu().numpy(), po.red_overlap(base, xs))
        d_img = to_dev(base)
        d_cnt = torch.tensor([c], dtype=torch.int32, device=DEV)
        core.red_overlap(d_img, to_dev(xs), d_cnt); core.synchronize()
        assert np.array_equal(d_img.cpu().numpy(), po.red_overlap(base, xs))


@pytest.mark.parametrize("clear", [True, False])
def test_red_stream_batch(po, clear):
    """The red motion maps of a whole batch from its packed stream = red_black_map_overlap per frame
    (kernels.cu:273-281,513-518), on zeroed frames or on top of given ones, with a padded frame stride."""
    w, h, T = 96, 40, 5
    n = 3 * w * h
    base, frames = synth.webcam_stream(T, w, h, seed=90)
    off, xs, df, _ = po.diff_stream(frames, base)
    rng = np.random.default_rng(0)
    canvas = rng.integers(0, 200, (T, n + 32), dtype=np.uint8)
    with CUDACore(w, h, max_batch=T) as core:
        d_canvas = to_dev(canvas)
        core.red_stream_batch(to_dev(off.view(np.int32)), to_dev(np.append(xs, np.int32(0))), T, d_canvas,
                              clear=clear, stride=n + 32)
        core.synchronize()
        got = d_canvas.cpu().numpy()
        for t in range(T):
            start = np.zeros(n, np.uint8) if clear else canvas[t, :n]
            assert np.array_equal(got[t, :n], po.red_overlap(start, xs[off[t]:off[t + 1]]))
            assert np.array_equal(got[t, n:], canvas[t, n:])      # the padding between frames is untouched


@pytest.mark.parametrize("w,h", [(640, 360), (1920, 1080), (333, 77)])
def test_red_stream_clear_many_slices(po, w, h):
    """The cleared form writes each frame slice by slice from a 256-ary search in the frame's ascending indices:
    sparse, dense (millions of entries: several search levels), empty and single-entry frames; ragged last slice."""
    n = 3 * w * h
    rng = np.random.default_rng(w)
    base = rng.integers(0, 256, n, dtype=np.uint8)
    sparse = np.where(rng.random(n) < 0.01, base ^ 0x80, base).astype(np.uint8)
    dense = synth.refrand_frame(n, 5)
    one = base.copy(); one[n // 2] ^= 0xFF
    cur = np.stack([sparse, dense, base, one, base ^ 0x80])
    prev = np.stack([base, synth.refrand_frame(n, 6), base, base, base])
    T = cur.shape[0]
    offs, xs = [0], []
    for t in range(T):
        c, x, _, _ = po.diff_pack(cur[t], prev[t])
        offs.append(offs[-1] + c); xs.append(x)
    xs = np.concatenate(xs + [np.zeros(1, np.int32)])
    stride = (n + 15) // 16 * 16 + 16
    canvas = rng.integers(1, 200, (T, stride), dtype=np.uint8)
    with CUDACore(w, h, max_batch=T) as core:
        d_canvas = to_dev(canvas)
        core.red_stream_batch(to_dev(np.array(offs, np.uint32).view(np.int32)), to_dev(xs), T, d_canvas, clear=True,
                              stride=stride)
        core.synchronize()
        got = d_canvas.cpu().numpy()
        for t in range(T):
            want = po.red_overlap(np.zeros(n, np.uint8), xs[offs[t]:offs[t + 1]])
            assert np.array_equal(got[t, :n], want), t
            assert np.array_equal(got[t, n:], canvas[t, n:])


def test_red_stream_clear_drops_entries_outside_their_slice():
    """A caller-built stream that is NOT ascending (the cleared form's search then hands a wave entries of other
    slices): such entries are dropped; no pixel outside the owning slice changes and the padding stays untouched."""
    w, h = 640, 360
    n = 3 * w * h
    rng = np.random.default_rng(7)
    good = np.sort(rng.choice(n, 5000, replace=False)).astype(np.int32)
    xs = good.copy()
    # entries far away from where the ascending order would put them (a later slice's index early in the stream
    # and the other way round), plus indices beyond the frame
    xs[10], xs[4000] = good[4900], good[5]
    xs[2500] = np.int32(n + 12345)
    stride = (n + 15) // 16 * 16 + 16
    canvas = rng.integers(1, 200, (1, stride), dtype=np.uint8)
    with CUDACore(w, h, max_batch=1) as core:
        d_canvas = to_dev(canvas)
        core.red_stream_batch(to_dev(np.array([0, xs.size], np.uint32).view(np.int32)), to_dev(xs), 1, d_canvas,
                              clear=True, stride=stride)
        core.synchronize()
        got = d_canvas.cpu().numpy()[0]
    assert np.array_equal(got[n:], canvas[0, n:])
    painted = np.flatnonzero(got[:n])
    assert np.all(got[painted] == 255) and np.all(painted % 3 == 2)
    # every painted pixel is owned by an in-range entry of the stream (no stray writes into neighbouring slices)
    inrange = xs[(xs >= 0) & (xs < n)]
    owners = set((inrange - inrange % 3 + 2).tolist())
    assert set(painted.tolist()) <= owners
    # (which of the other entries still land in the slice the search assigns them to is not defined for such a stream)
    assert painted.size > 0


# ---- exec_core: the per-frame host path (kernels.cu:430-525) ---------------------------------------

def oracle_exec(po, frame, state, vis, k, noise_filter, w, h):
    """The reference's exec_core sequence on the CPU: [conv] -> [vis] -> diff -> [red]."""
    cur = po.conv3x3(frame, w, h, k) if noise_filter else frame
    show = None
    if vis == lib.VIS_HEAT:
        show = po.heat_map(cur, state)
    elif vis == lib.VIS_GRAY:
        show = po.gray_weighted(cur)
    elif vis == lib.VIS_BINARIZE:
        gw = po.gray_weighted(cur)
        show = po.binarize(gw, po.two_max_threshold(po.histogram(gw)))
    c, xs, df, new_state = po.diff_pack(cur, state)
    if vis == lib.VIS_RED:
        show = po.red_overlap(np.zeros_like(cur), xs)
    elif vis == lib.VIS_RED_OVERLAP:
        show = po.red_overlap(state, xs)
    return c, xs, df, new_state, show


@pytest.mark.parametrize("vis", [0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("noise_filter", [False, True])
def test_exec_core_all_visualizers(po, vis, noise_filter):
    w, h, T = 96, 54, 4
    base, frames = synth.webcam_stream(T, w, h, seed=30 + vis)
    k = po.gaussian_kernel(3, 1.5)
    n = 3 * w * h
    with CUDACore(w, h, k=k, sample_mat_data=base, visualizer=vis, noise_filter=noise_filter) as core:
        h_frame, n_frame, o_frame, h_xs = CUDACore.alloc_arrays(h, w)
        state = base
        for t in range(T):
            h_frame.array[:n] = frames[t]
            pos = core.exec_core(h_frame.array, n_frame.array, "", h_xs.array)
            c, xs, df, state, show = oracle_exec(po, frames[t], state, vis, k, noise_filter, w, h)
            assert pos == c
            assert np.array_equal(h_xs.array[:pos], xs)
            assert np.array_equal(h_frame.array[:pos], df)
            if show is not None:
                assert np.array_equal(n_frame.array[:n], show)
            assert np.array_equal(core.get_state(), state)
        for a in (h_frame, n_frame, o_frame, h_xs):
            a.free()


@pytest.mark.parametrize("vis", [0, 2, 5])
def test_exec_core_pageable_buffers(po, vis):
    """Callers that did not use alloc_arrays: the copies back are sized on the host as in the reference
    (kernels.cu:507-524) instead of stored through mapped pointers; same results."""
    w, h, T = 96, 54, 3
    base, frames = synth.webcam_stream(T, w, h, seed=80 + vis)
    n = 3 * w * h
    with CUDACore(w, h, sample_mat_data=base, visualizer=vis) as core:
        frame, show, h_xs = np.zeros(n + 32, np.uint8), np.zeros(n + 32, np.uint8), np.zeros(n + 8, np.int32)
        state = base
        for t in range(T):
            frame[:n] = frames[t]
            pos = core.exec_core(frame, show, "", h_xs)
            c, xs, df, state, want_show = oracle_exec(po, frames[t], state, vis, None, False, w, h)
            assert pos == c and np.array_equal(h_xs[:pos], xs) and np.array_equal(frame[:pos], df)
            if want_show is not None:
                assert np.array_equal(show[:n], want_show)
        assert np.array_equal(core.get_state(), state)


def test_exec_core_text_overlay(po):
    """kernel2_char (kernels.cu:351-375): glyph rows are blitted into the frame before the diff."""
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
    with CUDACore(w, h, sample_mat_data=base, chars_px=atlas, chars_sz=(gh, gw), charset=charset) as core:
        h_frame, n_frame, o_frame, h_xs = CUDACore.alloc_arrays(h, w)
        h_frame.array[:3 * w * h] = frames[0]
        pos = core.exec_core(h_frame.array, None, text, h_xs.array)
        c, xs, df, st = po.diff_pack(exp, base)
        assert pos == c and np.array_equal(h_xs.array[:pos], xs) and np.array_equal(h_frame.array[:pos], df)
        assert np.array_equal(core.get_state(), st)


def test_exec_core_1080p(po):
    w, h = 1920, 1080
    base, frames = synth.webcam_stream(2, w, h, device=DEV)
    base, frames = base.cpu().numpy(), frames.cpu().numpy()
    n = 3 * w * h
    with CUDACore(w, h, sample_mat_data=base) as core:
        h_frame, n_frame, o_frame, h_xs = CUDACore.alloc_arrays(h, w)
        state = base
        for t in range(2):
            h_frame.array[:n] = frames[t]
            pos = core.exec_core(h_frame.array, None, "", h_xs.array)
            c, xs, df, state = po.diff_pack(frames[t], state)
            assert pos == c and np.array_equal(h_xs.array[:pos], xs) and np.array_equal(h_frame.array[:pos], df)


# ---- batched filters (one launch per kernel for T frames) ------------------------------------------

@pytest.mark.parametrize("w,h,T", [(64, 48, 5), (37, 11, 3), (640, 360, 4)])
def test_filter_batch_matches_per_frame_oracle(po, w, h, T):
    n = 3 * w * h
    base, frames = synth.webcam_stream(T + 1, w, h, seed=50)
    cur, prev = frames[1:], frames[:-1]
    k = po.gaussian_kernel(3, 1.5)
    stride = n + (16 - n % 16) % 16 + 16          # padded, 16-byte aligned stride
    def pad(a):
        buf = np.zeros((a.shape[0], stride), np.uint8)
        buf[:, :n] = a
        return to_dev(buf)
    d_cur, d_prev = pad(cur), pad(prev)
    with CUDACore(w, h, k=k, max_batch=T) as core:
        def run(op, second=None):
            d_o = torch.full((T, stride), 0x5A, dtype=torch.uint8, device=DEV)
            core.filter_batch(op, d_cur, d_o, T, d_in2=second, stride=stride)
            core.synchronize()
            return d_o.cpu().numpy()[:, :n]
        exp = {
            lib.OP_GRAY_AVG: [po.gray_avg(f) for f in cur],
            lib.OP_GRAY_WEIGHTED: [po.gray_weighted(f) for f in cur],
            lib.OP_GRAY_AVG_BINARIZE: [po.server_cpu_branch(f)[0] for f in cur],
            lib.OP_GRAY_WEIGHTED_BINARIZE: [po.binarize(po.gray_weighted(f), po.two_max_threshold(po.histogram(po.gray_weighted(f)))) for f in cur],
            lib.OP_CONV3X3: [po.conv3x3(f, w, h, k) for f in cur],
        }
        for op, e in exp.items():
            got = run(op)
            for t in range(T):
                assert np.array_equal(got[t], e[t]), (op, t)
        got = run(lib.OP_HEAT_MAP, d_prev)
        for t in range(T):
            assert np.array_equal(got[t], po.heat_map(cur[t], prev[t]))
        got = run(lib.OP_RED_DENSE, d_prev)
        for t in range(T):
            assert np.array_equal(got[t], po.red_dense(cur[t], prev[t]))
        # gray3 -> binarize (unfused form) on the batch
        d_gray = torch.zeros((T, stride), dtype=torch.uint8, device=DEV)
        core.filter_batch(lib.OP_GRAY_AVG, d_cur, d_gray, T, stride=stride)
        d_o = torch.zeros((T, stride), dtype=torch.uint8, device=DEV)
        core.filter_batch(lib.OP_BINARIZE, d_gray, d_o, T, stride=stride)
        core.synchronize()
        for t in range(T):
            assert np.array_equal(d_o[t, :n].cpu().numpy(), po.server_cpu_branch(cur[t])[0])


def test_fused_binarize_vs_reference_fixture():
    """config 3 fused chain (avg form) == the reference's own server.cpp CPU branch outputs."""
    g = golden("ref_server_cpu_64x48.npz")
    w, h = int(g["width"]), int(g["height"])
    T = g["frames"].shape[0]
    with CUDACore(w, h, max_batch=T) as core:
        d_o = torch.zeros_like(to_dev(g["frames"]))
        core.filter_batch(lib.OP_GRAY_AVG_BINARIZE, to_dev(g["frames"]), d_o, T)
        core.synchronize()
        assert np.array_equal(d_o.cpu().numpy(), g["out"])


def test_config3_and_config4_at_1080p(po):
    """BASELINE configs 3 and 4 at full size through exec_core: weighted gray + binarize visualiser,
    and 3x3 noise filter + red motion map, each followed by the diff/threshold/pack."""
    w, h = 1920, 1080
    n = 3 * w * h
    base, frames = synth.webcam_stream(2, w, h, seed=60, device=DEV)
    base, frames = base.cpu().numpy(), frames.cpu().numpy()
    k = po.gaussian_kernel(3, 1.5)
    for vis, nf in ((lib.VIS_BINARIZE, False), (lib.VIS_RED, True)):
        with CUDACore(w, h, k=k, sample_mat_data=base, visualizer=vis, noise_filter=nf) as core:
            h_frame, n_frame, o_frame, h_xs = CUDACore.alloc_arrays(h, w)
            state = base
            for t in range(2):
                h_frame.array[:n] = frames[t]
                pos = core.exec_core(h_frame.array, n_frame.array, "", h_xs.array)
                c, xs, df, state, show = oracle_exec(po, frames[t], state, vis, k, nf, w, h)
                assert pos == c and np.array_equal(h_xs.array[:pos], xs) and np.array_equal(h_frame.array[:pos], df)
                assert np.array_equal(n_frame.array[:n], show)
            for a in (h_frame, n_frame, o_frame, h_xs):
                a.free()


# ---- pins that do not pass through the oracle of the same box ------------------------------------------------------------
def test_fused_binarize_vs_reference_run_at_1080p():
    """The fused avg-gray + binarize chain at BASELINE size against a recorded run of the reference's own server.cpp
    CPU branch (tests/golden/ref_server_cpu_1080p.npz: SHA-256 of the outputs, white bytes; interior threshold and
    both clamps)."""
    import hashlib
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_ref_1080p import H, SEED, T, W
    g = golden("ref_server_cpu_1080p.npz")
    fr = torch.stack([synth.webcam_frame(t, W, H, seed=SEED, device=DEV) for t in range(T)])
    fr[1] = fr[1] // 6
    fr[2] = 200 + fr[2] // 5
    with CUDACore(W, H, max_batch=T) as core:
        d_o = torch.zeros_like(fr)
        core.filter_batch(lib.OP_GRAY_AVG_BINARIZE, fr, d_o, T)
        core.synchronize()
        out = d_o.cpu().numpy()
    for t in range(T):
        assert hashlib.sha256(out[t].tobytes()).digest() == bytes(g["sha256"][t]), f"frame {t}"
        assert int((out[t] == 255).sum()) == int(g["white"][t])


def test_heat_map_against_the_committed_lut_every_distance():
    """Every d = |dB| + |dG| + |dR| in 0..765 on the device against the COMMITTED 766 x 3 table
    (tests/golden/oracle_heat_lut.npz), not against an oracle evaluated with this box's libm: the table the library
    builds at mi355_create with the host's sin() is what is being checked."""
    lut = golden("oracle_heat_lut.npz")["lut"].reshape(766, 3)
    ds = np.arange(766)
    # a pixel pair for every d: differences spread over the three channels, both signs
    db, dg = np.minimum(ds, 255), np.minimum(np.maximum(ds - 255, 0), 255)
    dr = ds - db - dg
    npix = 768
    cur = np.zeros((npix, 3), np.uint8)
    prev = np.zeros((npix, 3), np.uint8)
    cur[:766, 0], prev[:766, 1], cur[:766, 2] = db, dg, dr      # |cur - prev| per channel: dB, dG, dR
    with CUDACore(npix, 1) as core:
        d_o = dev_out(3 * npix)
        core.heat_map(to_dev(cur.reshape(-1)), to_dev(prev.reshape(-1)), d_o)
        core.synchronize()
        out = d_o.cpu().numpy().reshape(-1, 3)
    assert np.array_equal(out[:766], lut), np.nonzero((out[:766] != lut).any(axis=1))[0][:10]


def test_batch_filters_against_the_committed_64x48_fixture():
    """Every filter of mi355_filter_batch on the committed inputs of tests/golden/oracle_filters_64x48.npz against the
    committed outputs (made in the build container), through the batch entry point."""
    g = golden("oracle_filters_64x48.npz")
    w, h = int(g["width"]), int(g["height"])
    T = 3
    img = np.repeat(g["img"][None, :], T, axis=0)
    prv = np.repeat(g["prev"][None, :], T, axis=0)
    with CUDACore(w, h, k=g["k"], max_batch=T) as core:
        for op, two, key in ((lib.OP_GRAY_AVG, False, "gray_avg"), (lib.OP_GRAY_WEIGHTED, False, "gray_weighted"),
                             (lib.OP_GRAY_WEIGHTED_BINARIZE, False, "binarized"), (lib.OP_HEAT_MAP, True, "heat"),
                             (lib.OP_RED_DENSE, True, "red"), (lib.OP_CONV3X3, False, "conv")):
            d_o = torch.zeros_like(to_dev(img))
            if two:
                core.filter_batch(op, to_dev(img), d_o, T, d_in2=to_dev(prv))
            else:
                core.filter_batch(op, to_dev(img), d_o, T)
            core.synchronize()
            out = d_o.cpu().numpy()
            for t in range(T):
                assert np.array_equal(out[t], g[key]), (key, t)
