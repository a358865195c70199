"""CPU tests of the oracle itself: pinned against the reference's own outputs and identities
 and against the committed golden vectors."""
import os
import sys

import numpy as np
import pytest

from conftest import golden
from cudavideostream_amd import synth

REF_PRESENT = os.path.exists("/root/reference/server/src/server.cpp")


# ---- pinned by the reference itself -------------------------------------------------------------

def test_cpu_branch_matches_reference_fixture(po):
    """gray-avg -> histogram -> two-max -> binarize == the reference's server.cpp CPU branch
    (outputs recorded from oracle/_ref/server_cpu, i.e. server/src/server.cpp:96-135 itself)."""
    g = golden("ref_server_cpu_64x48.npz")
    thrs = []
    for t in range(g["frames"].shape[0]):
        out, thr = po.server_cpu_branch(g["frames"][t])
        thrs.append(thr)
        assert np.array_equal(out, g["out"][t]), f"frame {t}"
    assert 50 in thrs and 200 in thrs  # both clamps of server.cpp:122-127 are exercised


@pytest.mark.skipif(not REF_PRESENT, reason="reference tree not present (GPU box)")
def test_cpu_branch_matches_reference_live(po, tmp_path):
    """Same check against a fresh run of the reference binary on new random frames."""
    po.build()
    rng = np.random.default_rng(7)
    w, h = 40, 30
    base = rng.integers(0, 256, 3 * w * h, dtype=np.uint8)
    frames = rng.integers(0, 256, (8, 3 * w * h), dtype=np.uint8)
    frames[3] //= 5
    frames[5] = 255 - frames[5] // 7
    out = po.run_ref_server_cpu(base, frames, w, h, str(tmp_path))
    for t in range(8):
        mine, _ = po.server_cpu_branch(frames[t])
        assert np.array_equal(mine, out[t])


def test_f1f2_changed_byte_count_is_the_reports(po):
    """The reference-held number for the hot path: f1.jpg vs f2.jpg differ in 369350 bytes at threshold 20
    (REPORT/report.tex:2594; counted by tests/noise_filter_benchmark/v2.cu:106-114,215).  Pins the oracle's
    strict comparison: `>=` would give 413893."""
    g = golden("ref_f1f2_1080p.npz")
    f1, f2 = g["f1"].reshape(-1), g["f2"].reshape(-1)
    want = int(g["count_gt20"])
    assert want == 369350
    for cur, prev in ((f2, f1), (f1, f2)):      # abs() is symmetric: both orders count the same bytes
        c, xs, df, st = po.diff_pack(cur, prev)
        assert c == want
        assert c != int(g["count_ge20"])
        # the entries are exactly the bytes getCountDifference counts, ascending (test.cu:563-573)
        flagged = np.flatnonzero(np.abs(cur.astype(np.int32) - prev.astype(np.int32)) > 20)
        assert np.array_equal(xs, flagged)
        assert np.array_equal(df, (cur[flagged].astype(np.int32) - prev[flagged]).astype(np.uint8))
        assert np.array_equal(st[flagged], cur[flagged])
        keep = np.ones(cur.size, bool); keep[flagged] = False
        assert np.array_equal(st[keep], prev[keep])
    # one threshold step either side moves the count: the comparison constant is pinned too
    assert po.diff_pack(f2, f1, thr=19)[0] == int(g["count_ge20"])
    assert po.diff_pack(f2, f1, thr=21)[0] < want
    # the row-band form used as the all-cores baseline counts the same
    assert po.diff_pack_mt(f2, f1, nthreads=8)[0] == want


def test_f1f2_red_map_marks_the_pixels_owning_those_bytes(po):
    """tests/heat_map_red_benchmark/cpu.cu:38-55 on the report's frame pair: a pixel is red iff it owns at
    least one of the 369350 counted bytes."""
    g = golden("ref_f1f2_1080p.npz")
    f1, f2 = g["f1"].reshape(-1), g["f2"].reshape(-1)
    _, xs, _, _ = po.diff_pack(f2, f1)
    red = po.red_dense(f2, f1).reshape(-1, 3)
    owners = np.zeros(f1.size // 3, bool)
    owners[xs // 3] = True
    assert np.array_equal(red[:, 2] == 255, owners)
    assert not red[:, :2].any() and set(np.unique(red[:, 2])) <= {0, 255}
    # the overlap form paints the same pixels onto the previous frame (kernels.cu:273-281)
    over = po.red_overlap(f1, xs).reshape(-1, 3)
    assert np.array_equal(over[owners, 2], np.full(int(owners.sum()), 255, np.uint8))
    assert np.array_equal(over[~owners], f1.reshape(-1, 3)[~owners])


def test_f1f2_filter_table(po):
    """REPORT/report.tex:2601-2611: share of bytes still changed after both frames went through the K x K
    filter -- mean K=3 3.37 %, Gaussian K=3 sigma=1 3.58 %.  The oracle's float-accumulator filter
    (kernels.cu:97-136 semantics) leaves 3.35 / 3.56 %, the reference's int-accumulator CPU statement
    (cpu.cu:72-98) 3.39 / 3.59 %: the report's figures lie between the two, all agree to two digits."""
    g = golden("ref_f1f2_1080p.npz")
    f1, f2 = g["f1"], g["f2"]
    n = f1.size
    for name, k in (("mean3", np.full(9, np.float32(1.0 / 9), np.float32)), ("gauss3_s1", po.gaussian_kernel(3, 1.0))):
        a, b = po.conv3x3(f1, 1920, 1080, k), po.conv3x3(f2, 1920, 1080, k)
        resid = po.diff_pack(b, a)[0]
        assert resid == int(g["resid_" + name])
        report = float(g["report_pct_" + name])
        assert abs(100.0 * resid / n - report) < 0.05
        ia = po.conv3x3_intacc(f1, 1920, 1080, k)
        d = a.astype(np.int32) - ia
        assert d.min() >= 0 and d.max() <= 8        # nine truncations of < 1 each (cpu.cu:82)
        ib = po.conv3x3_intacc(f2, 1920, 1080, k)
        resid_i = int((np.abs(ia - ib) > 20).sum())
        assert resid_i == int(g["resid_intacc_" + name])
        assert abs(100.0 * resid_i / n - report) < 0.05
        assert 100.0 * resid / n <= report <= 100.0 * resid_i / n


def test_f1f2_whole_filter_table(po):
    """All nine rows of REPORT/report.tex:2601-2611 (mean K = 3, 5, 7, 9; Gaussian K = 3..7 with its sigma): bytes of
    the f1/f2 pair still changed after both frames went through the K x K filter of the study
    (tests/noise_filter_benchmark/v2.cu:36-80 restated as ora_conv_kxk, kernels from v2.cu:116-124 / :139-160).
    Every row agrees with the report to 0.05 percentage points; even K (4, 6) fixes the tap window -K/2 .. K-1-K/2."""
    g = golden("ref_f1f2_1080p.npz")
    f1, f2 = g["f1"], g["f2"]
    n = f1.size
    assert list(g["table_report_pct"]) == [3.37, 2.31, 1.66, 1.24, 3.58, 2.87, 2.37, 1.98, 1.66]
    for kind, K, sigma, pct, want in zip(g["table_kind"], g["table_K"], g["table_sigma"], g["table_report_pct"],
                                         g["table_resid"]):
        k = po.mean_kernel(int(K)) if kind == 0 else po.gaussian_kernel(int(K), float(sigma))
        assert abs(float(k.sum()) - 1.0) < 1e-5
        a, b = po.conv_kxk(f1, 1920, 1080, k), po.conv_kxk(f2, 1920, 1080, k)
        resid = po.diff_pack(b, a)[0]
        assert resid == int(want), (int(K), float(sigma))
        assert abs(100.0 * resid / n - float(pct)) < 0.05
    # K = 3 is the server's filter: the general statement and the 3 x 3 one are the same function
    k3 = po.gaussian_kernel(3, 1.5)
    assert np.array_equal(po.conv_kxk(f1, 1920, 1080, k3), po.conv3x3(f1, 1920, 1080, k3))


def test_two_max_dead_branch(po):
    """server.cpp:116 `else if` can never fire (sec_max == max after every record)."""
    rng = np.random.default_rng(3)
    for _ in range(2000):
        hist = rng.integers(0, 50, 256).astype(np.int32)
        idx, sec = -1, -1
        mx = -1
        for i in range(256):  # prefix-maximum records
            if hist[i] >= mx:
                sec, idx, mx = idx, i, hist[i]
        thr = min(200, max(50, int((idx + sec) / 2)))
        assert po.two_max_threshold(hist) == thr


def test_integer_diff_identity(po):
    """tests/algorithms_benchmarks.cu:12-22: frame1 - frame2 == diff (with its own index quirk)."""
    L = po.lib()
    h, w = 1920, 1080  # as called at algorithms_benchmarks.cu:106
    n = h * w * 3
    a = np.empty(n, np.int32); b = np.empty(n, np.int32); d = np.empty(n, np.int32)
    L.ora_generate_image(a, h, w, 1)
    L.ora_generate_image(b, h, w, 2)
    assert a.min() >= 0 and a.max() <= 254       # rand() % 255
    L.ora_int_diff(a, b, d, n)
    assert L.ora_check_difference(a, b, d, h, w) == 0
    d[5] += 1
    assert L.ora_check_difference(a, b, d, h, w) == -1


def test_count_identity(po):
    """tests/test_cuda/pixel_diff.cu:47-59: #(cur != prev) == #(diff != 0) (threshold 0)."""
    rng = np.random.default_rng(11)
    cur = rng.integers(0, 4, 50000, dtype=np.uint8)
    prev = rng.integers(0, 4, 50000, dtype=np.uint8)
    c, xs, df, _ = po.diff_pack(cur, prev, thr=0)
    assert c == int((cur != prev).sum()) == int((df != 0).sum())


def test_client_reconstruction_identity(po):
    """client/opencv.cpp:64-66: prev[xs] += diff rebuilds exactly the server's state."""
    base, frames = synth.webcam_stream(4, 48, 32, seed=2)
    client = base.copy()
    state = base.copy()
    for t in range(4):
        c, xs, df, state = po.diff_pack(frames[t], state)
        client = po.client_apply(client, xs, df)
        assert np.array_equal(client, state)
        # everything the client does not know is within the threshold of the true frame
        assert np.abs(client.astype(int) - frames[t].astype(int)).max() <= 20


def test_inplace_form_matches(po):
    """test.cu:567 writes the diff bytes over the head of the frame buffer."""
    rng = np.random.default_rng(5)
    cur = rng.integers(0, 256, 9000, dtype=np.uint8)
    prev = rng.integers(0, 256, 9000, dtype=np.uint8)
    c, xs, df, st = po.diff_pack(cur, prev)
    buf, st2 = cur.copy(), prev.copy()
    xs2 = np.empty(9000, np.int32)
    c2 = po.lib().ora_diff_pack_inplace(buf, st2, 9000, 20, xs2)
    assert c2 == c and np.array_equal(xs2[:c], xs) and np.array_equal(buf[:c], df)
    assert np.array_equal(st2, st)


def test_mt_matches_single_thread(po):
    rng = np.random.default_rng(6)
    cur = rng.integers(0, 256, 100003, dtype=np.uint8)
    prev = rng.integers(0, 256, 100003, dtype=np.uint8)
    ref = po.diff_pack(cur, prev)
    for nt in (1, 2, 7, 8, 64):
        got = po.diff_pack_mt(cur, prev, nthreads=nt)
        assert got[0] == ref[0]
        for a, b in zip(got[1:], ref[1:]):
            assert np.array_equal(a, b)


def test_stream_mt_matches_single_thread(po):
    """bench.py's all-cores baseline (ora_diff_stream_mt: a row band per thread for all frames of the batch) produces
    the single-threaded stream bit for bit: offsets, entries, final state -- also with more threads than bytes, empty
    batches, a capacity that is too small."""
    rng = np.random.default_rng(16)
    n, T = 30011, 7
    base = rng.integers(0, 256, n, dtype=np.uint8)
    frames = np.clip(base.astype(int)[None, :] + rng.integers(-30, 31, (T, n)), 0, 255).astype(np.uint8)
    ref = po.diff_stream(frames, base)
    for nt in (1, 2, 5, 8, 64, 1024):
        got = po.diff_stream_mt(frames, base, nthreads=nt)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b), nt
    tiny = po.diff_stream_mt(frames[:, :3], base[:3], nthreads=8)
    for a, b in zip(tiny, po.diff_stream(frames[:, :3], base[:3])):
        assert np.array_equal(a, b)
    off, xs, df, st = po.diff_stream_mt(frames[:0], base, nthreads=4)
    assert off.tolist() == [0] and xs.size == 0 and np.array_equal(st, base)
    L = po.lib()
    o = np.zeros(T + 1, np.uint32)
    rc = L.ora_diff_stream_mt(frames.reshape(-1), T, base.copy(), n, 20, o, np.empty(8, np.int32), np.empty(8, np.uint8), 8, 4)
    assert rc == -1


# ---- semantics at the threshold -------------------------------------------------------------------

def test_edge_strip_threshold_semantics(po):
    cur, prev = synth.edge_strip()
    c, xs, df, st = po.diff_pack(cur, prev)
    d = cur.astype(int) - prev.astype(int)
    flagged = np.abs(d) > 20  # strict >, kernels.cu:312
    assert c == int(flagged.sum())
    assert np.array_equal(xs, np.nonzero(flagged)[0].astype(np.int32))
    assert np.array_equal(df, (d[flagged] & 0xFF).astype(np.uint8))
    assert np.array_equal(st, np.where(flagged, cur, prev))
    g = golden("oracle_diff_edge_strip.npz")
    assert c == int(g["count"]) and np.array_equal(xs, g["xs"]) and np.array_equal(df, g["diff"])


def test_static_flip_empty(po):
    cur, prev = synth.static_pair(5000)
    assert po.diff_pack(cur, prev)[0] == 0
    cur, prev = synth.flip_pair(5000)
    c, xs, _, st = po.diff_pack(cur, prev)
    assert c == 5000 and np.array_equal(xs, np.arange(5000, dtype=np.int32)) and np.array_equal(st, cur)
    c, xs, df, st = po.diff_pack(np.empty(0, np.uint8), np.empty(0, np.uint8))
    assert c == 0 and xs.size == 0 and st.size == 0


# ---- golden vectors ---------------------------------------------------------------------------------

def test_golden_stream(po):
    g = golden("oracle_diff_stream_64x48.npz")
    offsets, xs, df, st = po.diff_stream(g["frames"], g["base"])
    assert np.array_equal(offsets, g["offsets"]) and np.array_equal(xs, g["xs"])
    assert np.array_equal(df, g["diff"]) and np.array_equal(st, g["state"])


def test_golden_ragged(po):
    g = golden("oracle_diff_ragged_37x11.npz")
    c, xs, df, st = po.diff_pack(g["cur"], g["prev"])
    assert c == int(g["count"]) and np.array_equal(xs, g["xs"]) and np.array_equal(df, g["diff"])
    assert np.array_equal(st, g["state"])


def test_golden_filters(po):
    g = golden("oracle_filters_64x48.npz")
    w, h, img, prv = int(g["width"]), int(g["height"]), g["img"], g["prev"]
    assert np.array_equal(po.gray_avg(img), g["gray_avg"])
    gw = po.gray_weighted(img)
    assert np.array_equal(gw, g["gray_weighted"])
    assert np.array_equal(po.histogram(gw), g["hist"])
    assert po.two_max_threshold(g["hist"]) == int(g["thr"])
    assert np.array_equal(po.binarize(gw, int(g["thr"])), g["binarized"])
    assert np.array_equal(po.heat_map(img, prv), g["heat"])
    assert np.array_equal(po.red_dense(img, prv), g["red"])
    assert np.array_equal(po.conv3x3(img, w, h, g["k"]), g["conv"])


def test_heat_lut_spot_values(po):
    """SURVEY.md section 8a-7 spot values (normaliser 510: red falls off again above d=510)."""
    lut = po.heat_lut()
    assert np.array_equal(lut, golden("oracle_heat_lut.npz")["lut"])
    bgr = {0: (255, 0, 0), 1: (254, 1, 0), 255: (0, 255, 0), 510: (0, 0, 255), 600: (0, 0, 216),
           765: (0, 0, 0)}
    for d, v in bgr.items():
        assert tuple(lut[d]) == v


def test_gaussian_kernel(po):
    k = po.gaussian_kernel(3, 1.5)  # server.cpp:43 sigma = K*K/6.0
    assert np.array_equal(k, golden("oracle_gaussian_k3.npz")["k"])
    assert abs(float(k.sum()) - 1.0) < 1e-6 and k[4] == k.max()
    assert k[0] == k[2] == k[6] == k[8] and k[1] == k[3] == k[5] == k[7]


def test_gray_weighted_sample_and_numpy(po):
    g = golden("oracle_gray_weighted_sample.npz")
    bgr = g["bgr"]
    got = po.gray_weighted(bgr.reshape(-1)).reshape(-1, 3)
    assert np.array_equal(got[:, 0], g["gray"]) and np.array_equal(got[:, 1], got[:, 2])
    # same expression in numpy float64, left to right (tests/grayscale-weighted/cpu.cu:40)
    v = 0.114 * bgr[:, 0].astype(np.float64) + 0.587 * bgr[:, 1] + 0.299 * bgr[:, 2]
    assert np.array_equal(v.astype(np.uint8), g["gray"])


def test_conv_known_small_image(po):
    """3x3x3 hand image of tests/noise_filter_benchmark/v1.cu:122 (expected values are not recorded
    upstream): the float path must agree with exact rational arithmetic after truncation wherever the
    exact value is not within 1e-4 of an integer; borders see the zero halo."""
    img = np.array([1, 0, 120, 2, 0, 139, 3, 0, 90, 4, 0, 99, 5, 0, 126, 6, 0, 106, 7, 0, 46, 8, 0, 75,
                    9, 0, 88], np.uint8)
    k = po.gaussian_kernel(3, 1.5)
    out = po.conv3x3(img, 3, 3, k).reshape(3, 3, 3)
    src = img.reshape(3, 3, 3).astype(np.float64)
    pad = np.zeros((5, 5, 3)); pad[1:4, 1:4] = src
    k64 = k.astype(np.float64).reshape(3, 3)
    for y in range(3):
        for x in range(3):
            for c in range(3):
                exact = float((pad[y:y + 3, x:x + 3, c] * k64).sum())
                if abs(exact - round(exact)) > 1e-4:
                    assert out[y, x, c] == int(exact)
    assert (out[:, :, 1] == 0).all()


def test_red_overlap(po):
    img = np.zeros(30, np.uint8)
    out = po.red_overlap(img, np.array([0, 4, 8, 29], np.int32))
    assert np.array_equal(np.nonzero(out)[0], [2, 5, 8, 29])


# ---- synthetic inputs ---------------------------------------------------------------------------------

def test_synth_properties(po):
    a = synth.refrand_frame(200000, 1)
    b = synth.refrand_frame(200000, 2)
    assert a.max() <= 254 and 0.83 < po.diff_pack(a, b)[0] / a.size < 0.86
    base, frames = synth.webcam_stream(3, 192, 108)
    st = base
    fr = []
    for t in range(3):
        c, _, _, st = po.diff_pack(frames[t], st)
        fr.append(c / base.size)
    assert 0.01 < fr[1] < 0.06 and 0.01 < fr[2] < 0.06  # webcam-like sparsity
    torch = pytest.importorskip("torch")
    tb, tf = synth.webcam_stream(2, 192, 108, device="cpu")
    assert np.array_equal(tb.numpy(), base) and np.array_equal(tf.numpy(), frames[:2])


def test_wire_format_and_client_round_trip(po):
    """server/src/threads.cpp:227-229 (sender) and client/opencv.cpp:50-66 (receiver) restated: layout of a
    hand-made stream, and client == server state after a seeded stream."""
    off = np.array([0, 2, 2, 3], np.uint32)
    xs = np.array([1, 7, 4], np.int32)
    df = np.array([200, 3, 255], np.uint8)
    wire = po.wire_pack(off, xs, df)
    want = (b"\x02\x00\x00\x00" + b"\x01\x00\x00\x00\x07\x00\x00\x00" + b"\xc8\x03" +
            b"\x00\x00\x00\x00" +
            b"\x01\x00\x00\x00" + b"\x04\x00\x00\x00" + b"\xff")
    assert wire.tobytes() == want
    base = np.arange(9, dtype=np.uint8) + 100
    shown, counts = po.wire_client(base, wire, 3)
    assert counts.tolist() == [2, 0, 1]
    assert shown[0].tolist() == [100, 45, 102, 103, 104, 105, 106, 110, 108]      # 101+200 wraps (opencv.cpp:65)
    assert shown[1].tolist() == shown[0].tolist()
    assert shown[2].tolist() == [100, 45, 102, 103, 103, 105, 106, 110, 108]
    from cudavideostream_amd import synth
    b, fr = synth.webcam_stream(6, 48, 20, seed=3)
    o, x, d, st = po.diff_stream(fr, b)
    shown, counts = po.wire_client(b, po.wire_pack(o, x, d), 6)
    assert np.array_equal(shown[-1], st)
    assert np.array_equal(counts, np.diff(o.astype(np.int64)))
    assert (np.abs(shown.astype(np.int16) - fr.astype(np.int16)) <= 20).all()


def test_median5x5_restatement(po):
    """tests/noise_filter_benchmark/v3.cu:32-90: against numpy's sort on a zero-padded copy, and two values a
    reader can check by hand."""
    rng = np.random.default_rng(2)
    w, h = 11, 8
    img = rng.integers(0, 256, 3 * w * h, dtype=np.uint8)
    a = img.reshape(h, w, 3)
    pad = np.zeros((h + 4, w + 4, 3), np.uint8)
    pad[2:-2, 2:-2] = a
    ref = np.empty_like(a)
    for y in range(h):
        for x in range(w):
            for c in range(3):
                ref[y, x, c] = np.sort(pad[y:y + 5, x:x + 5, c].reshape(-1))[12]
    assert np.array_equal(po.median5x5(img, w, h).reshape(h, w, 3), ref)
    const = po.median5x5(np.full(3 * 9 * 9, 77, np.uint8), 9, 9).reshape(9, 9, 3)
    assert const[4, 4, 0] == 77          # interior: 25 equal values
    assert const[0, 0, 0] == 0           # corner: 9 image values, 16 zeros -> the 13th smallest is a zero
    assert const[0, 4, 1] == 77          # top edge: 15 image values, 10 zeros


def test_cpu_branch_matches_reference_run_at_1080p(po):
    """The restated CPU branch against a recorded run of the reference's own server.cpp at BASELINE size (digests;
    tests/golden/make_ref_1080p.py regenerates the seeded inputs)."""
    import hashlib
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_ref_1080p import frames_1080p
    g = golden("ref_server_cpu_1080p.npz")
    _, fr = frames_1080p()
    assert fr.shape[0] == int(g["nframes"])
    for t in range(fr.shape[0]):
        out, thr = po.server_cpu_branch(fr[t])[:2]
        assert hashlib.sha256(np.ascontiguousarray(out).tobytes()).digest() == bytes(g["sha256"][t]), f"frame {t}"
        assert int((out == 255).sum()) == int(g["white"][t])
        if int(g["thr"][t]) >= 0:
            assert int(thr) == int(g["thr"][t])


def test_integer_gray_equals_the_double_expression():
    """The integer form of the weighted gray the kernels use (csrc/filters.hip, gray_of): floor(K / 1000) by
    multiply-shift, K = 114 B + 587 G + 299 R, minus one for the triples of the exception table -- against the double
    expression of tests/grayscale-weighted/cpu.cu:40 on all 2^24 triples.  The table is rebuilt here the way
    fill_gray_exceptions builds it: for every (B, G) the one R in 0..255 that makes K a multiple of 1000, kept when
    the double expression truncates below K / 1000."""
    x = np.arange(1 << 24, dtype=np.int64)
    B, G, R = x & 255, (x >> 8) & 255, x >> 16
    ref = ((0.114 * B.astype(np.float64) + 0.587 * G) + 0.299 * R).astype(np.uint8)
    K = 114 * B + 587 * G + 299 * R
    q = ((K >> 3) * 33555) >> 22
    assert np.array_equal(q, K // 1000)                                    # the multiply-shift is an exact division
    # the table
    bb, gg = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    rr = (1000 - (114 * bb + 587 * gg) % 1000) % 1000 * 699 % 1000
    ok = rr <= 255
    kk = 114 * bb + 587 * gg + 299 * np.where(ok, rr, 0)
    assert np.all(kk[ok] % 1000 == 0)
    dbl = ((0.114 * bb.astype(np.float64) + 0.587 * gg) + 0.299 * np.where(ok, rr, 0)).astype(np.int64)
    table = np.where(ok & (dbl != kk // 1000), rr, 0xFFFF)
    assert int((table != 0xFFFF).sum()) == 1957
    # the kernel's rule
    exc = (K == q * 1000) & (table[B, G] == R)
    got = (q - exc).astype(np.uint8)
    assert np.array_equal(got, ref)
    # every mismatch of the plain floor is in the table, and only multiples of 1000 are
    assert int((q.astype(np.uint8) != ref).sum()) == 1957 == int(exc.sum())
