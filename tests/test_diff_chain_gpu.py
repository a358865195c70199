"""GPU parity of the opt-in one-pass pair form (csrc/diff_chain.hip, MI355_FLAG_CHAIN: chained scan with
decoupled look-back) against the CPU oracle (tests/cuda_streaming/test.cu:560-576 restated, stateless pairs as
in tests/algorithms_benchmarks.cu), bit-exact, through the C-ABI.  The experiment is not the product path; these
tests keep it honest."""
import numpy as np
import pytest

from conftest import golden
from cudavideostream_amd import synth

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
from gpu_util import CUDACore, oracle_pairs, run_stream  # noqa: E402


@pytest.fixture(autouse=True, scope="module")
def _needs_experiment_build():
    """The two single-pass experiments are only in the library when it was built with `make EXPERIMENTS=1`
    (cudavideostream_amd/csrc/Makefile); the default build refuses the flag."""
    try:
        with CUDACore(16, 16, max_batch=1, chain=True):
            pass
    except RuntimeError as e:
        if "EXPERIMENTS" in str(e):
            pytest.skip("library built without the experiment kernels (make EXPERIMENTS=1)")
        raise


def check_pairs(po, core, cur, prev, thr=20, **kw):
    off, xs, df, _ = run_stream(core, cur, pair_prev=prev, **kw)
    eo, exs, edf = oracle_pairs(po, cur, prev, thr)
    assert np.array_equal(off, eo), (off, eo)
    cap = kw.get("capacity")
    if cap is not None:
        exs, edf = exs[:cap], edf[:cap]
    assert np.array_equal(xs, exs) and np.array_equal(df, edf)
    return off


@pytest.mark.parametrize("w,h,T", [(64, 48, 5), (80, 60, 3), (16, 1, 2), (400, 300, 7), (1024, 40, 9)])
def test_sizes_and_partial_blocks(po, w, h, T):
    """One block or many per frame, partial last block / tile, several frames (the frame chain)."""
    n = 3 * w * h
    rng = np.random.default_rng(w + h + T)
    prev = rng.integers(0, 256, (T, n), dtype=np.uint8)
    cur = np.clip(prev.astype(int) + rng.integers(-35, 36, (T, n)), 0, 255).astype(np.uint8)
    marker = rng.integers(0, 256, n, dtype=np.uint8)
    with CUDACore(w, h, max_batch=T, chain=True) as core:
        core.set_state(marker)
        check_pairs(po, core, cur, prev)
        assert np.array_equal(core.get_state(), marker)      # the pair form leaves the state alone


def test_regimes_and_thresholds(po):
    n = 3 * 256 * 256
    cur, prev = synth.edge_strip(3)
    for thr in (0, 20, 127):
        with CUDACore(256, 256, threshold=thr, max_batch=1, chain=True) as core:
            check_pairs(po, core, cur[None, :], prev[None, :], thr=thr)
    with CUDACore(256, 256, max_batch=4, chain=True) as core:
        a = np.stack([synth.refrand_frame(n, 10 + t) for t in range(4)])       # S0, ~85 % flagged
        b = np.stack([synth.refrand_frame(n, 20 + t) for t in range(4)])
        off = check_pairs(po, core, a, b)
        assert 0.8 < off[1] / n < 0.9
        f, p0 = synth.flip_pair(n)
        assert check_pairs(po, core, f[None, :], p0[None, :])[1] == n           # every byte: arithmetic path
        s, p1 = synth.static_pair(n)
        assert check_pairs(po, core, s[None, :], p1[None, :])[1] == 0
        mixed = np.stack([f, s, a[0], p1])                                     # dense, empty, S0, empty frames
        check_pairs(po, core, mixed, np.stack([p0, p1, b[0], p1]))
        eo, _, _ = oracle_pairs(po, a, b)
        check_pairs(po, core, a, b, capacity=int(eo[2]) + 7)                    # capacity cuts inside frame 2


def test_1080p_and_reference_count(po):
    W, H = 1920, 1080
    g = golden("ref_f1f2_1080p.npz")      # REPORT/report.tex:2594
    f1, f2 = np.ascontiguousarray(g["f1"].reshape(-1)), np.ascontiguousarray(g["f2"].reshape(-1))
    base, frames = synth.webcam_stream(5, W, H, seed=3)
    cur = np.concatenate([np.stack([f2, f1]), frames[1:]])
    prev = np.concatenate([np.stack([f1, f2]), frames[:-1]])
    with CUDACore(W, H, max_batch=cur.shape[0], chain=True) as core:
        off = check_pairs(po, core, cur, prev)
        assert int(off[1]) == 369350 == int(g["count_gt20"]) and int(off[2]) == 2 * 369350
        # consecutive launches: the descriptors of the first are not mistaken for the second's
        check_pairs(po, core, cur[::-1].copy(), prev[::-1].copy())
