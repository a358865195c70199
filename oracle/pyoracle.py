"""ctypes binding of oracle/liboracle.so (the CPU restatement of the reference path).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package (cudavideostream_amd), which has no CPU path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")


def build():
    """Compile liboracle.so (and oracle/_ref when the reference tree is present)."""
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    # ORACLE_LIB: another build of the same file (the ASan/UBSan build of tests/sanitize.sh)
    path = os.environ.get("ORACLE_LIB") or os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    L.ora_diff_pack.restype = C.c_uint32
    L.ora_diff_pack.argtypes = [u8p, u8p, C.c_size_t, C.c_int, i32p, u8p]
    L.ora_diff_pack_inplace.restype = C.c_uint32
    L.ora_diff_pack_inplace.argtypes = [u8p, u8p, C.c_size_t, C.c_int, i32p]
    L.ora_diff_pack_mt.restype = C.c_uint32
    L.ora_diff_pack_mt.argtypes = [u8p, u8p, C.c_size_t, C.c_int, i32p, u8p, C.c_int]
    L.ora_diff_stream.restype = C.c_int
    L.ora_diff_stream.argtypes = [u8p, C.c_int, u8p, C.c_size_t, C.c_int, u32p, i32p, u8p,
                                  C.c_size_t]
    L.ora_diff_stream_mt.restype = C.c_int
    L.ora_diff_stream_mt.argtypes = [u8p, C.c_int, u8p, C.c_size_t, C.c_int, u32p, i32p, u8p, C.c_size_t, C.c_int]
    L.ora_client_apply.restype = None
    L.ora_client_apply.argtypes = [u8p, i32p, u8p, C.c_uint32]
    L.ora_generate_image.restype = None
    L.ora_generate_image.argtypes = [i32p, C.c_int, C.c_int, C.c_uint]
    L.ora_int_diff.restype = None
    L.ora_int_diff.argtypes = [i32p, i32p, i32p, C.c_size_t]
    L.ora_check_difference.restype = C.c_int
    L.ora_check_difference.argtypes = [i32p, i32p, i32p, C.c_int, C.c_int]
    L.ora_gaussian_kernel.restype = None
    L.ora_gaussian_kernel.argtypes = [f32p, C.c_int, C.c_float]
    L.ora_conv3x3.restype = None
    L.ora_conv3x3.argtypes = [u8p, u8p, C.c_int, C.c_int, f32p]
    L.ora_conv_kxk.restype = None
    L.ora_conv_kxk.argtypes = [u8p, u8p, C.c_int, C.c_int, f32p, C.c_int]
    L.ora_mean_kernel.restype = None
    L.ora_mean_kernel.argtypes = [f32p, C.c_int]
    L.ora_median5x5.restype = None
    L.ora_median5x5.argtypes = [u8p, u8p, C.c_int, C.c_int]
    L.ora_conv3x3_intacc.restype = None
    L.ora_conv3x3_intacc.argtypes = [i32p, i32p, C.c_int, C.c_int, f32p]
    L.ora_heat_lut.restype = None
    L.ora_heat_lut.argtypes = [u8p]
    L.ora_heat_map.restype = None
    L.ora_heat_map.argtypes = [u8p, u8p, u8p, C.c_size_t]
    L.ora_red_dense.restype = None
    L.ora_red_dense.argtypes = [u8p, u8p, u8p, C.c_size_t, C.c_int]
    L.ora_red_overlap.restype = None
    L.ora_red_overlap.argtypes = [u8p, i32p, C.c_uint32]
    L.ora_gray_avg.restype = None
    L.ora_gray_avg.argtypes = [u8p, u8p, C.c_size_t]
    L.ora_gray_weighted.restype = None
    L.ora_gray_weighted.argtypes = [u8p, u8p, C.c_size_t]
    L.ora_gray_weighted_px.restype = C.c_uint8
    L.ora_gray_weighted_px.argtypes = [C.c_uint8, C.c_uint8, C.c_uint8]
    L.ora_histogram.restype = None
    L.ora_histogram.argtypes = [u8p, C.c_size_t, i32p]
    L.ora_two_max_threshold.restype = C.c_int
    L.ora_two_max_threshold.argtypes = [i32p]
    L.ora_binarize.restype = None
    L.ora_binarize.argtypes = [u8p, u8p, C.c_size_t, C.c_int]
    L.ora_server_cpu_branch.restype = C.c_int
    L.ora_server_cpu_branch.argtypes = [u8p, C.c_size_t]
    _LIB = L
    return L


def _u8(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint8).reshape(-1))


# ---- convenience wrappers (numpy in, numpy out) ------------------------------------------------

def diff_pack(cur, state, thr=20):
    """Returns (count, xs[count], diff[count], next_state). `state` is not modified."""
    cur = _u8(cur)
    st = _u8(state).copy()
    n = cur.size
    xs = np.empty(max(n, 1), np.int32)
    df = np.empty(max(n, 1), np.uint8)
    c = lib().ora_diff_pack(cur, st, n, thr, xs, df)
    return int(c), xs[:c].copy(), df[:c].copy(), st


def diff_pack_mt(cur, state, thr=20, nthreads=8):
    cur = _u8(cur)
    st = _u8(state).copy()
    n = cur.size
    xs = np.empty(max(n, 1), np.int32)
    df = np.empty(max(n, 1), np.uint8)
    c = lib().ora_diff_pack_mt(cur, st, n, thr, xs, df, nthreads)
    return int(c), xs[:c].copy(), df[:c].copy(), st


def diff_stream(frames, state, thr=20):
    """frames: (T, n) uint8.  Returns (offsets[T+1], xs, diff, final_state)."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    T, n = frames.shape
    st = _u8(state).copy()
    cap = T * n
    offsets = np.zeros(T + 1, np.uint32)
    xs = np.empty(max(cap, 1), np.int32)
    df = np.empty(max(cap, 1), np.uint8)
    rc = lib().ora_diff_stream(frames.reshape(-1), T, st, n, thr, offsets, xs, df, cap)
    assert rc == 0
    tot = int(offsets[-1])
    return offsets, xs[:tot].copy(), df[:tot].copy(), st


def diff_stream_mt(frames, state, thr=20, nthreads=8, cap=None):
    """diff_stream over `nthreads` host threads (a row band per thread for all frames): identical output."""
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    T, n = frames.shape
    st = _u8(state).copy()
    cap = T * n if cap is None else cap
    offsets = np.zeros(T + 1, np.uint32)
    xs = np.empty(max(cap, 1), np.int32)
    df = np.empty(max(cap, 1), np.uint8)
    rc = lib().ora_diff_stream_mt(frames.reshape(-1), T, st, n, thr, offsets, xs, df, cap, nthreads)
    assert rc == 0, rc
    tot = int(offsets[-1])
    return offsets, xs[:tot].copy(), df[:tot].copy(), st


def client_apply(frame, xs, diff):
    out = _u8(frame).copy()
    xs = np.ascontiguousarray(xs, dtype=np.int32)
    diff = _u8(diff)
    lib().ora_client_apply(out, xs, diff, xs.size)
    return out


def wire_pack(offsets, xs, diff):
    """Bytes the sender thread writes for the frames of a packed stream, server/src/threads.cpp:227-229:
    per frame write(h_pos, 4) ; write(h_xs, h_pos*4) ; write(frame data, h_pos)."""
    offsets = np.asarray(offsets).astype(np.int64)
    out = []
    for t in range(offsets.size - 1):
        a, b = offsets[t], offsets[t + 1]
        out.append(np.uint32(b - a).tobytes())
        out.append(np.ascontiguousarray(xs[a:b], dtype=np.int32).tobytes())
        out.append(np.ascontiguousarray(diff[a:b], dtype=np.uint8).tobytes())
    return np.frombuffer(b"".join(out), np.uint8).copy()


def wire_client(base, wire, nframes):
    """The client's receive loop, client/opencv.cpp:50-66: read pos, pos ints, pos bytes; then
    frame2.data[xs[i]] += buffer[i].  Returns (frames (nframes, n) as displayed, counts)."""
    frame = _u8(base).copy()
    wire = _u8(wire)
    at = 0
    frames = np.empty((nframes, frame.size), np.uint8)
    counts = np.empty(nframes, np.uint32)
    for t in range(nframes):
        pos = int(wire[at:at + 4].view(np.uint32)[0]); at += 4
        xs = wire[at:at + 4 * pos].copy().view(np.int32); at += 4 * pos
        buf = wire[at:at + pos]; at += pos
        lib().ora_client_apply(frame, xs, np.ascontiguousarray(buf), pos)
        frames[t] = frame
        counts[t] = pos
    assert at == wire.size
    return frames, counts


def gaussian_kernel(K=3, sigma=1.5):
    k = np.zeros(K * K, np.float32)
    lib().ora_gaussian_kernel(k, K, sigma)
    return k


def conv3x3(img, w, h, k):
    img = _u8(img)
    out = np.empty_like(img)
    lib().ora_conv3x3(img, out, w, h, np.ascontiguousarray(k, dtype=np.float32))
    return out


def mean_kernel(K):
    k = np.zeros(K * K, np.float32)
    lib().ora_mean_kernel(k, K)
    return k


def conv_kxk(img, w, h, k):
    """tests/noise_filter_benchmark/v2.cu:36-80 with K = sqrt(len(k))."""
    img = _u8(img)
    k = np.ascontiguousarray(k, dtype=np.float32).reshape(-1)
    K = int(round(k.size ** 0.5))
    assert K * K == k.size
    out = np.empty_like(img)
    lib().ora_conv_kxk(img, out, w, h, k, K)
    return out


def conv3x3_intacc(img, w, h, k):
    """tests/noise_filter_benchmark/cpu.cu:72-98 on the frame widened to int (the reference's `int *y`)."""
    i = np.ascontiguousarray(np.asarray(img, dtype=np.uint8).reshape(-1).astype(np.int32))
    out = np.empty_like(i)
    lib().ora_conv3x3_intacc(i, out, w, h, np.ascontiguousarray(k, dtype=np.float32))
    return out


def median5x5(img, w, h):
    img = _u8(img)
    out = np.empty_like(img)
    lib().ora_median5x5(img, out, w, h)
    return out


def heat_lut():
    lut = np.zeros(766 * 3, np.uint8)
    lib().ora_heat_lut(lut)
    return lut.reshape(766, 3)


def heat_map(cur, prev):
    cur, prev = _u8(cur), _u8(prev)
    out = np.empty_like(cur)
    lib().ora_heat_map(cur, prev, out, cur.size // 3)
    return out


def red_dense(cur, prev, thr=20):
    cur, prev = _u8(cur), _u8(prev)
    out = np.empty_like(cur)
    lib().ora_red_dense(cur, prev, out, cur.size // 3, thr)
    return out


def red_overlap(img, xs):
    out = _u8(img).copy()
    xs = np.ascontiguousarray(xs, dtype=np.int32)
    lib().ora_red_overlap(out, xs, xs.size)
    return out


def gray_avg(img):
    img = _u8(img)
    out = np.empty_like(img)
    lib().ora_gray_avg(img, out, img.size // 3)
    return out


def gray_weighted(img):
    img = _u8(img)
    out = np.empty_like(img)
    lib().ora_gray_weighted(img, out, img.size // 3)
    return out


def histogram(gray3):
    gray3 = _u8(gray3)
    h = np.zeros(256, np.int32)
    lib().ora_histogram(gray3, gray3.size, h)
    return h


def two_max_threshold(hist):
    return int(lib().ora_two_max_threshold(np.ascontiguousarray(hist, dtype=np.int32)))


def binarize(img, thr):
    img = _u8(img)
    out = np.empty_like(img)
    lib().ora_binarize(img, out, img.size, thr)
    return out


def server_cpu_branch(frame):
    """server.cpp:96-135 on a copy of frame; returns (out, threshold)."""
    out = _u8(frame).copy()
    thr = lib().ora_server_cpu_branch(out, out.size)
    return out, int(thr)


def ref_server_cpu_path():
    """Path of the reference's own server.cpp (CPU branch) binary, or None if not built."""
    p = os.path.join(_HERE, "_ref", "server_cpu")
    return p if os.path.exists(p) else None


def run_ref_server_cpu(base, frames, w, h, tmpdir):
    """Drive oracle/_ref/server_cpu with seeded frames; returns the processed frames (T, 3wh)."""
    exe = ref_server_cpu_path()
    assert exe is not None
    frames = np.ascontiguousarray(frames, dtype=np.uint8)
    T = frames.shape[0]
    fin = os.path.join(tmpdir, "ref_in.bin")
    fout = os.path.join(tmpdir, "ref_out.bin")
    with open(fin, "wb") as f:
        f.write(np.array([w, h, T], np.int32).tobytes())
        f.write(_u8(base).tobytes())
        f.write(frames.tobytes())
    env = dict(os.environ, REF_IN=fin, REF_OUT=fout)
    subprocess.run([exe], env=env, check=True, stdout=subprocess.DEVNULL, timeout=120)
    out = np.fromfile(fout, dtype=np.uint8)
    return out.reshape(T, 3 * w * h)
