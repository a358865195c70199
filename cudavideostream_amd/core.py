"""Host-side mirror of the reference's operator for the hot path: ``diff::cuda::CUDACore``
(server/include/kernels.cuh:13-43, server/src/kernels.cu:377-536), over the C-ABI of
libmi355diff.so.  Same method names and argument meaning as the C++ class, plus the
device-resident batch entry points used by the benchmark and the parity tests.

PyTorch is used only as plumbing for device buffers / streams (``tensor.data_ptr()``); every
computation happens in the HIP library.
"""
import ctypes as C

import numpy as np

from . import lib as _l

CHARS_STR = "0123456789BFPSWbkps :/"  # server/include/common.h:13
LR_THRESHOLDS = 20                     # server/include/common.h:14


def _ptr(x):
    """Device/host address of a torch tensor, numpy array, int or None."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    return x.data_ptr()


class PinnedArray:
    """numpy view over hipHostMalloc'ed memory (CUDACore::alloc_arrays, kernels.cu:531-536)."""

    def __init__(self, nbytes, dtype=np.uint8):
        self._lib = _l.load()
        p = C.c_void_p()
        _l.check(self._lib.mi355_host_alloc(C.byref(p), nbytes))
        self.ptr = p.value
        buf = (C.c_uint8 * nbytes).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=np.uint8).view(dtype)

    def free(self):
        if self.ptr:
            self.array = None
            _l.check(self._lib.mi355_host_free(self.ptr))
            self.ptr = None


class CUDACore:
    """MI355X drop-in for the reference's CUDACore.

    Reference constructor (kernels.cu:377): CUDACore(charsPx, charsSz, k, total, sampleMatData,
    frameSz).  Here: the same data by keyword; `total` is implied by the frame size.
    """

    def __init__(self, width, height, k=None, sample_mat_data=None, chars_px=None, chars_sz=None,
                 charset=CHARS_STR, threshold=LR_THRESHOLDS, max_batch=1, device=-1,
                 noise_filter=False, visualizer=_l.VIS_NONE, flags=0):
        self._lib = _l.load()
        self.width, self.height = int(width), int(height)
        self.total = 3 * self.width * self.height
        self.max_batch = int(max_batch)
        cfg = _l.Config(self.width, self.height, int(threshold), self.max_batch, int(device),
                        int(bool(noise_filter)), int(visualizer), int(flags))   # flags: lib.FLAG_*
        h = C.c_void_p()
        _l.check(self._lib.mi355_create(C.byref(cfg), C.byref(h)))
        self._h = h
        # The device-resident entry points are asynchronous on the core's OWN stream, which PyTorch's caching allocator
        # knows nothing about: a tensor the caller drops right after the call (a temporary, a slice) could be handed to
        # another torch kernel while the library still reads or writes it.  Every such call therefore keeps its
        # arguments referenced here until synchronize() (include/mi355diff.h, "Lifetime of the caller's buffers").
        self._held = []
        self._own_stream = True
        if k is not None:  # cudaMemcpyToSymbol(dev_k, ...) kernels.cu:394
            k = np.ascontiguousarray(k, dtype=np.float32).reshape(-1)
            assert k.size == 9
            _l.check(self._lib.mi355_set_conv_kernel(self._h, k.ctypes.data))
        if chars_px is not None:  # kernels.cu:379-382
            gh, gw = chars_sz
            chars_px = np.ascontiguousarray(chars_px, dtype=np.uint8).reshape(-1)
            assert chars_px.size == len(charset) * 3 * gh * gw
            _l.check(self._lib.mi355_set_glyphs(self._h, chars_px.ctypes.data, len(charset), gh, gw,
                                                charset.encode()))
        if sample_mat_data is not None:  # kernels.cu:406
            self.set_state(sample_mat_data)

    # -- life cycle -------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.mi355_destroy(self._h)   # completes the core's streams first
            self._h = None
        self._held = []

    def _hold(self, *objs):
        """Keeps the arguments of an asynchronous call alive until the next synchronize()."""
        if not self._own_stream:
            return   # a caller's (torch) stream: the allocator orders reuse on that very stream
        if len(self._held) > 4096:   # a caller that never synchronises: bound the list
            self.synchronize()
        self._held.append([o for o in objs if o is not None and not isinstance(o, (int, np.ndarray))])

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def workspace_bytes(self):
        return self._lib.mi355_workspace_bytes(self._h)

    def prepare(self, what=_l.PREPARE_ALL):
        """mi355_prepare: make now what the entry points would otherwise make on first use (lib.PREPARE_*), so that no
        asynchronous call allocates."""
        _l.check(self._lib.mi355_prepare(self._h, int(what)))

    def alloc_outputs(self, capacity):
        """mi355_alloc_outputs: (d_xs, d_diff, draws) -- device addresses (ints) of an index array and a value array of
        `capacity` entries whose placement lets the dense expansion run at its fast speed; free with dev_free()."""
        xs, df, draws = C.c_void_p(), C.c_void_p(), C.c_int(0)
        _l.check(self._lib.mi355_alloc_outputs(self._h, int(capacity), C.byref(xs), C.byref(df), C.byref(draws)))
        return int(xs.value), int(df.value), draws.value

    def dev_free(self, d_ptr):
        _l.check(self._lib.mi355_dev_free(self._h, C.c_void_p(int(d_ptr))))

    def use_torch_stream(self):
        """Enqueue on PyTorch's current stream (0 = the default stream), so torch ops, events and
        collectives issued on it are ordered with the core's kernels."""
        import torch
        _l.check(self._lib.mi355_set_stream(self._h, C.c_void_p(torch.cuda.current_stream().cuda_stream)))   # waits for the old stream
        self._own_stream = False
        self._held = []

    def use_own_stream(self):
        _l.check(self._lib.mi355_use_own_stream(self._h))
        self._own_stream = True

    def synchronize(self):
        _l.check(self._lib.mi355_synchronize(self._h))
        self._held = []

    def set_option(self, option, value):
        """lib.OPT_*: the schedule of the own-stream batches (include/mi355diff.h, "Options"); never a result."""
        _l.check(self._lib.mi355_set_option(self._h, int(option), int(value)))
        self._held = []   # (the call completed what was queued)

    def get_option(self, option):
        v = C.c_int(0)
        _l.check(self._lib.mi355_get_option(self._h, int(option), C.byref(v)))
        return v.value

    # -- reference surface ------------------------------------------------------------------------
    @staticmethod
    def alloc_arrays(r, c):
        """kernels.cu:531-536: three pinned 3rc(+slack) frames and one pinned 3rc int array."""
        n = 3 * r * c
        slack = CUDACore.chunkt_size()
        h_frame = PinnedArray(n + slack)
        n_frame = PinnedArray(n + slack)
        o_frame = PinnedArray(n + slack)
        h_xs = PinnedArray(n * 4 + slack, np.int32)
        return h_frame, n_frame, o_frame, h_xs

    @staticmethod
    def chunkt_size():
        """kernels.cu:527-529 returns sizeof(long4)=32, the reference's access granule; the slack a
        caller must leave behind its buffers.  This implementation never touches bytes past N, the
        value is kept for callers that size their buffers with it."""
        return 32

    def exec_core(self, frame_data, show_ready_n_data, text, h_xs):
        """kernels.cu:430-525.  frame_data (uint8[>=N], in: frame, out: diff[0..h_pos)),
        show_ready_n_data (uint8[>=N] or None), text (str), h_xs (int32[>=N]).  Returns h_pos."""
        pos = C.c_uint32(0)
        t = text.encode() if text else None
        _l.check(self._lib.mi355_exec(self._h, _ptr(frame_data), _ptr(show_ready_n_data), t,
                                      C.addressof(pos), _ptr(h_xs)))
        return pos.value

    # -- exec_core, pipelined (threads.cpp's ring moved below the boundary) ----------------------------
    def pipe_open(self, depth=3):
        _l.check(self._lib.mi355_pipe_open(self._h, int(depth)))

    def pipe_close(self):
        _l.check(self._lib.mi355_pipe_close(self._h))

    def exec_submit(self, frame_data, show_ready_n_data, text, h_xs):
        """Arguments of exec_core (pinned buffers); returns a ticket for exec_wait."""
        ticket = C.c_int64(-1)
        t = text.encode() if text else None
        _l.check(self._lib.mi355_pipe_submit(self._h, _ptr(frame_data), _ptr(show_ready_n_data), t, _ptr(h_xs),
                                             C.byref(ticket)))
        return ticket.value

    def exec_wait(self, ticket):
        """Blocks until the frame's outputs are in its buffers; returns h_pos."""
        pos = C.c_uint32(0)
        _l.check(self._lib.mi355_pipe_wait(self._h, ticket, C.byref(pos)))
        return pos.value

    # -- state ------------------------------------------------------------------------------------
    def set_state(self, frame):
        frame = np.ascontiguousarray(frame, dtype=np.uint8).reshape(-1)
        assert frame.size == self.total
        _l.check(self._lib.mi355_set_state(self._h, frame.ctypes.data))

    def get_state(self):
        out = np.empty(self.total, np.uint8)
        _l.check(self._lib.mi355_get_state(self._h, out.ctypes.data))
        return out

    def state_ptr(self):
        return self._lib.mi355_state_device_ptr(self._h)

    # -- device-resident hot path -----------------------------------------------------------------
    def diff_stream_batch(self, d_frames, nframes, d_offsets, d_xs, d_diff, capacity, stride=None):
        self._hold(d_frames, d_offsets, d_xs, d_diff)
        stride = self.total if stride is None else stride
        _l.check(self._lib.mi355_diff_stream_batch(self._h, _ptr(d_frames), stride, nframes,
                                                   _ptr(d_offsets), _ptr(d_xs), _ptr(d_diff), capacity))

    def diff_pairs_batch(self, d_cur, d_prev, nframes, d_offsets, d_xs, d_diff, capacity, stride=None):
        self._hold(d_cur, d_prev, d_offsets, d_xs, d_diff)
        stride = self.total if stride is None else stride
        _l.check(self._lib.mi355_diff_pairs_batch(self._h, _ptr(d_cur), _ptr(d_prev), stride, nframes,
                                                  _ptr(d_offsets), _ptr(d_xs), _ptr(d_diff), capacity))

    # -- the stream either side of the path (wire format, client, row-band merge) -------------------
    def diff_stream_wire_batch(self, d_frames, nframes, d_offsets, d_wire, capacity_bytes, stride=None):
        """threads.cpp:227-229 byte stream of the batch: {u32 n, i32 xs[n], u8 diff[n]} per frame."""
        self._hold(d_frames, d_offsets, d_wire)
        stride = self.total if stride is None else stride
        _l.check(self._lib.mi355_diff_stream_wire_batch(self._h, _ptr(d_frames), stride, nframes,
                                                        _ptr(d_offsets), _ptr(d_wire), capacity_bytes))

    def wire_bytes(self, nframes, entries):
        return self._lib.mi355_wire_bytes(nframes, entries)

    def apply_batch(self, d_offsets, d_xs, d_diff, nframes, d_frames_out=None, stride=None):
        """client/opencv.cpp:64-66 on this core's state, frame by frame."""
        self._hold(d_offsets, d_xs, d_diff, d_frames_out)
        stride = self.total if stride is None else stride
        _l.check(self._lib.mi355_apply_batch(self._h, _ptr(d_offsets), _ptr(d_xs), _ptr(d_diff), nframes,
                                             _ptr(d_frames_out), stride))

    def apply_wire_batch(self, d_wire, counts, nframes, d_frames_out=None, stride=None):
        self._hold(d_wire, d_frames_out)
        stride = self.total if stride is None else stride
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        assert counts.size >= nframes
        _l.check(self._lib.mi355_apply_wire_batch(self._h, _ptr(d_wire), counts.ctypes.data, nframes,
                                                  _ptr(d_frames_out), stride))

    def merge_parts(self, d_part_offsets, part_base, xs_bias, d_xs_all, d_diff_all, nframes, d_offsets, d_xs,
                    d_diff, capacity):
        """Row-band streams of one video stream -> the stream of the whole frame."""
        self._hold(d_part_offsets, d_xs_all, d_diff_all, d_offsets, d_xs, d_diff)
        part_base = np.ascontiguousarray(part_base, dtype=np.uint32)
        xs_bias = np.ascontiguousarray(xs_bias, dtype=np.int32)
        assert part_base.size == xs_bias.size
        _l.check(self._lib.mi355_merge_parts(self._h, part_base.size, nframes, _ptr(d_part_offsets),
                                             part_base.ctypes.data, xs_bias.ctypes.data, _ptr(d_xs_all),
                                             _ptr(d_diff_all), _ptr(d_offsets), _ptr(d_xs), _ptr(d_diff),
                                             capacity))

    def int_diff(self, d_cur, d_prev, d_out, n):
        self._hold(d_cur, d_prev, d_out)
        _l.check(self._lib.mi355_int_diff(self._h, _ptr(d_cur), _ptr(d_prev), _ptr(d_out), n))

    # -- filters ----------------------------------------------------------------------------------
    def gray_avg(self, d_in, d_out):
        self._hold(d_in, d_out)
        _l.check(self._lib.mi355_gray_avg(self._h, _ptr(d_in), _ptr(d_out)))

    def gray_weighted(self, d_in, d_out):
        self._hold(d_in, d_out)
        _l.check(self._lib.mi355_gray_weighted(self._h, _ptr(d_in), _ptr(d_out)))

    def binarize_chain(self, d_gray, d_out, d_hist=None, d_thr=None):
        self._hold(d_gray, d_out, d_hist, d_thr)
        _l.check(self._lib.mi355_binarize_chain(self._h, _ptr(d_gray), _ptr(d_out), _ptr(d_hist),
                                                _ptr(d_thr)))

    def conv_kxk(self, d_in, d_out, k):
        """The K x K filter of the reference's filter study (noise_filter_benchmark/v2.cu:36-80); k: K*K floats."""
        self._hold(d_in, d_out)
        k = np.ascontiguousarray(k, dtype=np.float32).reshape(-1)
        K = int(round(k.size ** 0.5))
        assert K * K == k.size
        _l.check(self._lib.mi355_conv_kxk(self._h, _ptr(d_in), _ptr(d_out), k.ctypes.data, K))

    def heat_map(self, d_cur, d_prev, d_out):
        self._hold(d_cur, d_prev, d_out)
        _l.check(self._lib.mi355_heat_map(self._h, _ptr(d_cur), _ptr(d_prev), _ptr(d_out)))

    def red_dense(self, d_cur, d_prev, d_out):
        self._hold(d_cur, d_prev, d_out)
        _l.check(self._lib.mi355_red_dense(self._h, _ptr(d_cur), _ptr(d_prev), _ptr(d_out)))

    def red_overlap(self, d_img, d_xs, d_count=None, count=0):
        self._hold(d_img, d_xs, d_count)
        _l.check(self._lib.mi355_red_overlap(self._h, _ptr(d_img), _ptr(d_xs), _ptr(d_count), count))

    def red_stream_batch(self, d_offsets, d_xs, nframes, d_frames, clear=True, stride=None):
        """Red motion maps of a batch from its packed stream (kernels.cu:513-518 per frame)."""
        self._hold(d_offsets, d_xs, d_frames)
        stride = self.total if stride is None else stride
        _l.check(self._lib.mi355_red_stream_batch(self._h, _ptr(d_offsets), _ptr(d_xs), nframes, _ptr(d_frames),
                                                  stride, int(bool(clear))))

    def conv3x3(self, d_in, d_out):
        self._hold(d_in, d_out)
        _l.check(self._lib.mi355_conv3x3(self._h, _ptr(d_in), _ptr(d_out)))

    def median5x5(self, d_in, d_out):
        self._hold(d_in, d_out)
        _l.check(self._lib.mi355_median5x5(self._h, _ptr(d_in), _ptr(d_out)))

    def filter_batch(self, op, d_in, d_out, nframes, d_in2=None, stride=None):
        """Batched per-frame filter (lib.OP_*): one launch per kernel for nframes frames."""
        self._hold(d_in, d_out, d_in2)
        stride = self.total if stride is None else stride
        _l.check(self._lib.mi355_filter_batch(self._h, int(op), _ptr(d_in), _ptr(d_in2), _ptr(d_out), stride,
                                              nframes))

    # -- measurement ------------------------------------------------------------------------------
    def set_timing(self, on):
        _l.check(self._lib.mi355_set_timing(self._h, int(bool(on))))

    def reset_timing(self):
        _l.check(self._lib.mi355_reset_timing(self._h))

    def get_kernel_timing(self):
        """(ms in k_diff_pack, ms in k_scan_groups, ms in k_expand, launches) since reset."""
        a, b, d, n = C.c_double(0), C.c_double(0), C.c_double(0), C.c_int(0)
        _l.check(self._lib.mi355_get_kernel_timing(self._h, C.byref(a), C.byref(b), C.byref(d), C.byref(n)))
        return a.value, b.value, d.value, n.value

    def probe_clock(self, milliseconds=200):
        """Shader clock (MHz) the device holds under an integer-VALU load (csrc/diag.hip)."""
        mhz = C.c_double(0)
        _l.check(self._lib.mi355_probe_clock(self._h, int(milliseconds), C.byref(mhz)))
        return mhz.value

    def probe_hbm_read(self, megabytes=2048):
        """GB/s of a plain streaming read on this board (csrc/diag.hip)."""
        v = C.c_double(0)
        _l.check(self._lib.mi355_probe_hbm_read(self._h, int(megabytes), C.byref(v)))
        return v.value

    def probe_hbm_write(self, megabytes=2048, narrow=False):
        """GB/s of plain streaming writes on this board: 16 bytes per lane, or (narrow) the dense expansion's
        4-byte index + 1-byte value per lane (csrc/diag.hip)."""
        v = C.c_double(0)
        _l.check(self._lib.mi355_probe_hbm_write(self._h, int(megabytes), 1 if narrow else 0, C.byref(v)))
        return v.value

    def get_timing(self):
        """(ms in the diff/threshold/pack kernel, ms in pack+scan+gather, launches) since reset."""
        a, b, n = C.c_double(0), C.c_double(0), C.c_int(0)
        _l.check(self._lib.mi355_get_timing(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value
