"""Host-side mirror of the multi-GPU entry points of the C-ABI (include/mi355diff.h "multi-GPU",
csrc/group.hip): one member per process -- the shape torch.distributed.run gives bench.py -- or several
devices in one process.  The exchange itself (RCCL over xGMI: one all-gather of counts, then point-to-point
sends of index / xs / diff to the root) happens in the library; this file only passes pointers.

The reference has no multi-GPU code (server/src/kernels.cu:385 uses device 0); SURVEY.md section 8e.
"""
import ctypes as C

import numpy as np

from . import lib as _l
from .core import _ptr


def _ptr_array(items):
    arr = (C.c_void_p * len(items))()
    for i, x in enumerate(items):
        arr[i] = _ptr(x)
    return arr


def unique_id():
    """128 bytes naming one group: made by one process, handed to the others (e.g. dist.broadcast)."""
    lib = _l.load()
    buf = np.zeros(_l.GROUP_ID_BYTES, np.uint8)
    _l.check(lib.mi355_group_unique_id(buf.ctypes.data))
    return buf


class CUDAGroup:
    """A group of CUDACores, one per GPU.

    CUDAGroup.adopt(core, nranks, rank, id): this process contributes `core` as rank `rank` (one process per GPU).
    CUDAGroup.create(width, height, ndev, ...): one process drives ndev devices; cores are made by the library.
    """

    def __init__(self, handle, nranks, cores=None):
        self._lib = _l.load()
        self._h = handle
        self.nranks = nranks
        self._cores = cores          # adopted python cores, kept alive

    @classmethod
    def adopt(cls, core, nranks, rank, id128):
        lib = _l.load()
        id128 = np.ascontiguousarray(id128, dtype=np.uint8)
        assert id128.size == _l.GROUP_ID_BYTES
        h = C.c_void_p()
        _l.check(lib.mi355_group_adopt_rank(core._h, int(nranks), int(rank), id128.ctypes.data, C.byref(h)))
        return cls(h, int(nranks), [core])

    @classmethod
    def create(cls, width, height, ndev, devices=None, threshold=20, max_batch=1):
        lib = _l.load()
        cfg = _l.Config(int(width), int(height), int(threshold), int(max_batch), -1, 0, 0, 0)
        devs = None
        if devices is not None:
            devs = (C.c_int * ndev)(*[int(d) for d in devices])
        h = C.c_void_p()
        _l.check(lib.mi355_group_create(C.byref(cfg), int(ndev), devs, C.byref(h)))
        return cls(h, int(ndev))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mi355_group_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def local_members(self):
        return self._lib.mi355_group_local_members(self._h)

    def rank_of(self, i):
        return self._lib.mi355_group_rank_of(self._h, i)

    def core_handle(self, i):
        """Opaque mi355_core* of local member i (for the C-ABI's per-core calls, e.g. mi355_set_state)."""
        return self._lib.mi355_group_core(self._h, i)

    def diff_stream_batch(self, d_frames, nframes, d_offsets, d_xs, d_diff, capacity, stride):
        """Lists indexed by local member."""
        _l.check(self._lib.mi355_group_diff_stream_batch(self._h, _ptr_array(d_frames), stride, nframes,
                                                         _ptr_array(d_offsets), _ptr_array(d_xs),
                                                         _ptr_array(d_diff), capacity))

    def diff_pairs_batch(self, d_cur, d_prev, nframes, d_offsets, d_xs, d_diff, capacity, stride):
        _l.check(self._lib.mi355_group_diff_pairs_batch(self._h, _ptr_array(d_cur), _ptr_array(d_prev), stride,
                                                        nframes, _ptr_array(d_offsets), _ptr_array(d_xs),
                                                        _ptr_array(d_diff), capacity))

    def gather(self, root, nframes, d_offsets, d_xs, d_diff, member_capacity, d_root_offsets=None, d_root_xs=None,
               d_root_diff=None, root_capacity=0):
        """Collective over all ranks.  member_capacity = entries the local members' d_xs / d_diff hold.  Returns
        every rank's total (numpy uint64[nranks]); raises on EVERY rank when a member overflowed its buffers or the
        root's capacity is below the gathered total (decided before anything is sent)."""
        counts = np.zeros(self.nranks, np.uint64)
        _l.check(self._lib.mi355_group_gather(self._h, int(root), int(nframes), _ptr_array(d_offsets),
                                              _ptr_array(d_xs), _ptr_array(d_diff), int(member_capacity),
                                              _ptr(d_root_offsets), _ptr(d_root_xs), _ptr(d_root_diff),
                                              int(root_capacity), counts.ctypes.data_as(C.POINTER(C.c_uint64))))
        return counts

    def synchronize(self):
        _l.check(self._lib.mi355_group_synchronize(self._h))
