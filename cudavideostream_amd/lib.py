"""ctypes view of libmi355diff.so -- the C-ABI declared in include/mi355diff.h.

The library is built in-tree (cudavideostream_amd/libmi355diff.so) by `make -C cudavideostream_amd/csrc`
or `__graft_entry__.build()`.  There is no fallback: if the shared object is missing or cannot be
loaded, importing a symbol from here raises.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI355DIFF_LIB: another build of the same library (the laboratory builds of tools/ab_build.sh under build/ab/)
LIB_PATH = os.environ.get("MI355DIFF_LIB") or os.path.join(_HERE, "libmi355diff.so")

OK = 0
ERR_INVALID = -1
ERR_HIP = -2
ERR_STATE = -3

VIS_NONE, VIS_HEAT, VIS_RED, VIS_RED_OVERLAP, VIS_GRAY, VIS_BINARIZE = range(6)
(OP_GRAY_AVG, OP_GRAY_WEIGHTED, OP_BINARIZE, OP_GRAY_AVG_BINARIZE, OP_GRAY_WEIGHTED_BINARIZE, OP_HEAT_MAP,
 OP_RED_DENSE, OP_CONV3X3, OP_MEDIAN5X5) = range(1, 10)

(OPT_PIPELINE, OPT_SPLIT_PCT, OPT_DENSE_PCT, OPT_CHAIN_HINT, OPT_PACK_BLOCKS, OPT_MEDIAN_ROWS,
 OPT_SCAN_EPOCH_LEFT) = range(1, 8)   # MI355_OPT_*
FLAG_OWN_QUEUES = 1   # MI355_FLAG_*
PREPARE_BATCHES, PREPARE_GRAY_CHAIN, PREPARE_RED_CLEAR, PREPARE_CONV_KXK, PREPARE_EXEC, PREPARE_ALL = 1, 2, 4, 8, 16, 31   # MI355_PREPARE_*


class Config(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("threshold", C.c_int32),
        ("max_batch", C.c_int32),
        ("device", C.c_int32),
        ("noise_filter", C.c_int32),
        ("visualizer", C.c_int32),
        ("flags", C.c_int32),
    ]


# name -> (restype, argtypes); every symbol include/mi355diff.h declares
SYMBOLS = {
    "mi355_create": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "mi355_destroy": (None, [C.c_void_p]),
    "mi355_last_error": (C.c_char_p, []),
    "mi355_abi_version": (C.c_int, []),
    "mi355_frame_bytes": (C.c_size_t, [C.c_void_p]),
    "mi355_workspace_bytes": (C.c_size_t, [C.c_void_p]),
    "mi355_prepare": (C.c_int, [C.c_void_p, C.c_uint]),
    "mi355_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mi355_use_own_stream": (C.c_int, [C.c_void_p]),
    "mi355_synchronize": (C.c_int, [C.c_void_p]),
    "mi355_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "mi355_get_option": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "mi355_set_state": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mi355_get_state": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mi355_state_device_ptr": (C.c_void_p, [C.c_void_p]),
    "mi355_set_conv_kernel": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mi355_set_glyphs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_char_p]),
    "mi355_diff_stream_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_size_t]),
    "mi355_diff_pairs_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "mi355_diff_stream_wire_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p,
                                               C.c_void_p, C.c_size_t]),
    "mi355_wire_bytes": (C.c_size_t, [C.c_int, C.c_uint64]),
    "mi355_apply_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_size_t]),
    "mi355_apply_wire_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "mi355_merge_parts": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "mi355_int_diff": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "mi355_gray_avg": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi355_gray_weighted": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi355_binarize_chain": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi355_heat_map": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi355_red_dense": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi355_red_overlap": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    "mi355_red_stream_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                         C.c_int]),
    "mi355_conv3x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi355_conv_kxk": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "mi355_median5x5": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "mi355_filter_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "mi355_exec": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p, C.c_void_p]),
    "mi355_pipe_open": (C.c_int, [C.c_void_p, C.c_int]),
    "mi355_pipe_submit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_void_p,
                                    C.POINTER(C.c_int64)]),
    "mi355_pipe_wait": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_uint32)]),
    "mi355_pipe_close": (C.c_int, [C.c_void_p]),
    "mi355_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "mi355_host_free": (C.c_int, [C.c_void_p]),
    "mi355_dev_alloc": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t]),
    "mi355_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "mi355_alloc_outputs": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "mi355_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "mi355_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "mi355_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "mi355_get_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                   C.POINTER(C.c_int)]),
    "mi355_get_kernel_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                          C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "mi355_reset_timing": (C.c_int, [C.c_void_p]),
    "mi355_probe_clock": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    "mi355_probe_hbm_read": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_double)]),
    "mi355_probe_hbm_write": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)]),
    # multi-GPU (csrc/group.hip)
    "mi355_group_create": (C.c_int, [C.POINTER(Config), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    "mi355_group_unique_id": (C.c_int, [C.c_void_p]),
    "mi355_group_adopt_rank": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "mi355_group_destroy": (None, [C.c_void_p]),
    "mi355_group_ranks": (C.c_int, [C.c_void_p]),
    "mi355_group_local_members": (C.c_int, [C.c_void_p]),
    "mi355_group_core": (C.c_void_p, [C.c_void_p, C.c_int]),
    "mi355_group_rank_of": (C.c_int, [C.c_void_p, C.c_int]),
    "mi355_group_diff_stream_batch": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t, C.c_int,
                                                C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                                C.POINTER(C.c_void_p), C.c_size_t]),
    "mi355_group_diff_pairs_batch": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                               C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                               C.POINTER(C.c_void_p), C.c_size_t]),
    "mi355_group_gather": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_size_t, C.POINTER(C.c_uint64)]),
    "mi355_group_synchronize": (C.c_int, [C.c_void_p]),
}
GROUP_ID_BYTES = 128   # MI355_GROUP_ID_BYTES
ABI_VERSION = 6        # MI355_ABI_VERSION of the include/mi355diff.h these argument lists were written against

_lib = None


def build():
    """Compile libmi355diff.so for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-s"], check=True)


def load():
    """Load the shared library and bind every declared symbol.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch ships its own HIP runtime (libamdhip64 with the system's SONAME).  Whichever copy a process loads
    # first serves every later user; when this library came first, PyTorch ended up on the system runtime and
    # the process saw "no ROCm-capable device" (measured: build() then smoke() in one interpreter).  The Python
    # mirror exists for the tests and bench.py, which hand torch tensors to the library: let torch load first.
    try:
        import torch  # noqa: F401
    except Exception:  # noqa: BLE001  -- a host without PyTorch uses the system runtime, as C++ callers do
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C cudavideostream_amd/csrc` "
            "(there is no non-HIP fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.mi355_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} has ABI version {lib.mi355_abi_version()}, this binding was written for {ABI_VERSION}: "
                           "rebuild the library (make -C cudavideostream_amd/csrc)")
    _lib = lib
    return lib


class Mi355Error(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmi355diff error {code}: {msg}")
        self.code = code


def check(rc):
    if rc != OK:
        raise Mi355Error(rc, load().mi355_last_error().decode(errors="replace"))
