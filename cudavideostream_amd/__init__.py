"""cudavideostream_amd -- MI355X (gfx950) implementation of the frame-differencing + filter hot
path of MatteoBattilana/CUDAVideoStream behind the reference's own CUDACore call surface.

  csrc/      hand-written HIP kernels + the C-ABI (include/mi355diff.h) -> libmi355diff.so
  compat/    C++ `diff::cuda::CUDACore` drop-in that the reference server.cpp links against
  core.py    the same surface for Python callers (ctypes over the C-ABI)
  synth.py   deterministic synthetic frames
"""
from .core import CHARS_STR, LR_THRESHOLDS, CUDACore, PinnedArray  # noqa: F401
from . import lib  # noqa: F401
