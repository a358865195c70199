"""CPU-side checks of the drop-in boundary: the shared library loads without a GPU, exports every
symbol include/mi355diff.h declares, and fails loudly (no fallback) when no device is present."""
import ctypes as C
import os
import re
import subprocess

import pytest

from cudavideostream_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mi355diff.h")


@pytest.fixture(scope="module")
def built():
    lib.build()
    return lib.load()


def declared_functions():
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    return sorted(set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(built):
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(built, n), f"{n} declared in mi355diff.h but not exported"
        assert n in lib.SYMBOLS, f"{n} has no ctypes prototype in cudavideostream_amd/lib.py"
    assert sorted(lib.SYMBOLS) == names


def test_exports_are_plain_c(built):
    out = subprocess.run(["nm", "-D", "--defined-only", lib.LIB_PATH], capture_output=True, text=True,
                         check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    for n in declared_functions():
        assert n in exported  # unmangled extern "C"


def test_header_compiles_as_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "mi355diff.h"\nint main(void){ mi355_config c; (void)c; return sizeof(c) != 32; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
                    "-o", str(tmp_path / "t")], check=True)
    subprocess.run([str(tmp_path / "t")], check=True)
    assert C.sizeof(lib.Config) == 32


def test_argument_validation_without_gpu(built):
    h = C.c_void_p()
    assert built.mi355_create(None, C.byref(h)) == lib.ERR_INVALID
    bad = lib.Config(64, 48, 300, 1, -1, 0, 0, 0)
    assert built.mi355_create(C.byref(bad), C.byref(h)) == lib.ERR_INVALID
    assert b"threshold" in built.mi355_last_error()
    bad = lib.Config(64, 48, 20, 1, -1, 0, 0, 6)   # cfg.flags: MI355_FLAG_OWN_QUEUES (1) is the only flag; unknown bits are refused
    assert built.mi355_create(C.byref(bad), C.byref(h)) == lib.ERR_INVALID
    assert b"flags" in built.mi355_last_error()
    bad = lib.Config(1920, 1080, 20, 1000, -1, 0, 0, 0)  # 1000 * 6.2 MB >= 2^32
    assert built.mi355_create(C.byref(bad), C.byref(h)) == lib.ERR_INVALID
    # max_batch = 1: the code log (two KiB chunks per tile) is the larger log; a frame within 1 KiB of 2^31 bytes would
    # wrap its 32-bit byte offsets (the record log alone, one KiB per tile, would pass)
    bad = lib.Config(2, 357913771, 20, 1, -1, 0, 0, 0)   # 2 147 482 626 bytes < 2^31: 2^21 tiles, two KiB chunks each = 2^32
    assert built.mi355_create(C.byref(bad), C.byref(h)) == lib.ERR_INVALID
    assert b"2^32" in built.mi355_last_error()
    assert built.mi355_synchronize(None) == lib.ERR_INVALID
    assert built.mi355_frame_bytes(None) == 0


def test_no_cpu_fallback(built):
    """Without a HIP device create() must fail loudly instead of computing on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    cfg = lib.Config(64, 48, 20, 1, -1, 0, 0, 0)
    assert built.mi355_create(C.byref(cfg), C.byref(h)) == lib.ERR_HIP
    assert not h.value
    assert b"no CPU fallback" in built.mi355_last_error()
    from cudavideostream_amd import CUDACore
    with pytest.raises(lib.Mi355Error):
        CUDACore(64, 48)


def test_product_does_not_import_oracle():
    """The package must never reach into oracle/ (test infrastructure): no import, link or path."""
    banned = re.compile(r"import\s+oracle|from\s+oracle|oracle/|liboracle|cpu_ref|pyoracle")
    # the package, the C-ABI header and the measurement/example tools: only tests/, smoke() and bench.py's
    # cpu_baseline leg may touch the oracle
    for top in ("cudavideostream_amd", "include", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp", ".cuh", ".sh")) or f == "Makefile":
                    text = open(os.path.join(dirpath, f), errors="replace").read()
                    assert not banned.search(text), f"{os.path.join(dirpath, f)} references the oracle"
    out = subprocess.run(["ldd", lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_scan_epoch_arithmetic_on_the_host(tmp_path):
    """The index kernel's launch tags (csrc/internal.h): 1 .. 2^33 - 1, never 0, the wrap is signalled (the host then clears
    the totals: core.hip), and a 31-bit total + 33-bit tag share one 64-bit word.  The GPU half of the wrap path is
    tests/test_diff_pack_gpu.py::test_scan_epoch_wrap_clears_the_totals."""
    exe = tmp_path / "epoch_check"
    subprocess.run(["g++", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    "-I" + os.path.join(ROOT, "cudavideostream_amd", "csrc"), os.path.join(ROOT, "tests", "host", "epoch_check.cpp"),
                    "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok 8589934592"), (r.returncode, r.stdout)


def test_prepare_validates_its_arguments_without_gpu(built):
    from cudavideostream_amd import lib
    L = lib.load()
    assert L.mi355_prepare(None, lib.PREPARE_ALL) == lib.ERR_INVALID
