"""AddressSanitizer + UBSan over everything of this repository that runs on the HOST ("race detection /
sanitizers"): tests/sanitize.sh builds the C oracle, the host half of libmi355diff.so (device code is not
instrumented), the C++ CUDACore drop-in, the g++-only tools and the reference's server.cpp CPU branch (where the reference
tree is present) with -fsanitize=address,undefined, runs tests/test_oracle.py and tests/test_cabi.py on the instrumented
libraries, the tools' refusal paths without a GPU and server_cpu on seeded frames; any report fails it.  Build container
only -- GPU AddressSanitizer is not available on the pool and is never attempted (the script needs no GPU)."""
import glob
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"),
                    reason="ROCm clang's shared ASan runtime is not installed")
def test_host_code_is_clean_under_asan_and_ubsan():
    if os.environ.get("MI355_SANITIZED"):
        pytest.skip("already running inside tests/sanitize.sh")
    r = subprocess.run(["bash", os.path.join(ROOT, "tests", "sanitize.sh")], cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "== sanitizers: clean" in out, out[-4000:]
    assert "ERROR: AddressSanitizer" not in out and "runtime error:" not in out
