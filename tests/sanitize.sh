#!/bin/bash
# CPU-side sanitizer pass ("race detection / sanitizers"): AddressSanitizer + UBSan builds of
# everything that runs on the host -- the C oracle, the host half of libmi355diff.so (core / group / diag; the device
# code is not instrumented: -fno-gpu-sanitize), the C++ CUDACore drop-in, the g++-only tools over the C-ABI and the
# reference's own server.cpp CPU branch with the synthetic ThreadsCore -- and the CPU test files run on those builds.
# Build container only: GPU AddressSanitizer is not available on the pool and is never attempted.
#   bash tests/sanitize.sh            builds into build/asan/ and runs; exit status 0 = clean
set -eu
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
OUT=$ROOT/build/asan; mkdir -p $OUT
LLVM=/opt/rocm/lib/llvm
CC=$LLVM/bin/clang; CXX=$LLVM/bin/clang++
RT=$(ls $LLVM/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -shared-libsan -g -O1"
REF=${REF:-/root/reference}
echo "== build (runtime: $RT)"
$CC $SAN -fPIC -ffp-contract=off -fno-fast-math -std=c11 -Wall -Wextra -shared -o $OUT/liboracle.so oracle/cpu_ref.c -lm -lpthread
( cd cudavideostream_amd/csrc && /opt/rocm/bin/hipcc $SAN -fno-gpu-sanitize -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off \
    -shared -o $OUT/libmi355diff.so core.hip diff_pack.hip filters.hip stream_ops.hip group.hip diag.hip -ldl )
for f in cudacore utils; do $CXX $SAN -std=c++11 -fPIC -Wall -Wextra -c -o $OUT/$f.o cudavideostream_amd/compat/src/$f.cpp; done
ar rcs $OUT/libmi355compat.a $OUT/cudacore.o $OUT/utils.o
for t in roundtrip group_demo; do $CXX $SAN -std=c++11 -Wall -Wextra -o $OUT/$t tools/$t.cpp -L$OUT -lmi355diff -Wl,-rpath,$OUT; done
$CXX $SAN -std=c++11 -Wall -Wextra -o $OUT/compat_pipe tools/compat_pipe.cpp $OUT/libmi355compat.a -L$OUT -lmi355diff -Wl,-rpath,$OUT
if [ -f $REF/server/src/server.cpp ]; then   # the reference's CPU branch, compiled where it lies (oracle/Makefile says how)
  $CXX $SAN -std=c++11 -w -I$REF -DCOMM_H_ -DK=3 -DTILE_SIZE=10 '-DBLOCK_SIZE=(TILE_SIZE+K-1)' '-DCHARS_STR="0123456789BFPSWbkps :/"' \
      -DLR_THRESHOLDS=20 -DKERNEL2_NEGFEED_OPT -DCPU $REF/server/src/server.cpp $REF/server/src/utils.cpp \
      oracle/ref_harness/threads_synth.cpp -o $OUT/server_cpu -lpthread
fi
export LD_LIBRARY_PATH=$(dirname $RT):${LD_LIBRARY_PATH:-}
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
echo "== tests/test_oracle.py + tests/test_cabi.py on the instrumented oracle and library"
LD_PRELOAD=$RT ORACLE_LIB=$OUT/liboracle.so MI355DIFF_LIB=$OUT/libmi355diff.so MI355_SANITIZED=1 \
    python -m pytest tests/test_oracle.py tests/test_cabi.py -x -q -p no:cacheprovider
echo "== the tools' refusal paths without a GPU (every one must fail with the library's message, not with a report)"
for t in roundtrip group_demo compat_pipe; do
  set +e; $OUT/$t > $OUT/$t.out 2>&1; rc=$?; set -e
  if grep -q "ERROR: AddressSanitizer\|runtime error:" $OUT/$t.out; then echo "$t: sanitizer report"; cat $OUT/$t.out; exit 1; fi
  echo "$t: exit $rc, $(tail -1 $OUT/$t.out | cut -c1-120)"
done
if [ -x $OUT/server_cpu ]; then
  echo "== the reference's server.cpp CPU branch (instrumented) on seeded 64x48 frames"
  python - "$OUT" <<'PY'
import os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, os.getcwd())
from cudavideostream_amd import synth
out = sys.argv[1]
base, frames = synth.webcam_stream(6, 64, 48, seed=5)
with tempfile.TemporaryDirectory() as tmp:
    fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
    with open(fin, "wb") as f:
        f.write(np.array([64, 48, 6], np.int32).tobytes()); f.write(np.ascontiguousarray(base).tobytes()); f.write(np.ascontiguousarray(frames).tobytes())
    r = subprocess.run([os.path.join(out, "server_cpu")], env=dict(os.environ, REF_IN=fin, REF_OUT=fout, REF_REPEAT="2"),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = r.stdout.decode(errors="replace")
    bad = "ERROR: AddressSanitizer" in text or "runtime error:" in text
    print("server_cpu: exit", r.returncode, "| output bytes", os.path.getsize(fout) if os.path.exists(fout) else None, "| sanitizer report:", bad)
    if bad:
        print(text[-3000:])
    sys.exit(1 if bad or r.returncode != 0 else 0)
PY
fi
echo "== sanitizers: clean"
