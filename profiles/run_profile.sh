#!/bin/bash
# Profiles bench.py's hot path on the GPU box (run through gpurun from the repo root):
#   1. rocprofv3 --kernel-trace --stats      -> per-kernel durations
#   2. rocprofv3 --pmc FETCH_SIZE            -> HBM read bytes   (separate pass, see MI355X_MICROARCH.md)
#   3. rocprofv3 --pmc WRITE_SIZE            -> HBM write bytes  (separate pass)
# Raw output goes to gpurun_out/prof_<tag>/ (scratch); profiles/summarize.py condenses it into the
# committed profiles/<tag>_*.{csv,json}.
set -u
TAG=${1:-r01}
STEPS=${2:-5}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
sha256sum $ROOT/cudavideostream_amd/libmi355diff.so > $OUT/lib.sha256
python3 $ROOT/profiles/srchash.py > $OUT/src.sha256
ARGS="bench.py --steps $STEPS --warmup 2 --no-cpu --no-pair --no-host-path --no-filters --no-config5 --preheat-s 0.2 --steady-steps 100"
cd $ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || echo "trace failed"
# counter passes run the C++ harness (same workload, same library, no Python).  They are NOT attempted on
# `python3 bench.py` any more: that pass crashed in every round (profiles/archive/r02{g,h,i}_pmc_under_python.log) and its
# cause is known -- see profiles/README.md, "counter passes under PyTorch": the process then holds two HSA/HIP
# runtimes (PyTorch's bundled ROCm 7.0 copies and the profiler's /opt/rocm 7.2 ones).
# MI355_PIPELINE=0: one launch of every kernel per batch (a pipelined batch is packed by TWO k_diff_pack launches, half
# the tiles each -- csrc/core.hip, MI355_SPLIT -- and "bytes per launch" would be half a batch); the bytes a kernel moves
# do not depend on what runs beside it.
export MI355_PIPELINE=0
DB="tools/diffbench --steps $STEPS --warmup 2"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $DB > $OUT/fetch.log 2>&1 || echo "fetch failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $DB > $OUT/write.log 2>&1 || echo "write failed"
find $OUT -name "*.csv" | head -20
python3 profiles/summarize.py $OUT $TAG || true
