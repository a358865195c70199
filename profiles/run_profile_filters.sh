#!/bin/bash
# rocprofv3 evidence for the filter kernels and the BASELINE config 3 / 4 chains (run through gpurun from the repo
# root): kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in separate passes, all on the C++ harness
# (tools/diffbench --filters: mi355_filter_batch per kernel on a 96-frame 1080p batch, then the two chains).
# profiles/summarize_filters.py condenses the CSVs into profiles/<tag>_filters_*.{csv,json}.
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_filters
mkdir -p $OUT
export TMPDIR=/tmp
cd $ROOT
DB="tools/diffbench --filters --batch 96 --steps 3"
$DB > $OUT/lines.jsonl 2> $OUT/lines.err || echo "plain run failed"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $DB > $OUT/trace.log 2>&1 || echo "trace failed"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $DB > $OUT/fetch.log 2>&1 || echo "fetch failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $DB > $OUT/write.log 2>&1 || echo "write failed"
python3 profiles/summarize_filters.py $OUT $TAG || true
