#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04af
export TMPDIR=/tmp
{
for round in 1 2; do
REPS=2 bash tools/exp/run_matrix.sh \
 "one pack launch|sp||" \
 "split 50|sp|MI355_SPLIT=50|" \
 "split 45|sp|MI355_SPLIT=45|" \
 "split 40|sp|MI355_SPLIT=40|" \
 "split 30|sp|MI355_SPLIT=30|" \
 "split 50, K1 1536 blocks|sp|MI355_SPLIT=50 MI355_K1_BLOCKS=1536|"
done
env LD_LIBRARY_PATH=build/ab/sp MI355_SPLIT=45 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04af/tl -- tools/diffbench --steps 12 > /dev/null 2>&1
echo "== timeline split 45"; python3 tools/exp/timeline.py $(find gpurun_out/r04af/tl -name "*kernel_trace.csv" | head -1) 30; rm -rf gpurun_out/r04af/tl
} > gpurun_out/r04af/log.txt 2>&1
cat gpurun_out/r04af/log.txt
