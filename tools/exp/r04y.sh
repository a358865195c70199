#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04y
{
for round in 1 2; do
REPS=2 DIFFBENCH_HBM_PROBE=1 bash tools/exp/run_matrix.sh \
 "r03 diff_pack.hip pipelined|old|MI355_K1_BLOCKS=0|" \
 "new pipelined|p2|MI355_K1_BLOCKS=0|" \
 "new, K1 old loads pipelined|p2nd|MI355_K1_BLOCKS=0|" \
 "r03 diff_pack.hip pipelined 1024|old|MI355_K1_BLOCKS=1024|" \
 "new pipelined 1024|p2|MI355_K1_BLOCKS=1024|" \
 "new, K1 old loads pipelined 1024|p2nd|MI355_K1_BLOCKS=1024|" \
 "r03 seq|old|MI355_PIPELINE=0|" \
 "new seq|p2|MI355_PIPELINE=0|"
done
} > gpurun_out/r04y/log.txt 2>&1
cat gpurun_out/r04y/log.txt
