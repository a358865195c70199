#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04z
{
for round in 1 2; do
REPS=2 DIFFBENCH_HBM_PROBE=1 bash tools/exp/run_matrix.sh \
 "r03 diff_pack.hip pipelined 1024|old|MI355_K1_BLOCKS=1024|" \
 "new pipelined 1024|p3|MI355_K1_BLOCKS=1024|" \
 "new v64 pipelined 1024|v64|MI355_K1_BLOCKS=1024|" \
 "r03 pipelined full grid|old|MI355_K1_BLOCKS=0|" \
 "new v64 pipelined full grid|v64|MI355_K1_BLOCKS=0|"
done
} > gpurun_out/r04z/log.txt 2>&1
cat gpurun_out/r04z/log.txt
