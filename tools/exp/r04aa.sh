#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04aa
{
for round in 1 2; do
REPS=2 DIFFBENCH_HBM_PROBE=1 bash tools/exp/run_matrix.sh \
 "r03 diff_pack.hip pipelined 1024|old|MI355_K1_BLOCKS=1024|" \
 "new padded pipelined 1024|p4|MI355_K1_BLOCKS=1024|" \
 "r03 pipelined full grid|old|MI355_K1_BLOCKS=0|" \
 "new padded pipelined full grid|p4|MI355_K1_BLOCKS=0|" \
 "r03 seq|old|MI355_PIPELINE=0|" \
 "new padded seq|p4|MI355_PIPELINE=0|"
done
} > gpurun_out/r04aa/log.txt 2>&1
cat gpurun_out/r04aa/log.txt
