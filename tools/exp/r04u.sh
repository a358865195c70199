#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04u
{
REPS=3 bash tools/exp/run_matrix.sh \
 "OLD expander pipelined|xfast0||" \
 "NEW expander pipelined|p0||" \
 "OLD expander seq|xfast0|MI355_PIPELINE=0|" \
 "NEW expander seq|p0|MI355_PIPELINE=0|" \
 "OLD expander pipelined K1 1024|xfast0|MI355_K1_BLOCKS=1024|" \
 "NEW expander pipelined K1 1024|p0|MI355_K1_BLOCKS=1024|" \
 "OLD two cores|xfast0||--cores 2" \
 "NEW two cores|p0||--cores 2"
} > gpurun_out/r04u/log.txt 2>&1
cat gpurun_out/r04u/log.txt
