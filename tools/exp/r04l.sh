#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04l
{
REPS=1 bash tools/exp/run_matrix.sh \
 "x7stamp seq|x7stamp|MI355_PIPELINE=0|" \
 "x7stamp pipelined|x7stamp||"
} > gpurun_out/r04l/log.txt 2>&1
cat gpurun_out/r04l/log.txt
