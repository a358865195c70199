#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04d
{
REPS=2 bash tools/exp/run_matrix.sh \
 "xfa9 dispatch only|xfa9|MI355_PIPELINE=0|" \
 "xfa8 +first loads|xfa8|MI355_PIPELINE=0|" \
 "xfa1 prologue|xfa1|MI355_PIPELINE=0|"
} > gpurun_out/r04d/log.txt 2>&1
cat gpurun_out/r04d/log.txt
