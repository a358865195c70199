#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04c
{
REPS=2 bash tools/exp/run_matrix.sh \
 "xf seq|xf|MI355_PIPELINE=0|" \
 "xfa1 prologue|xfa1|MI355_PIPELINE=0|" \
 "xfa2 +codes|xfa2|MI355_PIPELINE=0|" \
 "xfa3 +rounds|xfa3|MI355_PIPELINE=0|" \
 "xfa4 +queued|xfa4|MI355_PIPELINE=0|"
} > gpurun_out/r04c/log.txt 2>&1
cat gpurun_out/r04c/log.txt
