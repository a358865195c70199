#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04s
{
REPS=2 bash tools/exp/run_matrix.sh \
 "seq 256|p0|MI355_PIPELINE=0|" \
 "seq 128|p0|MI355_PIPELINE=0|--batch 128" \
 "seq 96|p0|MI355_PIPELINE=0|--batch 96" \
 "seq 64|p0|MI355_PIPELINE=0|--batch 64" \
 "seq 32|p0|MI355_PIPELINE=0|--batch 32" \
 "pipelined 256|p0||" \
 "pipelined 128|p0||--batch 128" \
 "pipelined 64|p0||--batch 64"
} > gpurun_out/r04s/log.txt 2>&1
cat gpurun_out/r04s/log.txt
