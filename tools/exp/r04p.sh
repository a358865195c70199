#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04p
{
REPS=2 bash tools/exp/run_matrix.sh \
 "p0 seq|p0|MI355_PIPELINE=0|" \
 "pad8 seq|pad8|MI355_PIPELINE=0|" \
 "pad16 seq|pad16|MI355_PIPELINE=0|" \
 "pad32 seq|pad32|MI355_PIPELINE=0|" \
 "p0 pipelined|p0||" \
 "xp0 (expander prio 0) pipelined|xp0||" \
 "prio (K1 3, X 0) pipelined|prio||" \
 "prio1 (K1 2, X 1) pipelined|prio1||" \
 "prio (K1 3, X 0) seq|prio|MI355_PIPELINE=0|" \
 "p0 pipelined K1 1024 blocks|p0|MI355_K1_BLOCKS=1024|" \
 "prio pipelined K1 1024 blocks|prio|MI355_K1_BLOCKS=1024|" \
 "prio pipelined K1 768 blocks|prio|MI355_K1_BLOCKS=768|"
} > gpurun_out/r04p/log.txt 2>&1
cat gpurun_out/r04p/log.txt
