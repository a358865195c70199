#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04t
{
REPS=1 bash tools/exp/run_matrix.sh \
 "seq alone|p0|MI355_PIPELINE=0|" \
 "seq + valu 1024 waves|p0|MI355_PIPELINE=0|--corun valu --corun-blocks 1024" \
 "seq + valu 2048 waves|p0|MI355_PIPELINE=0|--corun valu --corun-blocks 2048" \
 "seq + valu 4096 waves|p0|MI355_PIPELINE=0|--corun valu --corun-blocks 4096" \
 "seq + mem 1024 waves|p0|MI355_PIPELINE=0|--corun mem --corun-blocks 1024" \
 "seq + mem 2048 waves|p0|MI355_PIPELINE=0|--corun mem --corun-blocks 2048" \
 "seq + mem 4096 waves|p0|MI355_PIPELINE=0|--corun mem --corun-blocks 4096" \
 "seq alone again|p0|MI355_PIPELINE=0|"
} > gpurun_out/r04t/log.txt 2>&1
cat gpurun_out/r04t/log.txt
