#!/bin/bash
# kernel timeline of the product (two pack launches per batch, non-temporal outputs), pipelined steady state
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04ay
export TMPDIR=/tmp
{
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04ay/tl -- tools/diffbench --steps 14 > /dev/null 2>&1
echo "== timeline, product"; python3 tools/exp/timeline.py $(find gpurun_out/r04ay/tl -name "*kernel_trace.csv" | head -1) 24; rm -rf gpurun_out/r04ay/tl
MI355_LOGSETS=3 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04ay/tl -- tools/diffbench --steps 14 > /dev/null 2>&1
echo "== timeline, 3 log sets"; python3 tools/exp/timeline.py $(find gpurun_out/r04ay/tl -name "*kernel_trace.csv" | head -1) 24; rm -rf gpurun_out/r04ay/tl
} > gpurun_out/r04ay/log.txt 2>&1
cat gpurun_out/r04ay/log.txt
