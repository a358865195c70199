#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04r
{
REPS=2 bash tools/exp/run_matrix.sh \
 "pipelined K1 1024|p0|MI355_K1_BLOCKS=1024|" \
 "pipelined K1 1024 no timing events|p0|MI355_K1_BLOCKS=1024 DIFFBENCH_NO_TIMING=1|" \
 "pipelined no timing events|p0|DIFFBENCH_NO_TIMING=1|" \
 "seq|p0|MI355_PIPELINE=0|" \
 "seq no timing events|p0|MI355_PIPELINE=0 DIFFBENCH_NO_TIMING=1|"
timeout -k 10 900 python -m pytest tests/test_group_gpu.py tests/test_filters_gpu.py -x -q -m gpu 2>&1 | tail -5
} > gpurun_out/r04r/log.txt 2>&1
cat gpurun_out/r04r/log.txt
