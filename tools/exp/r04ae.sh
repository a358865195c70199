#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04ae
{
for round in 1 2; do
REPS=2 bash tools/exp/run_matrix.sh \
 "tile in VGPRs seq|ut0|MI355_PIPELINE=0|" \
 "tile uniform seq|ut1|MI355_PIPELINE=0|" \
 "tile in VGPRs pipelined|ut0||" \
 "tile uniform pipelined|ut1||" \
 "tile in VGPRs pairs seq|ut0|MI355_PIPELINE=0|--pairs --batch 128" \
 "tile uniform pairs seq|ut1|MI355_PIPELINE=0|--pairs --batch 128"
done
timeout -k 10 900 python -m pytest tests/test_diff_pack_gpu.py tests/test_stream_ops_gpu.py tests/test_ref_f1f2_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -5
} > gpurun_out/r04ae/log.txt 2>&1
cat gpurun_out/r04ae/log.txt
