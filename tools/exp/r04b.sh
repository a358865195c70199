#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04b
{
REPS=2 bash tools/exp/run_matrix.sh \
 "xfast0 seq|xfast0|MI355_PIPELINE=0|" \
 "xfast seq|xfast|MI355_PIPELINE=0|" \
 "xfast0 pipelined|xfast0||" \
 "xfast pipelined|xfast||" \
 "xfast two cores|xfast||--cores 2" \
 "xfast pairs seq|xfast|MI355_PIPELINE=0|--pairs --batch 128" \
 "xfast 4k seq|xfast|MI355_PIPELINE=0|--width 3840 --height 2160 --batch 64" \
 "xfast0 4k seq|xfast0|MI355_PIPELINE=0|--width 3840 --height 2160 --batch 64"
timeout -k 10 900 python -m pytest tests/test_diff_pack_gpu.py tests/test_stream_ops_gpu.py tests/test_ref_f1f2_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -15
} > gpurun_out/r04b/log.txt 2>&1
cat gpurun_out/r04b/log.txt
