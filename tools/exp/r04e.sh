#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04e
{
REPS=2 bash tools/exp/run_matrix.sh \
 "xfast0 seq|xfast0|MI355_PIPELINE=0|" \
 "x2 K4 seq|x2|MI355_PIPELINE=0|" \
 "x2 K1 seq|x2k1|MI355_PIPELINE=0|" \
 "x2 K2 seq|x2k2|MI355_PIPELINE=0|" \
 "x2 K8 seq|x2k8|MI355_PIPELINE=0|" \
 "x2 K4 pipelined|x2||" \
 "x2 K4 two cores|x2||--cores 2" \
 "x2 4k seq|x2|MI355_PIPELINE=0|--width 3840 --height 2160 --batch 64"
timeout -k 10 900 python -m pytest tests/test_diff_pack_gpu.py tests/test_stream_ops_gpu.py tests/test_ref_f1f2_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -15
} > gpurun_out/r04e/log.txt 2>&1
cat gpurun_out/r04e/log.txt
