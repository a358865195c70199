#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04i
export TMPDIR=/tmp
{
REPS=2 bash tools/exp/run_matrix.sh \
 "x5 seq|x5|MI355_PIPELINE=0|" \
 "x5 pipelined|x5||" \
 "x5 two cores|x5||--cores 2" \
 "x5 4k seq|x5|MI355_PIPELINE=0|--width 3840 --height 2160 --batch 64" \
 "x5 pairs seq|x5|MI355_PIPELINE=0|--pairs --batch 128"
timeout -k 10 300 python tools/bench_regimes.py
timeout -k 10 900 python -m pytest tests/test_diff_pack_gpu.py tests/test_stream_ops_gpu.py tests/test_ref_f1f2_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -15
} > gpurun_out/r04i/log.txt 2>&1
cat gpurun_out/r04i/log.txt
