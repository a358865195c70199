#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04o
export TMPDIR=/tmp
{
REPS=2 bash tools/exp/run_matrix.sh \
 "k0 seq|k0|MI355_PIPELINE=0|" \
 "kv vconst seq|kv|MI355_PIPELINE=0|" \
 "kd desc seq|kd|MI355_PIPELINE=0|" \
 "kvd both seq|kvd|MI355_PIPELINE=0|" \
 "k0 seq again|k0|MI355_PIPELINE=0|" \
 "kvd both pipelined|kvd||" \
 "kvd pairs seq|kvd|MI355_PIPELINE=0|--pairs --batch 128" \
 "k0 pairs seq|k0|MI355_PIPELINE=0|--pairs --batch 128"
timeout -k 10 900 python -m pytest tests/test_diff_pack_gpu.py tests/test_stream_ops_gpu.py tests/test_ref_f1f2_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -5
} > gpurun_out/r04o/log.txt 2>&1
cat gpurun_out/r04o/log.txt
