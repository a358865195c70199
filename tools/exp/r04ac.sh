#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04ac
{
REPS=2 bash tools/exp/run_matrix.sh \
 "codes+records S0|ro0|MI355_PIPELINE=0|--regime s0 --batch 32" \
 "record-only S0|ro1|MI355_PIPELINE=0|--regime s0 --batch 32" \
 "codes+records P=N|ro0|MI355_PIPELINE=0|--regime flip --batch 32" \
 "record-only P=N|ro1|MI355_PIPELINE=0|--regime flip --batch 32" \
 "record-only stream pipelined|ro1||" \
 "codes+records stream pipelined|ro0||"
timeout -k 10 900 python -m pytest tests/test_diff_pack_gpu.py tests/test_stream_ops_gpu.py tests/test_ref_f1f2_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -5
} > gpurun_out/r04ac/log.txt 2>&1
cat gpurun_out/r04ac/log.txt
