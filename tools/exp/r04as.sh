#!/bin/bash
# fused gray+binarize chain in sub-batches that reuse the first slots of the gray1 scratch (cache-resident gray bytes)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04as
export TMPDIR=/tmp
{
for sub in 0 4 8 16 24 32 48 0 16; do
  echo "sub-batch $sub (batch 96):"; MI355_FUSED_SUB=$sub timeout -k 5 200 python3 tools/bench_filters.py 2>/dev/null | grep "fused (config 3)\|config 3:" | cut -c1-200
done
for sub in 0 8 16 32; do
  echo "sub-batch $sub (batch 192):"; MI355_FUSED_SUB=$sub timeout -k 5 200 python3 tools/bench_filters.py --batch 192 2>/dev/null | grep "fused (config 3)\|config 3:" | cut -c1-200
done
echo "== filter tests with sub-batches of 2"; MI355_FUSED_SUB=2 timeout -k 10 600 python -m pytest tests/test_filters_gpu.py tests/test_server_hip_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -3
} > gpurun_out/r04as/log.txt 2>&1
cat gpurun_out/r04as/log.txt
