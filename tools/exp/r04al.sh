#!/bin/bash
# histogram pass: wave-contiguous loads regrouped through LDS (l1) against 16 bytes per lane at a 48-byte stride (l0)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04al
export TMPDIR=/tmp
{
for v in l0 l1 l1a3 l0 l1; do
  echo "hist loads $v: "; MI355DIFF_LIB=$PWD/build/ab/$v/libmi355diff.so timeout -k 5 200 python3 tools/bench_filters.py 2>/dev/null | grep "binarize\|config 3" | cut -c1-160
done
echo "== tests on l1"; MI355DIFF_LIB=$PWD/build/ab/l1/libmi355diff.so timeout -k 10 600 python -m pytest tests/test_filters_gpu.py tests/test_server_hip_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -3
} > gpurun_out/r04al/log.txt 2>&1
cat gpurun_out/r04al/log.txt
