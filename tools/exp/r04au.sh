#!/bin/bash
# filters: s0 = plain visualiser stores, s1 = non-temporal (product; red map of the packed stream included), l1 = s1 + non-temporal colour-frame loads in gray / histogram
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04au
export TMPDIR=/tmp
{
for v in s0 s1 l1 s0 s1 l1; do
  echo "$v:"; MI355DIFF_LIB=$PWD/build/ab/$v/libmi355diff.so timeout -k 5 200 python3 tools/bench_filters.py --batch 192 2>/dev/null | grep "gray_w\|config" | cut -c1-175
done
echo "== filter tests (in-tree)"; timeout -k 10 600 python -m pytest tests/test_filters_gpu.py tests/test_server_hip_gpu.py tests/test_fuzz_gpu.py tests/test_ref_f1f2_gpu.py -x -q 2>&1 | tail -3
} > gpurun_out/r04au/log.txt 2>&1
cat gpurun_out/r04au/log.txt
