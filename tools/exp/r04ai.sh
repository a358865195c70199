#!/bin/bash
# weighted gray: fp64 as written (g1) / integers + exception table (g0) / integers + the fp64 expression for the one pixel in a thousand (g2)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04ai
export TMPDIR=/tmp
{
for v in g1 g0 g2 g1 g2; do
  echo "gray $v: "; MI355DIFF_LIB=$PWD/build/ab/$v/libmi355diff.so timeout -k 5 200 python3 tools/bench_filters.py 2>/dev/null | grep "gray\|config 3" | cut -c1-160
done
echo "batch 192, g2:"; MI355DIFF_LIB=$PWD/build/ab/g2/libmi355diff.so timeout -k 5 200 python3 tools/bench_filters.py --batch 192 2>/dev/null | grep "gray\|config" | cut -c1-160
echo "batch 192, g1:"; MI355DIFF_LIB=$PWD/build/ab/g1/libmi355diff.so timeout -k 5 200 python3 tools/bench_filters.py --batch 192 2>/dev/null | grep "gray\|config" | cut -c1-160
echo "== gray tests on g2"; MI355DIFF_LIB=$PWD/build/ab/g2/libmi355diff.so timeout -k 10 600 python -m pytest tests/test_filters_gpu.py tests/test_server_hip_gpu.py -x -q -k "gray or binarize or config3 or server" 2>&1 | tail -3
} > gpurun_out/r04ai/log.txt 2>&1
cat gpurun_out/r04ai/log.txt
