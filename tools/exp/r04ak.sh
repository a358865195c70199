#!/bin/bash
# histogram pass of the fused gray+binarize chain, timing builds: a1 = no LDS atomics, a2 = no gray1 store, a3 = neither, a4 = a3 + integer gray
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04ak
export TMPDIR=/tmp
{
for v in a0 a1 a2 a3 a4 a0 a1; do
  echo -n "hist ablation $v: "; MI355DIFF_LIB=$PWD/build/ab/$v/libmi355diff.so timeout -k 5 200 python3 tools/bench_filters.py 2>/dev/null | grep "fused (config 3)" | cut -c1-160
done
} > gpurun_out/r04ak/log.txt 2>&1
cat gpurun_out/r04ak/log.txt
