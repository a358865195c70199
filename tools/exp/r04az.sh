#!/bin/bash
# fused gray+binarize chain: the gray bytes between its two passes non-temporal (g1 stores, g2 loads, g3 both)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04az
export TMPDIR=/tmp
{
for v in g0 g1 g2 g3 g0 g1 g2 g3; do
  echo "$v:"; MI355DIFF_LIB=$PWD/build/ab/$v/libmi355diff.so timeout -k 5 200 python3 tools/bench_filters.py --batch 192 2>/dev/null | grep "fused\|config 3" | cut -c1-175
done
} > gpurun_out/r04az/log.txt 2>&1
cat gpurun_out/r04az/log.txt
