#!/bin/bash
# filter kernels: non-temporal stores of the visualiser frames (fnt), of the noise filter's output (cnt), both (fcnt)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04at
export TMPDIR=/tmp
{
for v in f0 fnt cnt fcnt f0 fnt; do
  echo "$v:"; MI355DIFF_LIB=$PWD/build/ab/$v/libmi355diff.so timeout -k 5 200 python3 tools/bench_filters.py --batch 192 2>/dev/null | grep -v median | cut -c1-175
done
} > gpurun_out/r04at/log.txt 2>&1
cat gpurun_out/r04at/log.txt
