#!/bin/bash
# histogram pass: rounds of 16 Ki pixels per workgroup (fewer contended global atomics per frame)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04aj
export TMPDIR=/tmp
{
for v in r1 r2 r4 r8 r16 r1 r4; do
  echo "hist rounds $v: "; MI355DIFF_LIB=$PWD/build/ab/$v/libmi355diff.so timeout -k 5 200 python3 tools/bench_filters.py 2>/dev/null | grep "binarize\|config 3" | cut -c1-160
done
echo "== tests on r4"; MI355DIFF_LIB=$PWD/build/ab/r4/libmi355diff.so timeout -k 10 600 python -m pytest tests/test_filters_gpu.py tests/test_server_hip_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -3
cd /tmp; MI355DIFF_LIB=$GRAFT_REPO_ROOT/build/ab/r4/libmi355diff.so rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04aj/prof -- python3 $GRAFT_REPO_ROOT/tools/bench_filters.py > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob
for f in glob.glob('gpurun_out/r04aj/prof/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r['Name'] for k in ('histogram','binarize','two_max')): print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
rm -rf gpurun_out/r04aj/prof
} > gpurun_out/r04aj/log.txt 2>&1
cat gpurun_out/r04aj/log.txt
