#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04j
export TMPDIR=/tmp
{
REPS=2 bash tools/exp/run_matrix.sh \
 "x6 seq|x6|MI355_PIPELINE=0|" \
 "x4 fast only seq|x4|MI355_PIPELINE=0 MI355_XDEBUG=1|"
for v in x6; do
LD_LIBRARY_PATH=build/ab/$v MI355_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04j/trace_$v -- tools/diffbench --steps 10 > /dev/null 2>&1
echo "== $v"; cat $(find gpurun_out/r04j/trace_$v -name "*kernel_stats.csv" | head -1) | cut -c1-150 | grep -v "webcam\|clock_probe\|rocclr"
done
LD_LIBRARY_PATH=build/ab/x4 MI355_XDEBUG=1 MI355_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04j/trace_x4 -- tools/diffbench --steps 10 > /dev/null 2>&1
echo "== x4 fast only"; cat $(find gpurun_out/r04j/trace_x4 -name "*kernel_stats.csv" | head -1) | cut -c1-150 | grep -v "webcam\|clock_probe\|rocclr"
timeout -k 10 300 python tools/bench_regimes.py
timeout -k 10 900 python -m pytest tests/test_diff_pack_gpu.py tests/test_stream_ops_gpu.py tests/test_ref_f1f2_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -5
} > gpurun_out/r04j/log.txt 2>&1
cat gpurun_out/r04j/log.txt
