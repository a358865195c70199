#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04k
export TMPDIR=/tmp
{
REPS=2 bash tools/exp/run_matrix.sh \
 "x7 seq|x7|MI355_PIPELINE=0|" \
 "x7 pipelined|x7||" \
 "x7 two cores|x7||--cores 2"
for v in x7; do
LD_LIBRARY_PATH=build/ab/$v MI355_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04k/trace_$v -- tools/diffbench --steps 10 > /dev/null 2>&1
echo "== $v"; cat $(find gpurun_out/r04k/trace_$v -name "*kernel_stats.csv" | head -1) | cut -c1-150 | grep -v "webcam\|clock_probe\|rocclr"
done
timeout -k 10 300 python tools/bench_regimes.py
timeout -k 10 900 python -m pytest tests/test_diff_pack_gpu.py tests/test_stream_ops_gpu.py tests/test_ref_f1f2_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -5
} > gpurun_out/r04k/log.txt 2>&1
cat gpurun_out/r04k/log.txt
