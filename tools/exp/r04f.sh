#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04f
export TMPDIR=/tmp
{
REPS=2 bash tools/exp/run_matrix.sh \
 "x3 K4 seq|x3|MI355_PIPELINE=0|" \
 "x3 K1 seq|x3k1|MI355_PIPELINE=0|" \
 "x3 K2 seq|x3k2|MI355_PIPELINE=0|" \
 "x3 K8 seq|x3k8|MI355_PIPELINE=0|" \
 "x3 K4 pipelined|x3||" \
 "x3 K4 two cores|x3||--cores 2" \
 "x3 4k seq|x3|MI355_PIPELINE=0|--width 3840 --height 2160 --batch 64"
for v in x3 x3k1; do
LD_LIBRARY_PATH=build/ab/$v MI355_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04f/trace_$v -- tools/diffbench --steps 10 > /dev/null 2>&1
echo "== $v"; cat $(find gpurun_out/r04f/trace_$v -name "*kernel_stats.csv" | head -1) | cut -c1-200
done
timeout -k 10 900 python -m pytest tests/test_diff_pack_gpu.py tests/test_stream_ops_gpu.py tests/test_ref_f1f2_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -15
} > gpurun_out/r04f/log.txt 2>&1
cat gpurun_out/r04f/log.txt
