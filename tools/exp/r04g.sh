#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04g
export TMPDIR=/tmp
{
REPS=2 bash tools/exp/run_matrix.sh \
 "x4 seq|x4|MI355_PIPELINE=0|" \
 "x4 pipelined|x4||" \
 "x4 two cores|x4||--cores 2" \
 "x4 4k seq|x4|MI355_PIPELINE=0|--width 3840 --height 2160 --batch 64" \
 "x4 pairs seq|x4|MI355_PIPELINE=0|--pairs --batch 128"
REPS=1 bash tools/exp/run_matrix.sh \
 "x4a1 prologue|x4a1|MI355_PIPELINE=0|" \
 "x4a2 +codes|x4a2|MI355_PIPELINE=0|" \
 "x4a3 +rounds|x4a3|MI355_PIPELINE=0|" \
 "x4a4 +queued|x4a4|MI355_PIPELINE=0|"
for v in x4; do
LD_LIBRARY_PATH=build/ab/$v MI355_PIPELINE=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04g/trace_$v -- tools/diffbench --steps 10 > /dev/null 2>&1
echo "== $v"; cat $(find gpurun_out/r04g/trace_$v -name "*kernel_stats.csv" | head -1) | cut -c1-150 | grep -v "webcam\|clock_probe\|rocclr"
done
timeout -k 10 900 python -m pytest tests/test_diff_pack_gpu.py tests/test_stream_ops_gpu.py tests/test_ref_f1f2_gpu.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | tail -15
} > gpurun_out/r04g/log.txt 2>&1
cat gpurun_out/r04g/log.txt
