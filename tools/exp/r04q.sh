#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04q
export TMPDIR=/tmp
{
REPS=2 bash tools/exp/run_matrix.sh \
 "p0 pipelined K1 1024|p0|MI355_K1_BLOCKS=1024|" \
 "p0 pipelined K1 1152|p0|MI355_K1_BLOCKS=1152|" \
 "p0 pipelined K1 1280|p0|MI355_K1_BLOCKS=1280|" \
 "p0 pipelined K1 896|p0|MI355_K1_BLOCKS=896|" \
 "p0 pipelined K1 1024 scan main|p0|MI355_K1_BLOCKS=1024 MI355_SCAN_MAIN=1|" \
 "p0 pipelined K1 1024 two cores|p0|MI355_K1_BLOCKS=1024|--cores 2"
for cfg in "MI355_K1_BLOCKS=1024" "MI355_K1_BLOCKS=0"; do
env LD_LIBRARY_PATH=build/ab/p0 $cfg rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04q/tl -- tools/diffbench --steps 12 > /dev/null 2>&1
echo "== timeline $cfg"; python3 tools/exp/timeline.py $(find gpurun_out/r04q/tl -name "*kernel_trace.csv" | head -1) 24; rm -rf gpurun_out/r04q/tl
done
} > gpurun_out/r04q/log.txt 2>&1
cat gpurun_out/r04q/log.txt
timeout -k 10 900 python -m pytest tests/test_group_gpu.py tests/test_filters_gpu.py tests/test_diff_pack_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee -a gpurun_out/r04q/log.txt
