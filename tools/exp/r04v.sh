#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04v
export TMPDIR=/tmp
{
REPS=3 bash tools/exp/run_matrix.sh \
 "device-scope events pipelined|p1||" \
 "system-fence events pipelined|p1|MI355_EVENT_SYSFENCE=1|" \
 "device-scope events seq|p1|MI355_PIPELINE=0|" \
 "system-fence events seq|p1|MI355_PIPELINE=0 MI355_EVENT_SYSFENCE=1|" \
 "device-scope events pipelined K1 1024|p1|MI355_K1_BLOCKS=1024|"
env LD_LIBRARY_PATH=build/ab/p1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04v/tl -- tools/diffbench --steps 12 > /dev/null 2>&1
echo "== timeline"; python3 tools/exp/timeline.py $(find gpurun_out/r04v/tl -name "*kernel_trace.csv" | head -1) 24; rm -rf gpurun_out/r04v/tl
} > gpurun_out/r04v/log.txt 2>&1
cat gpurun_out/r04v/log.txt
