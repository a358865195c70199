#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04a
{
rocm-smi --showclocks --showpower --showmaxpower --showperflevel 2>&1 | head -40
echo ---- amd-smi; (amd-smi metric -g 0 --clock --power 2>&1 | head -60) || true
echo ---- matrix
REPS=2 bash tools/exp/run_matrix.sh \
 "base pipelined|base||" \
 "scan on main|base|MI355_SCAN_MAIN=1|" \
 "side prio|base|MI355_SIDE_PRIO=1|" \
 "scan main + side prio|base|MI355_SCAN_MAIN=1 MI355_SIDE_PRIO=1|" \
 "sequential|base|MI355_PIPELINE=0|" \
 "pf2 pipelined|pf2||" \
 "pf2 scan main|pf2|MI355_SCAN_MAIN=1|" \
 "pf2 sequential|pf2|MI355_PIPELINE=0|" \
 "two cores|base||--cores 2" \
 "two cores scan main|base|MI355_SCAN_MAIN=1|--cores 2" \
 "pairs base|base||--pairs --batch 128" \
 "pairs seq|base|MI355_PIPELINE=0|--pairs --batch 128" \
 "base pipelined again|base||"
} > gpurun_out/r04a/matrix.log 2>&1
cat gpurun_out/r04a/matrix.log
