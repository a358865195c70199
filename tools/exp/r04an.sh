#!/bin/bash
# cache policy: non-temporal output stores in the expander (xnt); plain instead of non-temporal frame loads in the pack kernel (k1nt0)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04an
export TMPDIR=/tmp
{
for round in 1 2; do
REPS=2 bash tools/exp/run_matrix.sh \
 "base pipelined|base||" "xnt pipelined|xnt||" "k1nt0 pipelined|k1nt0||" \
 "base sequential|base|MI355_PIPELINE=0|" "xnt sequential|xnt|MI355_PIPELINE=0|" "k1nt0 sequential|k1nt0|MI355_PIPELINE=0|"
done
} > gpurun_out/r04an/log.txt 2>&1
cut -c1-250 gpurun_out/r04an/log.txt
