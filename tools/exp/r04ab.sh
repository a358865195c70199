#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04ab
{
for round in 1 2; do
REPS=1 bash tools/exp/run_matrix.sh \
 "p4 1024|p4|MI355_K1_BLOCKS=1024|" \
 "p4 896|p4|MI355_K1_BLOCKS=896|" \
 "p4 960|p4|MI355_K1_BLOCKS=960|" \
 "p4 1088|p4|MI355_K1_BLOCKS=1088|" \
 "p4 1152|p4|MI355_K1_BLOCKS=1152|" \
 "p4 768|p4|MI355_K1_BLOCKS=768|" \
 "p4v56 1024|p4v56|MI355_K1_BLOCKS=1024|" \
 "p4 xprio0 1024|p4xp0|MI355_K1_BLOCKS=1024|" \
 "p4 xprio3 1024|p4xp3|MI355_K1_BLOCKS=1024|" \
 "p4 1024 again|p4|MI355_K1_BLOCKS=1024|" \
 "p4 1024 two cores|p4|MI355_K1_BLOCKS=1024|--cores 2"
done
} > gpurun_out/r04ab/log.txt 2>&1
cat gpurun_out/r04ab/log.txt
