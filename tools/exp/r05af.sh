#!/bin/bash
# r05af: does the dense expansion's time (S0: 205 or 265 us per 32 pairs) depend on where its output arrays lie?
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05af; mkdir -p $O; : > $O/log.txt
run() { echo "$* : $(timeout -k 10 120 tools/diffbench --regime s0 --batch 32 --steps 10 --print-ptrs "$@" 2>&1 | tr '\n' ' ' | grep -o 'd_xs [^"]*\|"kernels_us": [^]]*]' | tr '\n' ' ')" >> $O/log.txt; }
run
for s in 256 4096 65536 1048576 2097152 16777216 33554432; do run --skew-df $s; done
for s in 256 4096 65536 1048576 2097152 16777216; do run --skew-xs $s; done
run --skew-xs 4096 --skew-df 65536
cat $O/log.txt
