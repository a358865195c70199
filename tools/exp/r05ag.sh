#!/bin/bash
# r05ag: the dense expansion against the displacement of its outputs, fine steps (which address bits matter?)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05ag; mkdir -p $O; : > $O/log.txt
run() { echo "$* : $(timeout -k 10 120 tools/diffbench --regime s0 --batch 32 --steps 10 "$@" 2>&1 | tr '\n' ' ' | grep -o '"kernels_us": [^]]*]')" >> $O/log.txt; }
for s in 0 64 128 256 512 768 1024 1280 1536 2048 3072 4096 8192 16384 32768; do run --skew-xs $s; done
for s in 64 128 256 512 1024 2048 4096 8192 16384 32768; do run --skew-df $s; done
for s in 256 512 1024 2048 4096; do run --skew-xs $s --skew-df $s; done
cat $O/log.txt
