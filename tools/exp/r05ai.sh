#!/bin/bash
# r05ai: does the STREAM's time depend on where the frames and the outputs lie?  (one box, pipelined and sequential)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05ai; mkdir -p $O; : > $O/log.txt
run() { echo "$* : $(timeout -k 10 120 tools/diffbench --batch 256 --steps 30 "$@" 2>&1 | tr '\n' ' ' | grep -o '"ms_per_step": [0-9.]*\|"kernels_us": [^]]*]' | tr '\n' ' ')" >> $O/log.txt; }
for rep in 1 2; do
run
for s in 256 1024 2048 4096 65536 1048576; do run --skew-frames $s; done
for s in 256 2048 4096 65536; do run --skew-xs $s; done
for s in 256 2048 4096 65536; do run --skew-df $s; done
done
for s in 0 256 2048 4096 65536 1048576; do echo "sequential --skew-frames $s : $(MI355_PIPELINE=0 timeout -k 10 120 tools/diffbench --batch 256 --steps 30 --skew-frames $s 2>&1 | tr '\n' ' ' | grep -o '"ms_per_step": [0-9.]*\|"kernels_us": [^]]*]' | tr '\n' ' ')" >> $O/log.txt; done
cat $O/log.txt
