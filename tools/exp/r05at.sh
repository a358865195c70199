#!/bin/bash
# r05at: are the dense expansion's "two speeds" (r05ag: by the displacement of its outputs) still there once the chip has
# reached its clocks?  S0 after 3 warm-up launches (as before: 1 ms of load) and after 100 (37 ms), same displacements
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05at; mkdir -p $O; : > $O/log.txt
run() { echo "$* : $(timeout -k 10 120 tools/diffbench --regime s0 --batch 32 "$@" 2>&1 | tr '\n' ' ' | grep -o '"kernels_us": [^]]*]\|"ms_per_step": [0-9.]*' | tr '\n' ' ')" >> $O/log.txt; }
for sk in "" "--skew-xs 2048" "--skew-xs 3072" "--skew-df 256" "--skew-df 4096" "--skew-xs 8192" "--skew-xs 16384"; do
  run --steps 10 --warmup 3 $sk
  run --steps 30 --warmup 100 $sk
  run --steps 30 --warmup 400 $sk
done
cat $O/log.txt
