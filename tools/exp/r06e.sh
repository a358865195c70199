#!/bin/bash
# r06e: does a different cache policy of the expander's output stores take the two speeds of the dense expansion away?
# Laboratory builds (csrc/lab.h, MI355_XSTORE): 0 both arrays non-temporal (the product), 1 the value array plain, 2 the index
# array plain, 3 both plain; per build three fresh processes of the S0 regime, each re-drawing its output arrays 3 + 3 + 3 times.
cd ${GRAFT_REPO_ROOT:-.}
O=$PWD/gpurun_out/r06e; mkdir -p $O; : > $O/summary.txt
for v in st0 st1 st2 st3; do
  for i in 1 2 3; do
    LD_LIBRARY_PATH=build/ab/$v timeout -k 10 150 tools/diffbench --regime s0 --batch 32 --steps 10 --warmup 30 --reroll 3 --digest > $O/$v.$i.log 2>&1 || echo "$v $i failed" | tee -a $O/summary.txt
    echo "$v process $i: first $(grep -o '"kernels_us": [^]]*]' $O/$v.$i.log | head -1) $(grep -o 'digest [0-9a-f]*' $O/$v.$i.log)  redraws: $(grep '^reroll outputs\|^reroll xs-only\|^reroll df-only' $O/$v.$i.log | sed 's/.*kernels_us \[[0-9.]*, [0-9.]*, \([0-9.]*\)\].*/\1/' | tr '\n' ' ')" | tee -a $O/summary.txt
  done
  # the headline stream and sparse pairs on the same build (must not get slower)
  echo "$v stream: $(LD_LIBRARY_PATH=build/ab/$v timeout -k 10 120 tools/diffbench --batch 256 --steps 60 --warmup 60 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"kernels_us": [^]]*]' | tr '\n' ' ')" | tee -a $O/summary.txt
done
