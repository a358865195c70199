#!/bin/bash
# r06i: k_pair_dense taken apart (source-level variants under build/ab/): without its look-back (wrong offsets), without its emission.
cd ${GRAFT_REPO_ROOT:-.}
O=$PWD/gpurun_out/r06i; mkdir -p $O; : > $O/summary.txt
for v in w8 nolb noemit; do for reg in s0 flip; do
  echo "$v $reg: $(LD_LIBRARY_PATH=build/ab/$v timeout -k 10 100 tools/diffbench --regime $reg --batch 32 --steps 20 --warmup 30 --lib-alloc --opt 8=2 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"kernels_us": [^]]*]' | tr '\n' ' ')" | tee -a $O/summary.txt
done; done
