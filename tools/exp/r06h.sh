#!/bin/bash
# r06h: k_pair_dense, workgroup sizes and occupancies (source-level variants under build/ab/): S0 and P = N, the kernel forced on.
cd ${GRAFT_REPO_ROOT:-.}
O=$PWD/gpurun_out/r06h; mkdir -p $O; : > $O/summary.txt
for v in w4 w8 w4o5 w4o6 w2o6; do for reg in s0 flip; do for i in 1 2; do
  echo "$v $reg $i: $(LD_LIBRARY_PATH=build/ab/$v timeout -k 10 100 tools/diffbench --regime $reg --batch 32 --steps 20 --warmup 30 --lib-alloc --opt 8=2 --digest 2>&1 | grep -o 'digest [0-9a-f]*\|"ms_per_step": [0-9.]*\|"kernels_us": [^]]*]' | tr '\n' ' ')" | tee -a $O/summary.txt
done; done; done
echo "two-pass s0: $(LD_LIBRARY_PATH=build/ab/w4 timeout -k 10 100 tools/diffbench --regime s0 --batch 32 --steps 20 --warmup 30 --lib-alloc --opt 8=0 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"kernels_us": [^]]*]' | tr '\n' ' ')" | tee -a $O/summary.txt
echo "two-pass flip: $(LD_LIBRARY_PATH=build/ab/w4 timeout -k 10 100 tools/diffbench --regime flip --batch 32 --steps 20 --warmup 30 --lib-alloc --opt 8=0 2>&1 | grep -o '"ms_per_step": [0-9.]*\|"kernels_us": [^]]*]' | tr '\n' ' ')" | tee -a $O/summary.txt
