#!/bin/bash
# r06j: k_pair_dense, third form (counts first, the tiles read again behind the look-back): 2, 3 and 4 workgroups per CU; parity first.
cd ${GRAFT_REPO_ROOT:-.}
O=$PWD/gpurun_out/r06j; mkdir -p $O; : > $O/summary.txt
timeout -k 10 240 python -m pytest tests/test_diff_pack_gpu.py -m gpu -x -q -k "one_pass" > $O/pytest.log 2>&1; rc=$?; echo "pytest one_pass rc $rc" | tee -a $O/summary.txt
tail -3 $O/pytest.log | tee -a $O/summary.txt
[ $rc -ne 0 ] && exit 1
for v in v3 v3o6 v3o8; do for reg in s0 flip; do for i in 1 2; do
  echo "$v $reg $i: $(LD_LIBRARY_PATH=build/ab/$v timeout -k 10 100 tools/diffbench --regime $reg --batch 32 --steps 20 --warmup 30 --lib-alloc --opt 8=2 --digest 2>&1 | grep -o 'digest [0-9a-f]*\|"ms_per_step": [0-9.]*\|"kernels_us": [^]]*]' | tr '\n' ' ')" | tee -a $O/summary.txt
done; done; done
echo "two-pass s0: $(timeout -k 10 100 tools/diffbench --regime s0 --batch 32 --steps 20 --warmup 30 --lib-alloc --opt 8=0 --digest 2>&1 | grep -o 'digest [0-9a-f]*\|"ms_per_step": [0-9.]*\|"kernels_us": [^]]*]' | tr '\n' ' ')" | tee -a $O/summary.txt
echo "two-pass flip: $(timeout -k 10 100 tools/diffbench --regime flip --batch 32 --steps 20 --warmup 30 --lib-alloc --opt 8=0 --digest 2>&1 | grep -o 'digest [0-9a-f]*\|"ms_per_step": [0-9.]*\|"kernels_us": [^]]*]' | tr '\n' ' ')" | tee -a $O/summary.txt
