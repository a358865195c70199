#!/bin/bash
# r06g: pair mode on dense input in ONE pass (k_pair_dense): parity tests under a short timeout, then the S0 / P = N / sparse
# pair regimes with the kernel forced on (--opt 8=2), off (--opt 8=0) and chosen by the totals (default), output arrays from
# mi355_alloc_outputs.
cd ${GRAFT_REPO_ROOT:-.}
O=$PWD/gpurun_out/r06g; mkdir -p $O; : > $O/summary.txt
timeout -k 10 240 python -m pytest tests/test_diff_pack_gpu.py -m gpu -x -q -k "one_pass" > $O/pytest.log 2>&1; rc=$?; echo "pytest one_pass rc $rc" | tee -a $O/summary.txt
tail -12 $O/pytest.log | tee -a $O/summary.txt
[ $rc -ne 0 ] && exit 1
for reg in s0 flip; do for o in 2 0 1; do
  echo "$reg opt8=$o: $(timeout -k 10 100 tools/diffbench --regime $reg --batch 32 --steps 20 --warmup 30 --lib-alloc --opt 8=$o --digest 2>&1 | grep -o 'digest [0-9a-f]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"kernels_us": [^]]*]' | tr '\n' ' ')" | tee -a $O/summary.txt
done; done
for o in 2 0; do
  echo "sparse pairs apart 1080p opt8=$o: $(timeout -k 10 100 tools/diffbench --apart --batch 128 --steps 20 --warmup 30 --opt 8=$o --digest 2>&1 | grep -o 'digest [0-9a-f]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"kernels_us": [^]]*]' | tr '\n' ' ')" | tee -a $O/summary.txt
  echo "sparse pairs apart 4K opt8=$o: $(timeout -k 10 100 tools/diffbench --apart --width 3840 --height 2160 --batch 64 --steps 10 --warmup 20 --opt 8=$o --digest 2>&1 | grep -o 'digest [0-9a-f]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.]*\|"kernels_us": [^]]*]' | tr '\n' ' ')" | tee -a $O/summary.txt
done
timeout -k 10 400 python -m pytest tests/test_diff_pack_gpu.py tests/test_fuzz_gpu.py -m gpu -x -q > $O/pytest_all.log 2>&1; echo "pytest diff_pack+fuzz rc $?" | tee -a $O/summary.txt
tail -3 $O/pytest_all.log | tee -a $O/summary.txt
