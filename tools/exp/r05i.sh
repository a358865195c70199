#!/bin/bash
# r05i: waves (items) per workgroup of the expander: 1 (rounds 1-4), 2, 4, 8 -- pipelined, sequential, the bare dispatch
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05i; mkdir -p $O
{
for rep in 1 2; do
for v in xw1 xw2 xw4 xw8; do
  echo -n "[$v pipelined] "; LD_LIBRARY_PATH=build/ab/$v timeout -k 5 100 tools/diffbench --steps 30 --digest 2>&1 | tr '\n' ' ' | grep -o 'digest [0-9a-f]*\|"ms_per_step": [0-9.]*, "frac": [0-9.]*\|"kernels_us": [^]]*]' | tr '\n' ' '; echo
  echo -n "[$v sequential] "; LD_LIBRARY_PATH=build/ab/$v timeout -k 5 100 tools/diffbench --steps 30 --opt 1=0 2>&1 | grep -o '"ms_per_step": [0-9.]*, "frac": [0-9.]*\|"kernels_us": [^]]*]' | tr '\n' ' '; echo
done
done
echo -n "[xw4 dispatch only] "; LD_LIBRARY_PATH=build/ab/xw4d timeout -k 5 100 tools/diffbench --steps 30 --opt 1=0 2>&1 | grep -o '"kernels_us": [^]]*]'
echo "=== parity"; timeout -k 10 600 python -m pytest tests/test_diff_pack_gpu.py tests/test_fuzz_gpu.py tests/test_stream_ops_gpu.py -x -q 2>&1 | tail -3
} > $O/log.txt 2>&1
cat $O/log.txt
