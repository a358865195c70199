#!/bin/bash
# r05l: triples in the expander's group path (three tiles per round where they fit) against the committed library
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05l; mkdir -p $O
{
echo "=== parity (product = triples)"; timeout -k 10 600 python -m pytest tests/test_diff_pack_gpu.py tests/test_fuzz_gpu.py tests/test_stream_ops_gpu.py tests/test_ref_f1f2_gpu.py -x -q 2>&1 | tail -3
for rep in 1 2 3; do
for v in fin tri; do
  echo -n "[$v pipelined] "; LD_LIBRARY_PATH=build/ab/$v timeout -k 5 100 tools/diffbench --steps 30 --digest 2>&1 | tr '\n' ' ' | grep -o 'digest [0-9a-f]*\|"ms_per_step": [0-9.]*, "frac": [0-9.]*\|"kernels_us": [^]]*]' | tr '\n' ' '; echo
  echo -n "[$v sequential] "; LD_LIBRARY_PATH=build/ab/$v timeout -k 5 100 tools/diffbench --steps 30 --opt 1=0 2>&1 | grep -o '"ms_per_step": [0-9.]*, "frac": [0-9.]*\|"kernels_us": [^]]*]' | tr '\n' ' '; echo
done
done
for v in fin tri; do
  echo -n "[$v 4K] "; LD_LIBRARY_PATH=build/ab/$v timeout -k 5 100 tools/diffbench --steps 20 --width 3840 --height 2160 --batch 64 --digest 2>&1 | tr '\n' ' ' | grep -o 'digest [0-9a-f]*\|"ms_per_step": [0-9.]*, "frac": [0-9.]*\|"kernels_us": [^]]*]' | tr '\n' ' '; echo
  echo -n "[$v apart] "; LD_LIBRARY_PATH=build/ab/$v timeout -k 5 100 tools/diffbench --steps 20 --apart --batch 128 --digest 2>&1 | tr '\n' ' ' | grep -o 'digest [0-9a-f]*\|"ms_per_step": [0-9.]*, "frac": [0-9.]*\|"kernels_us": [^]]*]' | tr '\n' ' '; echo
done
} > $O/log.txt 2>&1
cat $O/log.txt
