#!/bin/bash
# r05q: k_histogram requests its pixels before it clears its 32 KB of bins (the clearing falls into the loads' flight time)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05q; mkdir -p $O
{
echo "=== parity (product = hist)"; timeout -k 10 600 python -m pytest tests/test_filters_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -3
for rep in 1 2 3; do
for v in fin hist; do
  echo "--- filters $v"; LD_LIBRARY_PATH=build/ab/$v timeout -k 5 200 tools/diffbench --filters --batch 192 --steps 5 2>&1 | grep -E "fused|binarize \(gray3|config 3" | cut -c1-150
done
done
} > $O/log.txt 2>&1
cat $O/log.txt
