#!/bin/bash
# r05r: private copies of the 256 bins per wave in k_histogram: 8 (32 KB of LDS per workgroup: 5 workgroups per CU), 4, 2
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05r; mkdir -p $O
{
for rep in 1 2 3; do
for v in hist8 hist4 hist2; do
  echo "--- filters $v"; LD_LIBRARY_PATH=build/ab/$v timeout -k 5 200 tools/diffbench --filters --batch 192 --steps 5 2>&1 | grep -E "fused|binarize \(gray3|config 3" | cut -c1-150
done
done
} > $O/log.txt 2>&1
cat $O/log.txt
