#!/bin/bash
# r05u: the histogram's copies of the bins 257 words apart (copies of one bin in different LDS banks): flat frames, webcam frames, parity
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05u; mkdir -p $O
{
echo "=== parity (product = padded rows)"; timeout -k 10 600 python -m pytest tests/test_filters_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -2
for v in hprod hpad; do echo "--- $v"; MI355DIFF_LIB=$PWD/build/ab/$v/libmi355diff.so timeout -k 5 200 python tools/exp/r05s_flat_frames.py; done
for rep in 1 2 3; do
for v in hprod hpad; do
  echo "--- filters $v"; LD_LIBRARY_PATH=build/ab/$v timeout -k 5 200 tools/diffbench --filters --batch 192 --steps 5 2>&1 | grep -E "fused|config 3" | cut -c1-110
done
done
} > $O/log.txt 2>&1
cat $O/log.txt
