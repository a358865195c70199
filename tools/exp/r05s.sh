#!/bin/bash
# r05s: the histogram's worst case (flat frames: every pixel one bin) with 8 and with 4 private copies of the bins per wave
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05s; mkdir -p $O
{
echo "=== parity (product = 4 copies)"; timeout -k 10 600 python -m pytest tests/test_filters_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -2
for v in hist8 hist4 hist2; do echo "--- $v"; MI355DIFF_LIB=$PWD/build/ab/$v/libmi355diff.so timeout -k 5 200 python tools/exp/r05s_flat_frames.py; done
} > $O/log.txt 2>&1
cat $O/log.txt
