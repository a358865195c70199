#!/bin/bash
# r05w: the strip median (two bands per register, shared column sorts, pruned selection program): parity, then time
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05w; mkdir -p $O
echo "== median + fuzz tests" > $O/log.txt
timeout -k 10 500 python -m pytest tests/test_filters_gpu.py tests/test_fuzz_gpu.py -x -q -k "median or fuzz" --timeout 400 > $O/t.txt 2>&1; echo "rc=$?" >> $O/log.txt; tail -5 $O/t.txt >> $O/log.txt
echo "== diffbench --filters (192 frames)" >> $O/log.txt
timeout -k 10 300 tools/diffbench --filters --batch 192 --steps 10 > $O/filters192.txt 2>&1; echo "rc=$?" >> $O/log.txt; grep -E "median|conv3x3|gray " $O/filters192.txt >> $O/log.txt
echo "== diffbench --filters (1 frame)" >> $O/log.txt
timeout -k 10 300 tools/diffbench --filters --batch 1 --steps 50 > $O/filters1.txt 2>&1; echo "rc=$?" >> $O/log.txt; grep -E "median|conv3x3" $O/filters1.txt >> $O/log.txt
cat $O/log.txt
