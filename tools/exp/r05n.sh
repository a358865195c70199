#!/bin/bash
# r05n: what do the per-kernel timing events cost the pipelined batch? (bench.py's timed region runs with them on)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05n; mkdir -p $O
{
for rep in 1 2 3 4; do
  echo -n "[timing on ] "; timeout -k 5 100 tools/diffbench --steps 40 2>&1 | grep -o '"ms_per_step": [0-9.]*, "frac": [0-9.]*'
  echo -n "[timing off] "; DIFFBENCH_NO_TIMING=1 timeout -k 5 100 tools/diffbench --steps 40 2>&1 | grep -o '"ms_per_step": [0-9.]*, "frac": [0-9.]*'
done
} > $O/log.txt 2>&1
cat $O/log.txt
