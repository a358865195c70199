#!/bin/bash
# r05aa: is the slow dense expansion (S0: 262 instead of 205 us per 32 pairs on "some boards") a property of the board
# or of where a process's buffers land?  The same command in six fresh processes, then with the output arrays displaced.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05aa; mkdir -p $O; : > $O/log.txt
for i in 1 2 3 4 5 6; do
  echo "s0 run $i: $(timeout -k 10 120 tools/diffbench --regime s0 --batch 32 --steps 10 2>&1 | tail -1 | cut -c1-400)" >> $O/log.txt
done
for i in 1 2 3; do
  echo "pair run $i: $(timeout -k 10 120 tools/diffbench --pairs --batch 128 --steps 10 2>&1 | tail -1 | cut -c1-400)" >> $O/log.txt
done
cat $O/log.txt
