#!/bin/bash
# r05y: band length of the strip median (MI355_OPT_MEDIAN_ROWS) at 1080p x 192 and 4K x 64, two rounds
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05y; mkdir -p $O; : > $O/log.txt
for rep in 1 2; do
for rows in 0 20 30 40 45 60; do
  echo "rows=$rows 1080p: $(timeout -k 10 120 tools/diffbench --filters --batch 192 --steps 10 --opt 6=$rows 2>&1 | grep median)" >> $O/log.txt
done
for rows in 0 30 40 45 60; do
  echo "rows=$rows 4K: $(timeout -k 10 120 tools/diffbench --filters --width 3840 --height 2160 --batch 64 --steps 10 --opt 6=$rows 2>&1 | grep median)" >> $O/log.txt
done
done
cat $O/log.txt
