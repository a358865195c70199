#!/bin/bash
# r05ay: inside ONE process: the dense expansion after re-drawing its output arrays, then the core (the logs), then the frames
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05ay; mkdir -p $O; : > $O/log.txt
for i in 1 2; do
  echo "== process $i" >> $O/log.txt
  timeout -k 10 200 tools/diffbench --regime s0 --batch 32 --steps 10 --warmup 30 --reroll 5 2>&1 | grep -o 'reroll.*\|"kernels_us": [^]]*]' >> $O/log.txt
done
cat $O/log.txt
