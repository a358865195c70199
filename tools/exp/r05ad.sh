#!/bin/bash
# r05ad: which kind of box is this (S0's dense expansion 205 or 265 us), and what do mixed traffic patterns run at on it
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05ad; mkdir -p $O
{ echo "== $(date -u +%H:%M:%S) S0"; timeout -k 10 120 tools/diffbench --regime s0 --batch 32 --steps 10 2>&1 | tail -1 | grep -o '"kernels_us": [^]]*]';
  echo "== stream"; timeout -k 10 120 tools/diffbench --batch 256 --steps 20 2>&1 | tail -1 | grep -o '"ms_per_step": [0-9.]*';
  echo "== mix_probe"; timeout -k 10 120 tools/ubench/mix_probe 2>&1; } >> $O/log.txt
tail -9 $O/log.txt
