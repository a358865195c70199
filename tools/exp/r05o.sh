#!/bin/bash
# r05o: are the pack kernel's log stores dear because a frame's append ends inside a 64-byte sector (partial-sector writes)?
# Laboratory variant: every frame's codes and records are padded (with written zeros) to whole 64-byte / 32-byte pieces.
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05o; mkdir -p $O
{
for rep in 1 2; do
for v in base sect64 sect32; do
  echo -n "[$v sequential] "; LD_LIBRARY_PATH=build/ab/$v timeout -k 5 100 tools/diffbench --steps 30 --opt 1=0 --digest 2>&1 | tr '\n' ' ' | grep -o 'digest [0-9a-f]*\|"ms_per_step": [0-9.]*, "frac": [0-9.]*\|"kernels_us": [^]]*]' | tr '\n' ' '; echo
  echo -n "[$v pipelined ] "; LD_LIBRARY_PATH=build/ab/$v timeout -k 5 100 tools/diffbench --steps 30 2>&1 | grep -o '"ms_per_step": [0-9.]*, "frac": [0-9.]*\|"kernels_us": [^]]*]' | tr '\n' ' '; echo
done
done
} > $O/log.txt 2>&1
cat $O/log.txt
