#!/bin/bash
# r05az: the value array written by one 16-byte store per quad of lanes (laboratory build) instead of a dword per lane: does
# the dense expansion still have two speeds when the value array is re-drawn?  Same digests?
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05az; mkdir -p $O; : > $O/log.txt
for v in nt df16 nt df16; do
  echo "== $v" >> $O/log.txt
  LD_LIBRARY_PATH=build/ab/$v timeout -k 10 200 tools/diffbench --regime s0 --batch 32 --steps 10 --warmup 30 --reroll 4 --digest 2>&1 | grep -o 'digest.*\|reroll outputs.*\|reroll df-only.*\|reroll xs-only.*' | cut -c1-60 >> $O/log.txt
done
for v in nt df16; do
  echo "$v stream: $(LD_LIBRARY_PATH=build/ab/$v timeout -k 10 120 tools/diffbench --batch 256 --steps 40 --warmup 30 --digest 2>&1 | tr '\n' ' ' | grep -o 'digest [0-9a-f]*\|"ms_per_step": [0-9.]*' | tr '\n' ' ')" >> $O/log.txt
  echo "$v pairs: $(LD_LIBRARY_PATH=build/ab/$v timeout -k 10 120 tools/diffbench --pairs --batch 128 --steps 20 --warmup 30 --digest 2>&1 | tr '\n' ' ' | grep -o 'digest [0-9a-f]*\|"kernels_us": [^]]*]' | tr '\n' ' ')" >> $O/log.txt
done
cat $O/log.txt
