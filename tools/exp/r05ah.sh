#!/bin/bash
# r05ah: the dense expansion's two speeds against the store policy of its outputs (non-temporal / plain), at
# displacements that were fast and slow in r05ag
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05ah; mkdir -p $O; : > $O/log.txt
run() { v=$1; shift; echo "$v $* : $(LD_LIBRARY_PATH=build/ab/$v timeout -k 10 120 tools/diffbench --regime s0 --batch 32 --steps 10 "$@" 2>&1 | tr '\n' ' ' | grep -o '"kernels_us": [^]]*]')" >> $O/log.txt; }
for sk in "" "--skew-xs 2048" "--skew-xs 3072" "--skew-df 256" "--skew-df 4096" "--skew-xs 8192"; do
  for v in nt dfplain xsplain bothplain; do run $v $sk; done
done
for v in nt dfplain xsplain bothplain; do
  echo "$v stream: $(LD_LIBRARY_PATH=build/ab/$v timeout -k 10 120 tools/diffbench --batch 256 --steps 20 2>&1 | tr '\n' ' ' | grep -o '"ms_per_step": [0-9.]*')" >> $O/log.txt
done
cat $O/log.txt
