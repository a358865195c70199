#!/bin/bash
# r05am: which of the dense expansion's two output streams carries its two speeds?  Laboratory builds without the value
# stores / without the index stores / without both, at displacements that are fast and slow on this box
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05am; mkdir -p $O; : > $O/log.txt
run() { v=$1; shift; echo "$v $* : $(LD_LIBRARY_PATH=build/ab/$v timeout -k 10 120 tools/diffbench --regime s0 --batch 32 --steps 10 "$@" 2>&1 | tr '\n' ' ' | grep -o '"kernels_us": [^]]*]')" >> $O/log.txt; }
for sk in "" "--skew-xs 2048" "--skew-xs 3072" "--skew-df 256" "--skew-df 4096" "--skew-xs 8192" "--skew-xs 32768"; do
  for v in nt nodf noxs nostores dfplain; do run $v $sk; done
done
cat $O/log.txt
