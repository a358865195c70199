#!/bin/bash
# r05g: schedule options of the pipelined batch on the round's library (the index kernel is three times shorter than when
# they were tuned): split share, pack workgroups, dense threshold irrelevant here
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05g; mkdir -p $O
{
for rep in 1 2; do
for o in "" "--opt 2=0" "--opt 2=35" "--opt 2=65" "--opt 5=768" "--opt 5=1280" "--opt 5=1536" "--opt 5=2048" "--opt 5=0" "--opt 5=1280 --opt 2=35" "--opt 5=1536 --opt 2=0"; do
  echo -n "[$o] "; timeout -k 5 100 tools/diffbench --steps 30 $o 2>&1 | grep -o '"ms_per_step": [0-9.]*, "frac": [0-9.]*, "kernel_ms": [0-9.]*, "all_kernels_ms": [0-9.]*, "kernels_us": [^]]*]'
done
done
} > $O/log.txt 2>&1
cat $O/log.txt
