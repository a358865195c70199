#!/bin/bash
# r05ae: SQ counters and the effective clock of the S0 regime's kernels on this box (its kind: see the first line)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05ae; mkdir -p $O
T=$(date -u +%H%M%S)
{ echo "== $T S0"; timeout -k 10 120 tools/diffbench --regime s0 --batch 32 --steps 10 2>&1 | tail -1 | grep -o '"kernels_us": [^]]*]'; } > $O/box_$T.txt
bash profiles/pmc_sq.sh s0box_$T --regime s0 --batch 32 >> $O/box_$T.txt 2>&1
grep -A30 "k_expand" $O/box_$T.txt | head -34; head -2 $O/box_$T.txt; tail -6 $O/box_$T.txt
