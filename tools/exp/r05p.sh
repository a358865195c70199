#!/bin/bash
# r05p: long soaks on the final library (tagged frame totals, pinned-memory note, group path)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05p; mkdir -p $O
timeout -k 10 900 python tests/soak.py 9000 > $O/soak.txt 2>&1; echo "soak rc=$?" > $O/log.txt; tail -2 $O/soak.txt >> $O/log.txt
timeout -k 10 900 python tests/soak_chain.py 600 11 > $O/soak_chain.txt 2>&1; echo "soak_chain rc=$?" >> $O/log.txt; tail -2 $O/soak_chain.txt >> $O/log.txt
cat $O/log.txt
