#!/bin/bash
# r05aj: long soaks on the round's final library (median strip kernel in the chains), and the fuzz / diag tests added since the evidence run
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05aj; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_fuzz_gpu.py tests/test_diag_gpu.py tests/test_filters_gpu.py -x -q > $O/t.txt 2>&1; echo "tests rc=$?" > $O/log.txt; tail -2 $O/t.txt >> $O/log.txt
timeout -k 10 900 python tests/soak.py 9000 > $O/soak.txt 2>&1; echo "soak rc=$?" >> $O/log.txt; tail -2 $O/soak.txt >> $O/log.txt
timeout -k 10 1000 python tests/soak_chain.py 600 11 > $O/soak_chain.txt 2>&1; echo "soak_chain rc=$?" >> $O/log.txt; tail -2 $O/soak_chain.txt >> $O/log.txt
cat $O/log.txt
