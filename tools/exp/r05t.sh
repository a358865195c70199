#!/bin/bash
# r05t: why do flat frames take the histogram pass 5 us instead of 2.8?  Laboratory variants (wrong results): without the
# workgroups' global atomics, without the LDS atomics
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05t; mkdir -p $O
{ for v in hprod hnoglob hnolds; do echo "--- $v"; MI355DIFF_LIB=$PWD/build/ab/$v/libmi355diff.so timeout -k 5 200 python tools/exp/r05s_flat_frames.py; done; } > $O/log.txt 2>&1
cat $O/log.txt
