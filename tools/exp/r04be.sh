#!/bin/bash
# ordering of the second pack launch (a stream of its own) against the core's stream: the new test on the library before the fix (nofork) and after; cost of the fix
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04be
export TMPDIR=/tmp
{
echo "== the ordering test on the library BEFORE the fix (may fail: that is the bug)"; MI355DIFF_LIB=$PWD/build/ab/nofork/libmi355diff.so timeout -k 10 300 python -m pytest tests/test_diff_pack_gpu.py -q -k "orders_every_part" 2>&1 | tail -3
echo "== the ordering test, fixed library"; timeout -k 10 300 python -m pytest tests/test_diff_pack_gpu.py -q -k "orders_every_part" 2>&1 | tail -2
for round in 1 2; do
REPS=2 bash tools/exp/run_matrix.sh "before|nofork||" "fixed|-||"
done
echo "== gpu suite"; timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== soak"; timeout -k 10 400 python tests/soak.py 2000 2>&1 | tail -1
} > gpurun_out/r04be/log.txt 2>&1
cut -c1-220 gpurun_out/r04be/log.txt
