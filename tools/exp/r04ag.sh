#!/bin/bash
# split pack as the default (rebuilt after the lost edit) + a third log set; then the suites on that library
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04ag
export TMPDIR=/tmp
{
for round in 1 2; do
REPS=2 bash tools/exp/run_matrix.sh \
 "split 50 (default), 2 sets|-||" \
 "one launch, 2 sets|-|MI355_SPLIT=0|" \
 "split 50, 3 sets|-|MI355_LOGSETS=3|" \
 "one launch, 3 sets|-|MI355_SPLIT=0 MI355_LOGSETS=3|" \
 "split 50, 3 sets, K1 1536|-|MI355_LOGSETS=3 MI355_K1_BLOCKS=1536|"
done
env MI355_LOGSETS=3 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r04ag/tl -- tools/diffbench --steps 12 > /dev/null 2>&1
echo "== timeline split 50, 3 sets"; python3 tools/exp/timeline.py $(find gpurun_out/r04ag/tl -name "*kernel_trace.csv" | head -1) 30; rm -rf gpurun_out/r04ag/tl
echo "== gpu suite"; timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo "== soak"; timeout -k 10 400 python tests/soak.py 3000 2>&1 | tail -2
} > gpurun_out/r04ag/log.txt 2>&1
cat gpurun_out/r04ag/log.txt
