#!/bin/bash
# r05f: rehearsal tests (output unbuffered into the log), then the full bench line on the current product library
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r05f; mkdir -p $O
echo "=== rehearsal tests" > $O/log.txt
timeout -k 10 700 python -m pytest tests/test_rehearsal_gpu.py -x -v --timeout 330 >> $O/log.txt 2>&1
echo "rc=$?" >> $O/log.txt
echo "=== bench" >> $O/log.txt
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/log.txt
tail -5 $O/bench.err >> $O/log.txt
grep -E "PASSED|FAILED|ERROR|passed|failed|rc=" $O/log.txt
