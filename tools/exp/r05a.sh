#!/bin/bash
# r05a: the cleaned library (no lab switches, tagged totals, options API): GPU tests, bench line, conv SQ counters.
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
O=gpurun_out/r05a; mkdir -p $O
{
echo "=== pytest -m gpu"; timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
echo "=== bench"; timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo rc=$?
echo "=== SQ counters, filters (conv)"; bash profiles/pmc_sq.sh r05_filters --filters --batch 96 2>&1 | tail -120
} > $O/log.txt 2>&1
tail -30 $O/log.txt
