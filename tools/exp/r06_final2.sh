#!/bin/bash
# r06_final2: behind r06_final.sh -- the whole GPU suite again (a test's assertion on a rounded fraction was too strict) and the
# driver's bench command with the round's counters committed (profiles/pmc_summary.json: roofline.traffic / frac_actual).
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r06final2; mkdir -p $O; : > $O/log.txt
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/suite.txt 2>&1; echo "suite rc $?" >> $O/log.txt; tail -3 $O/suite.txt >> $O/log.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> $O/log.txt
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/log.txt
timeout -k 10 600 python bench.py > $O/bench_default_invocation.json 2> $O/bench_default.err; echo "bench default invocation rc=$?" >> $O/log.txt
cat $O/log.txt
