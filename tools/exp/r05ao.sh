#!/bin/bash
# r05ao: the headline against the length of the warm-up and of the timed region (one box, alternating)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r05ao; mkdir -p $O; : > $O/log.txt
one() { timeout -k 10 300 python bench.py --steps $1 --warmup $2 --no-cpu --no-pair --no-filters --no-host-path --no-config5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps $1 warmup $2:', d['value'], d['ms_per_step'], d['roofline']['frac'], [k['avg_us'] for k in d['roofline']['kernels']])" >> $O/log.txt; }
for rep in 1 2 3; do one 20 5; one 20 100; one 100 10; one 20 400; done
cat $O/log.txt
