#!/bin/bash
# r06r: should MI355_FLAG_OWN_QUEUES be the default?  bench.py WITHOUT a launcher, alternating --own-queues 0 / 1, one box.
cd ${GRAFT_REPO_ROOT:-.}
O=$PWD/gpurun_out/r06r; mkdir -p $O; : > $O/summary.txt
A="--steps 100 --warmup 10 --no-cpu --no-pair --no-filters --no-host-path --no-config5 --preheat-s 2 --steady-steps 1000"
for i in 1 2 3 4; do for q in 0 1; do
  timeout -k 10 200 python bench.py $A --own-queues $q > $O/q${q}_$i.json 2> $O/q${q}_$i.err
  python3 -c "
import json
x=json.loads(open('$O/q${q}_$i.json').read().strip().splitlines()[-1])
print('own-queues $q run $i: value', x['value'], 'ms', x['ms_per_step'], 'steady', x['steady_state']['ms_per_step'], 'cold', x['cold_start_window']['ms_per_step'])" | tee -a $O/summary.txt
done; done
