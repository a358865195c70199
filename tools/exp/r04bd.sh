#!/bin/bash
# configs 3 and 4 on the core's own stream: frame filters beside the expansion of the batch before (product) or after it (MI355_FILTER_JOIN=1)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04bd
export TMPDIR=/tmp
{
for round in 1 2; do
for j in 0 1; do
  MI355_FILTER_JOIN=$j python bench.py --steps 10 --warmup 3 --no-cpu --no-pair --no-host-path > gpurun_out/r04bd/b_$j.json 2>/dev/null
  python3 - $j <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r04bd/b_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print("join=%s headline %.4f | config3 own %.3f seq %.3f | config4 own %.3f seq %.3f"%(sys.argv[1], d['ms_per_step'], d['config3']['us_per_frame'], d['config3']['sequential_us_per_frame'], d['config4']['us_per_frame'], d['config4']['sequential_us_per_frame']))
PY
done
done
} > gpurun_out/r04bd/log.txt 2>&1
cat gpurun_out/r04bd/log.txt
