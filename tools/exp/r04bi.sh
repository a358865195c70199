#!/bin/bash
# chain hint: a batch that follows a frame filter runs one kernel after the other (default) against overlapped regardless (MI355_CHAIN_HINT=0): configs 3 / 4 on the core's own stream
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04bi
export TMPDIR=/tmp
{
for round in 1 2; do
for j in 1 0; do
  MI355_CHAIN_HINT=$j python bench.py --steps 10 --warmup 3 --no-cpu --no-pair --no-host-path > gpurun_out/r04bi/b_$j.json 2>/dev/null
  python3 - $j <<'PY'
import json,sys
d=json.loads(open('gpurun_out/r04bi/b_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print("chain hint %s: headline %.4f | config3 own stream %.3f, caller's stream %.3f | config4 own %.3f, caller's %.3f | parity %s"%(sys.argv[1], d['ms_per_step'], d['config3']['us_per_frame'], d['config3']['sequential_us_per_frame'], d['config4']['us_per_frame'], d['config4']['sequential_us_per_frame'], d.get('parity')))
PY
done
done
echo "== gpu suite"; timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== chain soak"; timeout -k 10 300 python tests/soak_chain.py 400 51 2>&1 | tail -1
} > gpurun_out/r04bi/log.txt 2>&1
cat gpurun_out/r04bi/log.txt
