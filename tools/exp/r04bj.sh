#!/bin/bash
# expander pair path: per-round facts prepared once in the vector lanes (xr1) against scalar shifts / masks behind every v_readlane (xr0)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04bj
export TMPDIR=/tmp
{
for round in 1 2 3; do
REPS=1 bash tools/exp/run_matrix.sh "xr0 seq|xr0|MI355_PIPELINE=0|" "xr1 seq|xr1|MI355_PIPELINE=0|" "xr0|xr0||" "xr1|xr1||"
done
} > gpurun_out/r04bj/log.txt 2>&1
python3 - <<'PY'
import re,json
for l in open('gpurun_out/r04bj/log.txt'):
    m=re.match(r'(.*?): digest (\w+) (\{.*\})',l)
    if not m: print(l.strip()[:200]); continue
    d=json.loads(m.group(3)); print("%-10s %s %.4f ms/step  frac %.4f  kernels %s"%(m.group(1),m.group(2),d['ms_per_step'],d['frac'],d['kernels_us']))
PY
