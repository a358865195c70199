#!/bin/bash
# expander prologue: the two prefix words with scalar loads issued beside the meta load (xs1) against plain loads behind the early exit (xs0)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04ba
export TMPDIR=/tmp
{
for round in 1 2 3; do
REPS=1 bash tools/exp/run_matrix.sh \
 "xs0|xs0||" "xs1|xs1||" "xs0 seq|xs0|MI355_PIPELINE=0|" "xs1 seq|xs1|MI355_PIPELINE=0|" \
 "xs0 s0|xs0|MI355_PIPELINE=0|--regime s0 --batch 32" "xs1 s0|xs1|MI355_PIPELINE=0|--regime s0 --batch 32" \
 "xs0 apart|xs0||--apart --batch 128" "xs1 apart|xs1||--apart --batch 128"
done
} > gpurun_out/r04ba/log.txt 2>&1
python3 - <<'PY'
import re,json
for l in open('gpurun_out/r04ba/log.txt'):
    m=re.match(r'(.*?): digest (\w+) (\{.*\})',l)
    if not m: print(l.strip()[:200]); continue
    d=json.loads(m.group(3)); print("%-10s %s %.4f ms/step  frac %.4f  kernels %s"%(m.group(1),m.group(2),d['ms_per_step'],d['frac'],d['kernels_us']))
PY
