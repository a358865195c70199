#!/bin/bash
# expander thresholds after the positional dense walk: lanes with more than XLIGHT bytes are "heavy" (4), a wave with more than XHEAVYMAX heavy lanes walks densely (12)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04bk
export TMPDIR=/tmp
{
for round in 1 2; do
REPS=1 bash tools/exp/run_matrix.sh \
 "t0 seq|t0|MI355_PIPELINE=0|" "l3 seq|l3|MI355_PIPELINE=0|" "l6 seq|l6|MI355_PIPELINE=0|" "h6 seq|h6|MI355_PIPELINE=0|" "h3 seq|h3|MI355_PIPELINE=0|" "h20 seq|h20|MI355_PIPELINE=0|" \
 "t0 apart seq|t0|MI355_PIPELINE=0|--apart --batch 128" "h3 apart seq|h3|MI355_PIPELINE=0|--apart --batch 128" "h6 apart seq|h6|MI355_PIPELINE=0|--apart --batch 128"
done
} > gpurun_out/r04bk/log.txt 2>&1
python3 - <<'PY'
import re,json
for l in open('gpurun_out/r04bk/log.txt'):
    m=re.match(r'(.*?): digest (\w+) (\{.*\})',l)
    if not m: print(l.strip()[:200]); continue
    d=json.loads(m.group(3)); print("%-14s %s %.4f ms/step  frac %.4f  kernels %s"%(m.group(1),m.group(2),d['ms_per_step'],d['frac'],d['kernels_us']))
PY
