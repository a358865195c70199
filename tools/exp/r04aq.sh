#!/bin/bash
# expander log loads non-temporal (n1 codes, n2 records, n3 both, n7 + meta) on top of non-temporal output stores (n0); x0 = plain output stores
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04aq
export TMPDIR=/tmp
{
for round in 1 2 3; do
REPS=1 bash tools/exp/run_matrix.sh \
 "x0|x0||" "n0|n0||" "n1|n1||" "n2|n2||" "n3|n3||" "n7|n7||" \
 "x0 seq|x0|MI355_PIPELINE=0|" "n0 seq|n0|MI355_PIPELINE=0|" "n3 seq|n3|MI355_PIPELINE=0|" "n7 seq|n7|MI355_PIPELINE=0|" \
 "x0 pairs|x0||--pairs --batch 128" "n0 pairs|n0||--pairs --batch 128" "n3 pairs|n3||--pairs --batch 128"
done
} > gpurun_out/r04aq/log.txt 2>&1
python3 - <<'PY'
import re,json
for l in open('gpurun_out/r04aq/log.txt'):
    m=re.match(r'(.*?): digest \w+ (\{.*\})',l)
    if not m: print(l.strip()[:200]); continue
    d=json.loads(m.group(2)); print("%-10s %-6s %.4f ms/step  frac %.4f  kernels %s"%(m.group(1),d['mode'],d['ms_per_step'],d['frac'],d['kernels_us']))
PY
