#!/bin/bash
# deeper prefetch in the pack kernel (frames per register group: 4 = product, 6, 8) under the pipelined regime, where its loads see longer latencies
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/r04bg
export TMPDIR=/tmp
{
for round in 1 2; do
REPS=1 bash tools/exp/run_matrix.sh \
 "pf4|pf4||" "pf6|pf6||" "pf8|pf8||" \
 "pf6 1280|pf6|MI355_K1_BLOCKS=1280|" "pf8 768|pf8|MI355_K1_BLOCKS=768|" "pf8 1280|pf8|MI355_K1_BLOCKS=1280|" \
 "pf4 seq|pf4|MI355_PIPELINE=0|" "pf6 seq|pf6|MI355_PIPELINE=0|" "pf8 seq|pf8|MI355_PIPELINE=0|"
done
} > gpurun_out/r04bg/log.txt 2>&1
python3 - <<'PY'
import re,json
for l in open('gpurun_out/r04bg/log.txt'):
    m=re.match(r'(.*?): digest (\w+) (\{.*\})',l)
    if not m: print(l.strip()[:200]); continue
    d=json.loads(m.group(3)); print("%-12s %s %.4f ms/step  frac %.4f  kernels %s"%(m.group(1),m.group(2),d['ms_per_step'],d['frac'],d['kernels_us']))
PY
